"""Probe: the streaming runtime (paffy_hip_stream_*) returns its output at 49 GB/s in most runs and at 9-20 GB/s in others, and inside one
process fast and slow runs alternate.  Every run opens a stream (its own pinned pieces, device buffers and two HIP streams).  Here: the
rate of each of several runs and the NUMA node the kernel reports for the pages of its output pieces (/proc/self/numa_maps), to see
whether a slow run is a run whose pinned pieces landed on the other socket.  GPU box, repo root."""
import ctypes as C
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import paffy_amd
from paffy_amd.engine import PlanInfo, Stage, lib


def numa_of(addr):
    """the numa_maps line of the mapping that holds addr: its N<node>=<pages> fields"""
    best = None
    for line in open("/proc/self/numa_maps"):
        a = int(line.split()[0], 16)
        if a <= addr and (best is None or a > best[0]):
            best = (a, line.strip())
    if not best:
        return "?"
    return " ".join(re.findall(r"N\d+=\d+", best[1])) or best[1][:80]


eng = paffy_amd.Engine()
stages = [paffy_amd.stage(paffy_amd.INVERT), paffy_amd.stage(paffy_amd.TRIM_IDENTITY), paffy_amd.stage(paffy_amd.SHATTER)]
chunks = []
for b in range(2):
    buf, nbytes = eng.synth(0x5EED0003, 2048, b * 131072, 131072)
    chunks.append(bytes(buf[:nbytes].cpu().numpy().tobytes()))
ballast = [bytearray(1 << 30) for _ in range(int(os.environ.get("BALLAST_GIB", "0")))]  # host memory held by the process, as bench.py holds its CPU legs' buffers
L = lib()
piece_bytes = int(os.environ.get("PIECE_MIB", "64")) << 20
print("gpu numa nodes:", [open(p).read().strip() for p in sorted(__import__("glob").glob("/sys/class/drm/card*/device/numa_node"))], flush=True)
for rep in range(int(os.environ.get("REPS", "8"))):
    arr = (Stage * len(stages))(*stages)
    st = C.c_void_p()
    assert L.paffy_hip_stream_open(eng._ctx, arr, len(stages), max(len(c) for c in chunks), piece_bytes, C.byref(st)) == 0
    t0 = time.perf_counter()
    stamps, pieces, submits = [], set(), []
    out_bytes = 0

    def drain():
        global out_bytes
        while True:
            piece, n = C.c_void_p(), C.c_int64()
            assert L.paffy_hip_stream_read(st, C.byref(piece), C.byref(n)) == 0
            if n.value == 0:
                return
            out_bytes += n.value
            pieces.add(piece.value)
            stamps.append(time.perf_counter() - t0)

    pending = False
    for chunk in chunks:
        cap = C.c_int64()
        bufp = L.paffy_hip_stream_input(st, len(chunk), 0, C.byref(cap))
        C.memmove(bufp, chunk, len(chunk))
        info = PlanInfo()
        t_s = time.perf_counter()
        assert L.paffy_hip_stream_submit(st, len(chunk), C.byref(info)) == 0
        submits.append((round((t_s - t0) * 1e3, 1), round((time.perf_counter() - t_s) * 1e3, 1)))
        if pending:
            drain()
        pending = True
    drain()
    dt = time.perf_counter() - t0
    where = sorted({numa_of(p) for p in pieces})
    L.paffy_hip_stream_close(st)
    gaps_ms = sorted((b - a) * 1e3 for a, b in zip(stamps, stamps[1:]))
    big = sorted(((stamps[i + 1] - stamps[i]) * 1e3, i) for i in range(len(stamps) - 1))[-3:]
    print(f"   submits (start ms, took ms): {submits}; first piece at {stamps[0] * 1e3:.1f} ms, last at {stamps[-1] * 1e3:.1f} ms; three largest gaps (ms, after piece): {[(round(g, 1), i) for g, i in big]}")
    print(f"run {rep}: {out_bytes / dt / 1e9:.1f} GB/s, median gap between pieces {gaps_ms[len(gaps_ms) // 2]:.2f} ms; pieces at {[hex(p) for p in sorted(pieces)]}: pages on {where}", flush=True)
