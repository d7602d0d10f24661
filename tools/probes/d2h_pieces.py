"""Probe: the streaming runtime (paffy_hip_stream_*) returns its output at 49 GB/s in most runs and at 9-20 GB/s in some, on a box whose
link gives a plain pinned copy 56 GB/s whatever the buffer or the stream (d2h_buffers.py).  Here: the arrival time of every output
piece of several runs -- is a slow run slow throughout, or does it stall?  GPU box, repo root."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import paffy_amd
from paffy_amd.engine import PlanInfo, Stage, lib

eng = paffy_amd.Engine()
stages = [paffy_amd.stage(paffy_amd.INVERT), paffy_amd.stage(paffy_amd.TRIM_IDENTITY), paffy_amd.stage(paffy_amd.SHATTER)]
chunks = []
for b in range(2):
    buf, nbytes = eng.synth(0x5EED0003, 2048, b * 131072, 131072)
    chunks.append(bytes(buf[:nbytes].cpu().numpy().tobytes()))
L = lib()
piece_bytes = int(os.environ.get("PIECE_MIB", "64")) << 20
for rep in range(int(os.environ.get("REPS", "8"))):
    arr = (Stage * len(stages))(*stages)
    st = C.c_void_p()
    assert L.paffy_hip_stream_open(eng._ctx, arr, len(stages), max(len(c) for c in chunks), piece_bytes, C.byref(st)) == 0
    t0 = time.perf_counter()
    stamps = []
    out_bytes = 0

    def drain():
        global out_bytes
        while True:
            piece, n = C.c_void_p(), C.c_int64()
            assert L.paffy_hip_stream_read(st, C.byref(piece), C.byref(n)) == 0
            if n.value == 0:
                return
            out_bytes += n.value
            stamps.append(time.perf_counter() - t0)

    pending = False
    for chunk in chunks:
        cap = C.c_int64()
        bufp = L.paffy_hip_stream_input(st, len(chunk), 0, C.byref(cap))
        C.memmove(bufp, chunk, len(chunk))
        info = PlanInfo()
        assert L.paffy_hip_stream_submit(st, len(chunk), C.byref(info)) == 0
        if pending:
            drain()
        pending = True
    drain()
    dt = time.perf_counter() - t0
    L.paffy_hip_stream_close(st)
    gaps = [b - a for a, b in zip(stamps, stamps[1:])]
    gaps_ms = sorted(g * 1e3 for g in gaps)
    slow = [round(g, 1) for g in gaps_ms if g > 3.0]
    print(f"run {rep}: {out_bytes / dt / 1e9:.1f} GB/s, {len(stamps)} pieces; gap between pieces: median {gaps_ms[len(gaps_ms) // 2]:.2f} ms, "
          f"p90 {gaps_ms[int(len(gaps_ms) * 0.9)]:.2f} ms, max {gaps_ms[-1]:.1f} ms, {len(slow)} gaps over 3 ms (sum {sum(slow):.0f} ms)", flush=True)
