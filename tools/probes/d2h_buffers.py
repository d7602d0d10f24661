"""Probe: is a slow D2H a property of the pinned buffer (its pages) or of the stream (its copy engine)?  Eight pinned buffers of 64 MiB,
four streams: every (buffer, stream) pair timed (best of 3 copies of 64 MiB from one device buffer).  GPU box."""
import torch

dev = torch.device("cuda", 0)
n = 64 << 20
src = torch.empty(n, dtype=torch.uint8, device=dev)
bufs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(8)]
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
torch.cuda.synchronize()
for rnd in range(2):
    print("round", rnd, "(rows: buffers, columns: streams; GB/s)")
    for b in bufs:
        row = []
        for st in streams:
            best = 0.0
            with torch.cuda.stream(st):
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    b.copy_(src, non_blocking=True)
                    e1.record(st)
                    e1.synchronize()
                    best = max(best, n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
            row.append(f"{best:5.1f}")
        print("  ", " ".join(row), flush=True)
# many back-to-back copies into one buffer on one stream: does the rate hold?
import time
st = streams[0]
for b in bufs[:3]:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(st):
        for _ in range(64):
            b.copy_(src, non_blocking=True)
    st.synchronize()
    print(f"64 copies back to back: {64 * n / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
