"""Probe: what a plain streaming write / copy reaches on this GPU (ceiling for the emit kernel)."""
import time
import torch
x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
y = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
for name, fn, nbytes in (("fill", lambda: x.fill_(7), 8 << 30), ("zero", lambda: x.zero_(), 8 << 30), ("copy", lambda: y.copy_(x), 16 << 30)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {nbytes / dt / 1e9:.0f} GB/s")
