#!/usr/bin/env python3
"""Debug aid (GPU box): tile a PAF file with a given build of the library through a handful of C-ABI calls only (older builds lack newer
symbols), compare with the oracle.  python tools/dbg_tile_lib.py LIB.so FILE"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import oracle_lib as O  # noqa: E402
from paffy_amd.engine import PlanInfo  # noqa: E402

L = C.CDLL(sys.argv[1])
data = open(sys.argv[2], "rb").read()
want, werr = O.tile(data)
ctx = C.c_void_p()
assert L.paffy_hip_create(C.byref(ctx), C.c_int(-1)) == 0
d_in = torch.frombuffer(bytearray(data + b"\0" * 64), dtype=torch.uint8).cuda()
info = PlanInfo()
L.paffy_hip_tile_begin.argtypes = [C.c_void_p]
L.paffy_hip_tile_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
L.paffy_hip_tile_run.argtypes = [C.c_void_p, C.POINTER(PlanInfo)]
L.paffy_hip_emit.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert L.paffy_hip_tile_begin(ctx) == 0
assert L.paffy_hip_tile_add(ctx, C.c_void_p(d_in.data_ptr()), len(data)) == 0
rc = L.paffy_hip_tile_run(ctx, C.byref(info))
out = torch.empty(info.out_bytes + 64, dtype=torch.uint8, device="cuda")
rc2 = L.paffy_hip_emit(ctx, C.c_void_p(out.data_ptr()), out.numel())
torch.cuda.synchronize()
got = bytes(out[: info.out_bytes].cpu().numpy().tobytes())
gl, wl = got.splitlines(), want.splitlines()
bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
print(os.path.basename(sys.argv[1]), "rc", rc, rc2, "bytes", len(got), len(want), "equal", got == want, "differing lines", len(bad))
