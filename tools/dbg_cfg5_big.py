#!/usr/bin/env python3
"""Debug aid: tile N synthetic cfg5 records on one GPU, directly and through the sharded worker, and show the tail of the last line when a
run fails.  python tools/dbg_cfg5_big.py 4000000 [per_batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import paffy_amd
from paffy_amd import shard

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
per_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
mode = sys.argv[3] if len(sys.argv) > 3 else "both"
eng = paffy_amd.Engine()


def gen():
    out = []
    for r0 in range(0, n, per_batch):
        buf, nbytes = eng.synth(0x5EED0005, 2048, r0, min(per_batch, n - r0))
        out.append((buf, nbytes))
    return out


def tail(buf, nbytes, k=160):
    return bytes(buf[max(0, nbytes - k): nbytes].cpu().numpy().tobytes())


if mode in ("both", "direct"):
    b = gen()
    print("generated", len(b), "batches", sum(x[1] for x in b) / 1e9, "GB; last line tail:", tail(*b[-1])[-60:], flush=True)
    info = eng.tile_batches(b)
    print("direct tile_batches:", "error", info.error.code, "record", info.error.record, "rows", info.n_rows, "out", info.out_bytes, flush=True)
    del b
if mode in ("both", "shard"):
    b = gen()
    worker = shard.GpuTileWorker(eng)
    try:
        res = shard.tile_sharded(worker, None, 0, 1, b, 0, "cuda", consume=True)
        print("sharded (world 1): ok, lines", res["keys"].shape[0], "total", res["total"], flush=True)
    except RuntimeError as e:
        print("sharded (world 1):", e, flush=True)
        k = worker.keep
        print("pieces", len(k), "last piece bytes", k[-1][1], "tail:", tail(*k[-1])[-80:], flush=True)
        print("first piece head:", bytes(k[0][0][:80].cpu().numpy().tobytes()))
eng.close()
