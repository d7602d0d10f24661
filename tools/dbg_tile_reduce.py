#!/usr/bin/env python3
"""Debug aid (GPU box): shrink a PAF file on which `tile` differs from the oracle.  python tools/dbg_tile_reduce.py FILE OUT [max_runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import paffy_amd  # noqa: E402

lines = open(sys.argv[1], "rb").read().splitlines(keepends=True)
max_runs = int(sys.argv[3]) if len(sys.argv) > 3 else 300
eng = paffy_amd.Engine()
runs = 0


def bad(ls):
    global runs
    runs += 1
    data = b"".join(ls)
    if not data:
        return False
    want, werr = O.tile(data)
    got, info = eng.tile(data, raise_on_error=False)
    return got != want or info.error.code != werr.code


assert bad(lines)
lo, hi = 1, len(lines)  # smallest failing prefix
while lo < hi:
    mid = (lo + hi) // 2
    if bad(lines[:mid]):
        hi = mid
    else:
        lo = mid + 1
cur = lines[:lo]
print("smallest failing prefix:", len(cur), "lines", flush=True)
chunk = max(1, len(cur) // 2)
while chunk >= 1 and runs < max_runs:
    i, shrunk = 0, False
    while i < len(cur) and runs < max_runs:
        cand = cur[:i] + cur[i + chunk:]
        if cand and bad(cand):
            cur = cand
            shrunk = True
        else:
            i += chunk
    if not shrunk or chunk == 1:
        chunk //= 2
    print("chunk", chunk, "lines", len(cur), "runs", runs, flush=True)
open(sys.argv[2], "wb").write(b"".join(cur))
print("reduced to", len(cur), "lines,", sum(map(len, cur)), "bytes after", runs, "runs")
data = b"".join(cur)
want, _ = O.tile(data)
got, _ = eng.tile(data, raise_on_error=False)
for a, b in zip(got.splitlines(), want.splitlines()):
    if a != b:
        print("got :", a[:160])
        print("want:", b[:160])
if len(cur) <= 12:
    for l in cur:
        print(l[:300])
eng.close()
