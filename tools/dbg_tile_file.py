#!/usr/bin/env python3
"""Debug aid: tile a PAF file on the GPU and on the oracle, show where the outputs differ.  python tools/dbg_tile_file.py FILE"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import paffy_amd  # noqa: E402

data = open(sys.argv[1], "rb").read()
want, werr = O.tile(data)
eng = paffy_amd.Engine()
for rep in range(2):
    got, info = eng.tile(data, raise_on_error=False)
    print("lib", os.environ.get("PAFFY_HIP_LIB", "default"), "rep", rep, "codes", info.error.code, werr.code, "len", len(got), len(want), "equal", got == want, flush=True)
    if got != want:
        gl, wl = got.splitlines(), want.splitlines()
        bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
        print("lines", len(gl), len(wl), "differing", len(bad), "first at", bad[:5])
        for i in bad[:3]:
            a, b = gl[i], wl[i]
            k = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), min(len(a), len(b)))
            print(" line", i, "len", len(a), len(b), "first byte diff at", k)
            print("  got :", a[max(0, k - 80): k + 40])
            print("  want:", b[max(0, k - 80): k + 40])
eng.close()
