#!/usr/bin/env python3
"""Locate the records of gpurun_out/fuzz_fail.paf whose shatter output differs from the oracle and show the first differing rows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import paffy_amd  # noqa: E402

data = open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "fuzz_fail.paf"), "rb").read()
pipe = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["4"])]
eng = paffy_amd.Engine()
shown = 0
for ri, line in enumerate(data.splitlines(keepends=True)):
    want, werr = O.run([O.stage(k) for k in pipe], line)
    got, info = eng.run([paffy_amd.stage(k) for k in pipe], line, raise_on_error=False)
    if got != want or info.error.code != werr.code:
        f = line.split(b"\t")
        print(f"record {ri}: qname_len {len(f[0])} tname_len {len(f[5])} fields {f[1:5]} {f[6:12]} ops {line.count(b'M') + line.count(b'I') + line.count(b'D')} err gpu {info.error.code} cpu {werr.code} bytes {len(got)} {len(want)}")
        gw, ww = got.splitlines(), want.splitlines()
        for k, (a, b) in enumerate(zip(gw, ww)):
            if a != b:
                print("  row", k, "of", len(ww))
                print("   gpu:", a[:200])
                print("   cpu:", b[:200])
                break
        shown += 1
        if shown >= 4:
            break
print("done, differing records shown:", shown)
