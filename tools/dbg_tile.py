import sys, os, random
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import paffy_amd, oracle_lib as O
e = paffy_amd.Engine()
def run(name, data):
    want, werr = O.tile(data)
    got, info = e.tile(data, raise_on_error=False)
    ok = got == want and info.error.code == werr.code
    print(name, "OK" if ok else "FAIL", "gpu err", info.error.code, info.error.record, info.error.aux, "oracle err", werr.code, werr.record, flush=True)
    if not ok and not info.error.code and not werr.code:
        gl, wl = got.splitlines(), want.splitlines()
        print("  lines", len(gl), len(wl))
        bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
        print("  first diffs", bad[:5])
        for i in bad[:3]:
            print("   got ", gl[i][:200]); print("   want", wl[i][:200])
    return ok
ok = b"q\t100\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\tAS:i:9\tcg:Z:5M\n"
run("one", ok)
run("two", ok + ok)
run("indel", b"q\t100\t0\t8\t+\tt\t100\t0\t7\t5\t5\t60\tAS:i:9\tcg:Z:3M3I2M2D\n")
run("long", b"q\t100000\t10\t50010\t+\tt\t100000\t0\t50000\t5\t5\t60\tAS:i:9\tcg:Z:50000M\n")
run("long2", b"q\t100000\t10\t70010\t+\tt\t100000\t0\t70000\t5\t5\t60\tAS:i:9\tcg:Z:30000M10000I30000M10000D\n")
for k in (3, 4, 5, 9, 17):
    run(f"pile{k}", ok * k)
rng = random.Random(1)
recs = []
for r in range(30):
    qs = rng.randrange(0, 50); L = rng.randrange(1, 40)
    recs.append(f"q\t100\t{qs}\t{qs+L}\t+\tt\t100\t0\t{L}\t5\t5\t60\tAS:i:{rng.randrange(100)}\tcg:Z:{L}M\n")
for k in (3, 5, 8, 30):
    run(f"rand{k}", "".join(recs[:k]).encode())
hc = open("tests/golden/human_chimp.paf", "rb").read()
lines = hc.splitlines(keepends=True)
for i in (0, 1, 32):
    run(f"hc{i}", lines[i])
run("hc0-40", b"".join(lines[:40]))
run("hc", hc)
