"""paffy dedupe (impl/paf_dedupe.c:117-143): oracle known answers on CPU, HIP path vs oracle on the GPU.

No golden outputs exist in the reference for dedupe; the known answers follow its loop: first seen wins on (query name,
target name, strand, four coordinates), cigar text kept verbatim (never parsed), -a also drops a record whose swapped
twin was written and runs paf_check on every record it looks up that way."""
import random
import subprocess

import pytest

import oracle_lib as O
from test_cli import PAFFY

A = b"q\t100\t0\t10\t+\tt\t200\t5\t15\t10\t10\t60\tAS:i:5\tcg:Z:10M\n"
A2 = b"q\t100\t0\t10\t+\tt\t200\t5\t15\t3\t3\t1\tAS:i:99\tcg:Z:5M5X\n"          # same key, other payload: dropped
B = b"q\t100\t0\t10\t-\tt\t200\t5\t15\t10\t10\t60\tcg:Z:10M\n"                      # other strand: kept
AI = b"t\t200\t5\t15\t+\tq\t100\t0\t10\t10\t10\t60\tcg:Z:10M\n"                     # A with query and target swapped
RAW = b"q2\t50\t1\t4\t+\tt\t200\t0\t3\t3\t3\t60\tzz:Z:x\tcg:Z:3S\n"                  # cigar never parsed: bad op letter survives
BADC = b"q3\t50\t10\t4\t+\tt\t200\t0\t3\t3\t3\t60\tcg:Z:3M\n"                       # paf_check fails (only reached with -a)


def norm(line):
    return O.dedupe(line)[0]


def test_oracle_known_answers():
    out, err = O.dedupe(A + A2 + B + AI + A + RAW + RAW)
    assert err.code == 0 and out == norm(A) + norm(B) + norm(AI) + norm(RAW)
    assert norm(RAW).endswith(b"\tcg:Z:3S\n") and b"zz:Z" not in norm(RAW)   # verbatim cigar text, unknown tag dropped
    out, err = O.dedupe(A + A2 + B + AI + A + RAW, check_inverse=True)
    assert err.code == 0 and out == norm(A) + norm(B) + norm(RAW)              # the swapped twin is gone
    out, err = O.dedupe(A + BADC + B)                                          # no paf_check without -a
    assert err.code == 0 and out.count(b"\n") == 3
    out, err = O.dedupe(A + BADC + B, check_inverse=True)
    assert err.code != 0 and O.exit_status(err.code) == 1 and err.record == 1 and out == norm(A)  # st_errAbort in paf_check


def build_dupes(rng, base, n):
    lines = base.splitlines(keepends=True)
    out = []
    for _ in range(n):
        l = rng.choice(lines)
        r = rng.random()
        if r < 0.3:  # swapped twin
            f = l.rstrip(b"\n").split(b"\t")
            f[0], f[5] = f[5], f[0]
            f[1], f[6] = f[6], f[1]
            f[2], f[7] = f[7], f[2]
            f[3], f[8] = f[8], f[3]
            l = b"\t".join(f) + b"\n"
        elif r < 0.5:  # near miss: one coordinate off
            f = l.rstrip(b"\n").split(b"\t")
            f[3] = str(int(f[3]) + 1).encode()
            l = b"\t".join(f) + b"\n"
        out.append(l)
    return b"".join(out)


@pytest.mark.gpu
def test_gpu_dedupe_matches_oracle(human_chimp):
    import paffy_amd

    eng = paffy_amd.Engine()
    rng = random.Random(9)
    data = build_dupes(rng, human_chimp[:200000] + A + B + AI + RAW, 3000)
    for inv in (False, True):
        want, werr = O.dedupe(data, inv)
        got, info = eng.dedupe(data, inv, raise_on_error=False)
        assert info.error.code == werr.code and got == want and 0 < got.count(b"\n") < 3000
        # the same stream in three batches through one context: earlier batches are remembered
        lines = data.splitlines(keepends=True)
        parts = [b"".join(lines[:700]), b"".join(lines[700:1500]), b"".join(lines[1500:])]
        outs = [eng.dedupe(p, inv, reset=(i == 0))[0] for i, p in enumerate(parts)]
        assert b"".join(outs) == want
    for bad, inv in ((A + BADC + B, True), (A + BADC + B, False), (A + b"q\t1\t2\n" + B, False)):
        want, werr = O.dedupe(bad, inv)
        got, info = eng.dedupe(bad, inv, raise_on_error=False)
        assert (info.error.code, info.error.record if werr.code else 0) == (werr.code, werr.record if werr.code else 0) and got == want
    eng.close()


@pytest.mark.gpu
def test_gpu_dedupe_random_streams():
    """Random streams with many repeats, swapped twins and records that fail paf_check (reached or not, depending on what was written
    before them), in one batch and cut into random batches through one context: output, error code and failing record as the oracle's loop."""
    import paffy_amd

    eng = paffy_amd.Engine()
    for seed in range(12):
        rng = random.Random(100 + seed)
        pool = []
        for k in range(60):
            qn, tn = b"q%d" % rng.randrange(6), b"t%d" % rng.randrange(6)
            ql, tl = 1000, 2000
            qs, ts = rng.randrange(0, 900), rng.randrange(0, 1900)
            ln = rng.randrange(1, 90)
            strand = rng.choice([b"+", b"-"])
            pool.append(b"\t".join([qn, b"%d" % ql, b"%d" % qs, b"%d" % (qs + ln), strand, tn, b"%d" % tl, b"%d" % ts, b"%d" % (ts + ln), b"%d" % ln, b"%d" % ln, b"60",
                                     b"cg:Z:%dM" % ln]) + b"\n")
        bad = []  # coordinates paf_check rejects; they parse
        for k in range(3):
            bad.append(b"qb%d\t50\t10\t4\t+\ttb\t200\t0\t3\t3\t3\t60\tcg:Z:3M\n" % k)
        n = rng.choice([50, 400, 3000])
        p_bad = rng.choice([0.0, 0.002, 0.02])
        lines = []
        for _ in range(n):
            l = rng.choice(pool)
            r = rng.random()
            if r < p_bad:
                l = rng.choice(bad)
            if rng.random() < 0.35:
                f = l.rstrip(b"\n").split(b"\t")
                f[0], f[5] = f[5], f[0]
                f[1], f[6] = f[6], f[1]
                f[2], f[7] = f[7], f[2]
                f[3], f[8] = f[8], f[3]
                l = b"\t".join(f) + b"\n"
            lines.append(l)
        data = b"".join(lines)
        for inv in (False, True):
            want, werr = O.dedupe(data, inv)
            got, info = eng.dedupe(data, inv, raise_on_error=False)
            assert info.error.code == werr.code and got == want, (seed, inv)
            if werr.code:
                assert info.error.record == werr.record, (seed, inv)
            # the same stream in random batches through one context
            cuts = sorted(rng.sample(range(1, n), min(n - 1, rng.randrange(1, 6))))
            parts = [b"".join(lines[a:b]) for a, b in zip([0] + cuts, cuts + [n])]
            outs, base, code, rec = [], 0, 0, 0
            for i, part in enumerate(parts):
                o, inf = eng.dedupe(part, inv, reset=(i == 0), raise_on_error=False)
                outs.append(o)
                if inf.error.code:
                    code, rec = inf.error.code, base + inf.error.record
                    break
                base += part.count(b"\n")
            assert b"".join(outs) == want and code == werr.code and (not code or rec == werr.record), (seed, inv, cuts)
    eng.close()


@pytest.mark.gpu
def test_cli_dedupe(human_chimp):
    rng = random.Random(10)
    data = build_dupes(rng, human_chimp[:100000], 1500)
    for args, inv in (([], False), (["-a"], True), (["--checkInverse"], True)):
        p = subprocess.run([PAFFY, "dedupe"] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={"PAFFY_CHUNK_MB": "1", "PATH": "/usr/bin:/bin"})
        assert p.returncode == 0, p.stderr
        assert p.stdout == O.dedupe(data, inv)[0]
    p = subprocess.run([PAFFY, "dedupe", "-a"], input=A + BADC, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and p.stdout == O.dedupe(A)[0]
