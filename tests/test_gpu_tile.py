"""paffy tile on the GPU vs the oracle (levels, visiting order, verbatim cigar, error rules)."""
import hashlib
import json
import os
import random

import pytest

import oracle_lib as O
import synth_lib
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import paffy_amd

    e = paffy_amd.Engine()
    yield e
    e.close()


def check(eng, data):
    want, werr = O.tile(data)
    got, info = eng.tile(data, raise_on_error=False)
    assert info.error.code == werr.code, (info.error.code, werr.code, info.error.record, werr.record)
    if werr.code:
        assert info.error.record == werr.record
    assert got == want
    return got, info


def test_fixture(eng, human_chimp):
    got, info = check(eng, human_chimp)
    with open(os.path.join(GOLDEN, "human_chimp_digests.json")) as fh:
        golden = json.load(fh)["tile"]
    assert hashlib.sha256(got).hexdigest() == golden["sha256"] and info.n_rows == 207


def overlapping_records(rng, n, contigs=3, qlen=5000):
    """Many records piled on a few short query sequences: levels climb, scores tie, strands mix."""
    out = []
    for r in range(n):
        c = rng.randrange(contigs)
        ops, qspan, tspan = [], 0, 0
        for k in range(rng.choice([1, 3, 10, 40])):
            L = rng.choice([1, 5, 30, 200])
            if qspan + L + 20 >= qlen:
                break
            ops.append(f"{L}{rng.choice('M=X')}"); qspan += L; tspan += L
            g = rng.choice([1, 2, 9])
            if rng.random() < 0.5:
                ops.append(f"{g}I"); qspan += g
            else:
                ops.append(f"{g}D"); tspan += g
        ops.append("2M"); qspan += 2; tspan += 2
        qs = rng.randrange(0, qlen - qspan)
        tags = []
        if rng.random() < 0.8:
            tags.append(f"AS:i:{rng.choice([10, 20, 20, 500])}")
        if rng.random() < 0.5:
            tags.append(f"s1:i:{rng.choice([7, 7, 90])}")
        if rng.random() < 0.3:
            tags.append("tp:A:" + rng.choice("PSI"))
        out.append(f"q{c}\t{qlen}\t{qs}\t{qs + qspan}\t{rng.choice('+-')}\tt\t100000\t{1000}\t{1000 + tspan}\t{tspan}\t{tspan}\t60\t"
                   + "\t".join(tags + [f"cg:Z:{''.join(ops)}"]) + "\n")
    return "".join(out).encode()


def test_piled_records_and_ties(eng):
    rng = random.Random(11)
    data = overlapping_records(rng, 1500)
    got, _ = check(eng, data)
    levels = [int(l.split(b"\ttl:i:")[1].split(b"\t")[0]) for l in got.splitlines()]
    assert max(levels) > 20 and min(levels) == 1
    check(eng, overlapping_records(rng, 300, contigs=40, qlen=900))


def test_synthetic_stream(eng):
    check(eng, synth_lib.generate(0x5EED0005, 300, 0, 600))


def test_errors_and_edges(eng):
    ok = b"q\t100\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\tAS:i:9\tcg:Z:5M\n"
    assert eng.tile(b"")[0] == b""
    check(eng, ok)
    check(eng, ok[:-1])
    for bad in (ok.replace(b"\tcg:Z:5M", b""),               # no cg tag: NULL cigar
                ok.replace(b"5M", b"5M1S"),                   # bad cigar character
                ok.replace(b"5M", b"7M"),                     # walks past query_end
                ok.replace(b"q\t100", b"q\t101"),             # same name, other length
                ok.replace(b"\t+\t", b"\t*\t"),                # parse error: first in input order
                ok.replace(b"cg:Z:5M", b"cg:Z:")):            # empty cigar with a non-empty span
        got, info = check(eng, ok + bad + ok)
        assert info.error.code != 0 and got == b""
    # a name longer than the header staging
    check(eng, ok + b"N" * 4000 + b"\t100\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\tAS:i:3\tcg:Z:5M\n")
    # zero aligned bases -> level 32767
    check(eng, b"q\t100\t3\t5\t+\tt\t100\t0\t0\t0\t0\t60\tcg:Z:2I\n")


def kernels_used(eng, data):
    eng.profile(True)
    check(eng, data)
    names = {k for k, (ms, launches) in eng.profile_read().items() if launches > 0}
    eng.profile(False)
    return names


def test_sliced_path_and_fallback(eng):
    """Records spanning several 1 Mi-base slices stay on the sliced path; many distinct levels inside one
    slice or levels beyond the LDS window repeat the batch on the one-workgroup-per-sequence kernel."""
    rng = random.Random(5)
    qlen = 5_000_000
    lines = []
    for r in range(400):
        span = rng.choice([1000, 300_000, 1_048_576, 2_500_000])
        qs = rng.randrange(0, qlen - span)
        a = span // 3
        cig = f"{a}M7D{span - 2 * a}=5I{a - 5}X" if a > 5 else f"{span}M"
        tspan = span - 5 + 7 if a > 5 else span
        lines.append(f"big{r % 2}\t{qlen}\t{qs}\t{qs + span}\t{rng.choice('+-')}\tt\t9000000\t10\t{10 + tspan}\t{span}\t{span}\t60\t"
                     f"AS:i:{rng.randrange(50)}\tcg:Z:{cig}\n")
    data = "".join(lines[:12]).encode()
    used = kernels_used(eng, data)
    assert "k_tile_slices" in used and "k_tile" not in used
    data = "".join(lines).encode()      # deep, ragged pile: dozens of distinct levels in one (record, slice) histogram
    used = kernels_used(eng, data)
    assert "k_tile_slices" in used and "k_tile" not in used
    # a staircase: record k covers [100 k, 100 k + 20000), the last record spans them all and meets 150 distinct levels
    stairs = [f"st\t100000\t{100 * k}\t{100 * k + 20000}\t+\tt\t9000000\t0\t20000\t20000\t20000\t60\tAS:i:{1000 - k}\tcg:Z:20000M\n" for k in range(150)]
    stairs.append("st\t100000\t0\t40000\t-\tt\t9000000\t0\t40000\t40000\t40000\t60\tAS:i:1\tcg:Z:40000M\n")
    used = kernels_used(eng, "".join(stairs).encode())  # more than 128 distinct levels: the exact fallback takes over
    assert "k_tile" in used
    used = kernels_used(eng, "".join(stairs[:100] + stairs[-1:]).encode())  # 100 levels: still sliced
    assert "k_tile_slices" in used and "k_tile" not in used
    # levels beyond the 4096-level window
    one = b"deep\t10\t2\t3\t+\tt\t10\t0\t1\t1\t1\t60\tcg:Z:1M\n"
    used = kernels_used(eng, one * 4200)
    assert "k_tile" in used
