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


def check(eng, data, **kw):
    want, werr = O.tile(data)
    got, info = eng.tile(data, raise_on_error=False, **kw)
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


def kernels_used(eng, data, **kw):
    eng.profile(True)
    check(eng, data, **kw)
    names = {k for k, (ms, launches) in eng.profile_read().items() if launches > 0}
    eng.profile(False)
    return names


def test_slices_piles_and_wide_level_spreads(eng):
    """Records spanning many 32 Ki-base slices, deep ragged piles, a staircase whose last record meets more distinct levels than
    the LDS histogram tells apart (the exact window-by-window path), levels in the thousands: always the slice walk, always exact."""
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
    used = kernels_used(eng, "".join(lines[:12]).encode())
    assert "k_cov_walk" in used and "k_cov_bitmap" in used and "k_cov_merge" in used
    check(eng, "".join(lines).encode())  # deep, ragged pile: dozens of distinct levels in one (record, slice) histogram
    # a staircase: record k covers [10 k, 10 k + 20000), the last record spans them all and meets 1500 distinct levels (> 1024)
    stairs = [f"st\t100000\t{10 * k}\t{10 * k + 20000}\t+\tt\t9000000\t0\t20000\t20000\t20000\t60\tAS:i:{5000 - k}\tcg:Z:20000M\n" for k in range(1500)]
    stairs.append("st\t100000\t0\t40000\t-\tt\t9000000\t0\t40000\t40000\t40000\t60\tAS:i:1\tcg:Z:40000M\n")
    got, _ = check(eng, "".join(stairs).encode())
    assert got.splitlines()[-1].split(b"\ttl:i:")[1].split(b"\t")[0] == b"751"  # the oracle's value: the last record meets levels 2 .. 1501
    check(eng, "".join(stairs[:100] + stairs[-1:]).encode())
    # a pile 4200 deep on one base
    one = b"deep\t10\t2\t3\t+\tt\t10\t0\t1\t1\t1\t60\tcg:Z:1M\n"
    got, _ = check(eng, one * 4200)
    assert got.splitlines()[-1].split(b"\ttl:i:")[1].split(b"\t")[0] == b"4200"


def test_records_behind_a_record_whose_levels_overflow_the_lds_histogram(eng):
    """Round 3, found by the soak (tools/fuzz_gpu.py tile): a record whose new counts spread over more than the 256 levels of the LDS
    histogram is re-histogrammed window by window, and that path cleared 1 024 of the buffer's 1 056 words -- the record two places
    later in the same slice then read the stale words as levels 100..127 (mod 128) of its own and got a median one to four levels low.
    The fixture is the input the soak failed on (1 500 records on one 70 000-base contig, 445-fold coverage, 28 wrong levels between
    228 and 279); the constructed case puts many small records behind a wide one on a 300-step staircase."""
    with open(os.path.join(GOLDEN, "fuzz", "tile_r3_fail.paf"), "rb") as fh:
        data = fh.read()
    got, info = check(eng, data)
    assert info.n_rows == 1500
    rng = random.Random(11)
    stairs = [f"st\t100000\t{10 * k}\t{10 * k + 20000}\t+\tt\t9000000\t0\t20000\t20000\t20000\t60\tAS:i:{5000 - k}\tcg:Z:20000M\n" for k in range(300)]
    stairs.append("st\t100000\t0\t24000\t-\tt\t9000000\t0\t24000\t24000\t24000\t60\tAS:i:1000\tcg:Z:24000M\n")  # meets levels 2 .. 301
    for m in range(400):  # behind it, in the same slice: small records all over the staircase, the wide one in between again
        a = rng.randrange(0, 22000)
        n = rng.choice([8, 40, 300, 1500])
        stairs.append(f"st\t100000\t{a}\t{a + n}\t{rng.choice('+-')}\tt\t9000000\t5\t{5 + n}\t{n}\t{n}\t60\tAS:i:{900 - m}\tcg:Z:{n}M\n")
        if m % 50 == 49:
            stairs.append(f"st\t100000\t0\t24000\t+\tt\t9000000\t0\t24000\t24000\t24000\t60\tAS:i:{900 - m}\tcg:Z:24000M\n")
    check(eng, "".join(stairs).encode())


def test_counters_saturate_at_32766(eng):
    """impl/paf.c:700: a counter stops at INT16_MAX - 1, and so do the levels."""
    one = b"deep\t10\t2\t4\t+\tt\t10\t0\t2\t2\t2\t60\tcg:Z:2M\n"
    got, _ = check(eng, one * 32800)
    levels = [int(l.split(b"\ttl:i:")[1].split(b"\t")[0]) for l in got.splitlines()]
    assert levels[0] == 1 and levels[32765] == 32766 and levels[-1] == 32766


def test_batches_and_chunks_give_the_same_bytes(eng, human_chimp):
    """An input held as several text batches, and a walk cut into chunks of entries (bitmap budget), equal the one-batch run."""
    data = human_chimp + synth_lib.generate(0x5EED0005, 300, 0, 500) + overlapping_records(random.Random(3), 800)
    want, _ = check(eng, data)
    for batch_bytes in (1 << 20, 200_000, 70_000):
        got, info = eng.tile(data, batch_bytes=batch_bytes)
        assert got == want and info.n_records == data.count(b"\n")
    os.environ["PAFFY_COV_BITMAP_MB"] = "1"
    try:
        assert eng.tile(data)[0] == want
        assert eng.tile(data, batch_bytes=300_000)[0] == want
        check(eng, overlapping_records(random.Random(4), 3000, contigs=2, qlen=70_000))
    finally:
        del os.environ["PAFFY_COV_BITMAP_MB"]
    # an error in a later batch is reported with its record number over all batches
    bad = data + b"q\t100\t0\t5\t*\tt\t100\t0\t5\t5\t5\t60\tcg:Z:5M\n"
    got, info = eng.tile(bad, raise_on_error=False, batch_bytes=200_000)
    assert got == b"" and info.error.code == 2 and info.error.record == data.count(b"\n")


def test_emit_in_pieces(eng, human_chimp):
    """paffy_hip_emit_lines: the output drained through a small staging buffer equals the one-shot emit."""
    want, _ = O.tile(human_chimp)
    bufs = [(eng.to_device(p), len(p)) for p in eng.split_lines(human_chimp, 400_000)]  # must outlive the emits: the lines are written from them
    info = eng.tile_batches(bufs)
    assert info.error.code == 0 and info.out_bytes == len(want)
    stage = eng.alloc_out(200_000)
    got, first = b"", 0
    while first < info.n_rows:
        n = min(7, info.n_rows - first)
        while True:  # as many lines as fit
            try:
                nbytes = eng.emit_lines(first, n, stage)
                break
            except RuntimeError:
                n -= 1
                assert n >= 1
        eng.sync()
        got += bytes(stage[:nbytes].cpu().numpy().tobytes())
        first += n
    assert got == want
    keys = eng.tile_keys(info.n_rows).cpu().numpy()
    assert keys.shape == (207, 5) and int(keys[:, 3].sum()) == len(want)
    assert [int(x) for x in keys[:, 4]] == [int(l.split(b"\ttl:i:")[1].split(b"\t")[0]) for l in want.splitlines()]


def test_more_than_2_gib_at_depth(eng):
    """`paffy tile` on an input beyond the 2 GiB a single batch can hold -- 500 000 records of mean 2k ops on ONE
    query contig (~150x coverage) in four text batches -- byte-equal to the oracle."""
    import hashlib

    import torch

    bufs, host = [], []
    for b in range(4):
        buf, nbytes = eng.synth(0x5EED0005, 2048, b * 125_000, 125_000, n_contigs=1)
        bufs.append((buf, nbytes))
        host.append(bytes(buf[:nbytes].cpu().numpy().tobytes()))
    data = b"".join(host)
    del host
    assert len(data) > (1 << 31)
    info = eng.tile_batches(bufs)
    assert info.error.code == 0 and info.n_records == 500_000
    d_out = eng.alloc_out(info.out_bytes)
    eng.emit(d_out)
    eng.sync()
    got = hashlib.sha256(d_out[: info.out_bytes].cpu().numpy().tobytes()).hexdigest()
    levels = eng.tile_keys(info.n_rows)[:, 4]
    assert int(levels.max()) > 50  # deep: the median level of the last records is in the hundreds
    del d_out, bufs
    torch.cuda.empty_cache()
    want, err = O.tile(data)
    assert err.code == 0 and len(want) == info.out_bytes
    assert hashlib.sha256(want).hexdigest() == got


def _collision_pair():
    """Two different 13-character names with the same 64-bit FNV-1a hash (tools/fnv_collide.c made them; salt 0 of cov_name_hash)."""
    from paffy_amd import shard

    with open(os.path.join(GOLDEN, "fnv_collision.txt")) as fh:
        a, b, h = fh.read().split()
    assert a != b and shard.name_hash(a.encode()) == shard.name_hash(b.encode()) == int(h, 16)
    return a, b


def test_names_that_share_a_hash_keep_their_own_counters(eng):
    """The reference keys its coverage arrays by the name STRING (impl/paf_tile.c:160-161, impl/paf.c:675-688). Two query names
    with equal hashes must not share counters: sequences of different lengths (shared counters would trip the length assert,
    impl/paf.c:685) piled with records whose levels depend on what was counted before them."""
    a, b = _collision_pair()
    rng = random.Random(11)
    recs = []
    for r in range(400):
        name, qlen = (a, 3000) if rng.random() < 0.5 else (b, 4100)
        L = rng.choice([20, 150, 600])
        qs = rng.randrange(0, qlen - L - 5)
        recs.append(f"{name}\t{qlen}\t{qs}\t{qs + L}\t{rng.choice('+-')}\tt{r % 3}\t900000\t{10 + r}\t{10 + r + L}\t{L}\t{L}\t60\tAS:i:{rng.choice([5, 80, 900])}\tcg:Z:{L}M\n")
    data = "".join(recs).encode()
    got, info = check(eng, data)
    assert info.error.code == 0 and info.n_rows == 400
    levels = {ln.split(b"\ttl:i:")[1].split(b"\t")[0] for ln in got.splitlines()}
    assert len(levels) > 2  # the piles are deep enough for the levels to matter
    # to_bed walks the same counters: one run list per name
    want_bed, werr = O.to_bed(data)
    got_bed, binfo = eng.to_bed(data, raise_on_error=False)
    assert werr.code == 0 and binfo.error.code == 0 and got_bed == want_bed
    assert a.encode() in got_bed and b.encode() in got_bed
