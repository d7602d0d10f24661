"""The flat sizing pass (paffy_amd/csrc/flat_kernel.h) against the oracle: the lean pipes are parsed in chunks of 1 KiB pieces of cigar
text whatever record they belong to, and sized per record from the pieces' summaries. Built to reach its edges: cigars that start at
every offset inside a 1 KiB tile and ops whose digits straddle tile and chunk boundaries, what the pass leaves to the record kernels
(numbers of five digits or with leading zeros, lengths of 8 192 and more, zero lengths, = and X ops, bad characters, a cigar that ends
in digits, failing checks) next to records it keeps, records of more than 64 pieces (their summaries are scanned in place), trims that
cut deep into a record, and the very long records of SURVEY section 5 (about 100 000 and 1 000 000 ops; impl/paf.c:398-403)."""
import hashlib
import os
import random

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

LEAN_PIPES = ([O.SHATTER], [O.INVERT, O.TRIM_IDENTITY, O.SHATTER], [O.INVERT], [O.TRIM_IDENTITY], [O.INVERT, O.TRIM_IDENTITY],
              [O.TRIM_IDENTITY, O.SHATTER], [O.INVERT, O.INVERT, O.SHATTER])


STATS = []  # (pipe, (records left to the record kernels, counts per reason)) of every run, in order


def kept_all(n_runs=None):
    """the flat pass took the last runs' records itself: the run exercised the code under test"""
    runs = STATS[-n_runs:] if n_runs else STATS
    return all(left == 0 for _, _, (left, _) in runs)


@pytest.fixture(scope="module")
def eng():
    import paffy_amd

    e = paffy_amd.Engine()
    yield e
    e.close()
    if os.environ.get("PAFFY_FLAT_STATS_OUT"):  # who sized what, run by run (diagnostics)
        import json

        with open(os.environ["PAFFY_FLAT_STATS_OUT"], "w") as fh:
            json.dump([{"test": t, "pipe": p, "left": left, "why": why} for t, p, (left, why) in STATS], fh)


def record(ops, strand="+", qname="hs.chr3", tname="pt.chr9", qlen=250_000_000, tlen=240_000_000, qs=None, ts=None, tags="tp:A:P\tAS:i:777\ts1:i:42",
           rng=None, cigar=None):
    """ops: list of (length, letter); the coordinates are consistent with them unless qs / ts say otherwise"""
    qspan = sum(L for L, c in ops if c in "M=XI")
    tspan = sum(L for L, c in ops if c in "M=XD")
    if qs is None:
        qs = rng.randrange(0, qlen - qspan) if rng else 1000
    if ts is None:
        ts = rng.randrange(0, tlen - tspan) if rng else 2000
    text = cigar if cigar is not None else "".join(f"{L}{c}" for L, c in ops)
    return f"{qname}\t{qlen}\t{qs}\t{qs + qspan}\t{strand}\t{tname}\t{tlen}\t{ts}\t{ts + tspan}\t{qspan}\t{qspan + tspan}\t60\t{tags}\tcg:Z:{text}\n"


def random_ops(rng, n, lens=(1, 2, 3, 7, 12, 40, 99, 150, 1234), indel=(1, 2, 3, 9, 25)):
    ops = []
    for k in range(n):
        if k % 2 == 0:
            ops.append((rng.choice(lens), "M"))
        else:
            ops.append((rng.choice(indel), rng.choice("ID")))
    if ops[-1][1] != "M":
        ops.append((rng.choice(lens), "M"))
    return ops


def run_both(eng, data, pipes=LEAN_PIPES, params=None):
    import paffy_amd

    for pipe in pipes:
        ost = [O.stage(k, *(params or {}).get(k, ())) for k in pipe]
        gst = [paffy_amd.stage(k, *(params or {}).get(k, ())) for k in pipe]
        want, werr = O.run(ost, data)
        got, info = eng.run(gst, data, raise_on_error=False)
        STATS.append((os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0], tuple(pipe), eng.flat_stats()))
        assert info.error.code == werr.code, (pipe, info.error.code, werr.code, info.error.record, werr.record)
        assert len(got) == len(want) and hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest(), pipe


def test_every_tile_offset_and_boundary_straddles(eng):
    """the cigar's first byte at every offset mod 1024 (names of growing length in front of it), cigars of one piece to a few chunks"""
    rng = random.Random(4101)
    recs = []
    for pad in range(0, 1100, 7):
        strand = "+-"[pad & 1]
        n = rng.choice([1, 2, 3, 60, 300, 340, 700, 1400, 2800, 5000])
        recs.append(record(random_ops(rng, n), strand, qname="q" + "x" * (pad % 37), tags="tp:A:S\tAS:i:5\tzz:Z:" + "p" * pad, rng=rng))
    run_both(eng, "".join(recs).encode())


def test_numbers_across_tile_and_chunk_boundaries(eng):
    """four-digit lengths everywhere: many numbers have their digits on both sides of a 1 KiB / 4 KiB boundary"""
    rng = random.Random(4102)
    recs = []
    for k in range(60):
        ops = random_ops(rng, rng.choice([400, 900, 2500, 6000]), lens=(1000, 1001, 4321, 8191, 9, 77), indel=(1000, 2222, 8191, 3))
        recs.append(record(ops, "+-"[k & 1], qlen=2_000_000_000, tlen=2_000_000_000, rng=rng))
    run_both(eng, "".join(recs).encode())


def test_what_the_flat_pass_leaves_to_the_record_kernels(eng):
    """irregular records between regular ones: the outputs (and the errors) are the record kernels'"""
    rng = random.Random(4103)
    good = lambda n=200: record(random_ops(rng, n), rng.choice("+-"), rng=rng)  # noqa: E731
    base = random_ops(rng, 300)
    cases = []
    for cigar_edit in ("8192M", "9999M", "10000M", "123456M", "007M", "00M", "0M", "M", "5=", "7X", "3N", "5M7", "12m", "1234567M", "12345678M", "5M5M"):
        ops = list(base)
        text = "".join(f"{L}{c}" for L, c in ops[:100]) + cigar_edit + "".join(f"{L}{c}" for L, c in ops[100:])
        # coordinates follow the parsed meaning where there is one: let the oracle decide what happens
        cases.append(text)
    recs_ok, recs_err = [], []
    for text in cases:
        import re

        parsed = [(int(a) if a else 0, b) for a, b in re.findall(r"(\d*)([A-Za-z=])", text)]
        ops = [(L, c) for L, c in parsed if c in "MIDX="]
        r = record(ops, rng.choice("+-"), cigar=text, rng=rng)
        recs_ok.append(r)
    # one stream per irregular record, surrounded by regular ones: an error ends the output where the reference's would
    for r in recs_ok:
        data = (good() + good(1500) + r + good() + good(3000)).encode()
        run_both(eng, data, pipes=([O.SHATTER], [O.INVERT, O.TRIM_IDENTITY, O.SHATTER], [O.INVERT], [O.TRIM_IDENTITY]))
    # failing paf_check: coordinates that do not match the cigar
    bad = record(random_ops(rng, 500), "+", qs=5, ts=7, rng=rng).replace("\t5\t", "\t6\t", 1)
    run_both(eng, (good() + bad + good()).encode(), pipes=([O.SHATTER], [O.INVERT, O.SHATTER], [O.TRIM_IDENTITY]))
    # no cigar at all, an empty cigar
    nocg = "a\t100\t1\t50\t+\tb\t100\t1\t50\t49\t49\t60\ttp:A:P\n"
    empty = "a\t100\t1\t50\t+\tb\t100\t1\t50\t49\t49\t60\ttp:A:P\tcg:Z:\n"
    run_both(eng, (good() + nocg + good() + empty + good()).encode(), pipes=([O.INVERT], [O.TRIM_IDENTITY], [O.INVERT, O.TRIM_IDENTITY]))


def test_rows_whose_digit_counts_change_inside_the_record(eng):
    """start and end coordinates with different digit counts: the flat pass leaves the record, its neighbours stay"""
    rng = random.Random(4104)
    recs = []
    for near in (10, 100, 1000, 99990, 1000000, 99999990, 100000000):
        for strand in "+-":
            ops = random_ops(rng, 400)
            recs.append(record(ops, strand, qs=max(0, near - 50), ts=rng.randrange(1000, 5000), rng=rng))
            recs.append(record(ops, strand, qs=rng.randrange(1000, 5000), ts=max(0, near - 50), rng=rng))
            recs.append(record(random_ops(rng, 900), strand, rng=rng))
    run_both(eng, "".join(recs).encode(), pipes=([O.SHATTER], [O.INVERT, O.TRIM_IDENTITY, O.SHATTER]))


def test_trims_that_cut_deep(eng):
    """noisy ends of many pieces: the identity trim drops thousands of ops on either end, on both strands; thresholds as the CLI passes them"""
    rng = random.Random(4105)
    recs = []
    for k in range(80):
        noisy_front = [(rng.choice([1, 2, 3]), "M") if i % 2 == 0 else (rng.choice([5, 9, 30]), rng.choice("ID")) for i in range(rng.choice([0, 10, 400, 3000, 9000]))]
        noisy_back = [(rng.choice([5, 9, 30]), rng.choice("ID")) if i % 2 == 0 else (rng.choice([1, 2, 3]), "M") for i in range(rng.choice([0, 10, 400, 3000, 9000]))]
        core = random_ops(rng, rng.choice([50, 3000, 20000]), lens=(40, 99, 150, 1234), indel=(1, 2))
        ops = noisy_front + ([(1, "M")] if noisy_front and noisy_front[-1][1] != "M" and core[0][1] != "M" else []) + core
        ops = ops + [o for o in noisy_back] + [(7, "M")]
        # no two indels or two M side by side requirement in the reference: any order of ops is a valid cigar
        recs.append(record(ops, "+-"[k & 1], rng=rng))
    data = "".join(recs).encode()
    run_both(eng, data, pipes=([O.TRIM_IDENTITY], [O.INVERT, O.TRIM_IDENTITY, O.SHATTER], [O.TRIM_IDENTITY, O.SHATTER]))
    for p0, p1 in ((0.05, 1.0), (0.2, 0.3), (0.5, 0.1), (0.01, 0.02)):
        run_both(eng, data, pipes=([O.TRIM_IDENTITY], [O.TRIM_IDENTITY, O.SHATTER]), params={O.TRIM_IDENTITY: (p0, p1)})


def test_records_of_more_than_64_pieces(eng):
    """70 KB to 600 KB of cigar text: the summaries are scanned in place in HBM, the four-wave writers take the records above 32 768 ops"""
    rng = random.Random(4106)
    recs = []
    for k, n in enumerate([22000, 23000, 32700, 32768, 32769, 33000, 40000, 65536, 70001, 150000]):
        recs.append(record(random_ops(rng, n, lens=(1, 5, 30, 60, 110)), "+-"[k & 1], rng=rng))
        recs.append(record(random_ops(rng, 300), "+-"[k & 1], rng=rng))
    run_both(eng, "".join(recs).encode())


@pytest.mark.parametrize("n_ops", [100_001, 1_000_001])
def test_very_long_records(eng, n_ops):
    """real lines reach megabytes -- one record of about 100 000 ops and one of about 1 000 000 (the reference grows its
    buffers by doubling, impl/paf.c:398-403), both strands, long and short names, through the stream pipes"""
    rng = random.Random(4107 + n_ops)
    recs = []
    for strand, qn, tn in (("+", "q", "t"), ("-", "hs.chr" + "Q" * 40, "pt.chr" + "T" * 44)):
        recs.append(record(random_ops(rng, 50), strand, rng=rng))
        recs.append(record(random_ops(rng, n_ops, lens=(1, 5, 30, 60, 110)), strand, qname=qn, tname=tn, rng=rng))
    data = "".join(recs).encode()
    run_both(eng, data, pipes=([O.SHATTER], [O.INVERT, O.TRIM_IDENTITY, O.SHATTER], [O.INVERT], [O.TRIM_IDENTITY], [O.TRIM_FIXED]))


def test_no_flat_switch_gives_the_same_bytes(eng, human_chimp):
    """PAFFY_NO_FLAT=1 (read once per process: a second engine in a child process) sizes everything with the record kernels"""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r); import paffy_amd; e = paffy_amd.Engine();"
            "d = open(%r, 'rb').read();"
            "got, _ = e.run([paffy_amd.stage(paffy_amd.INVERT), paffy_amd.stage(paffy_amd.TRIM_IDENTITY), paffy_amd.stage(paffy_amd.SHATTER)], d);"
            "print(hashlib.sha256(got).hexdigest())") % (os.path.dirname(here), here, os.path.join(here, "golden", "human_chimp.paf"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PAFFY_NO_FLAT="1"), capture_output=True, text=True, check=True).stdout.strip()
    got, _ = eng.run([__import__("paffy_amd").stage(k) for k in (O.INVERT, O.TRIM_IDENTITY, O.SHATTER)], human_chimp)
    assert out == hashlib.sha256(got).hexdigest()


def test_row_pieces_of_the_sizing_pass(eng):
    """The constant pieces of a record's rows (name and length columns, mapq and tags) are put together by the sizing pass, one lane per
    record (lane_row_pieces, flat_kernel.h), and copied by the row writer: names of every length the one-wave row writer takes (pieces
    of at most 48 bytes; longer ones take the general writer), every tag present or absent, negative and many-digit values, both
    strands, with and without an invert in front (the names swap), next to each other in one batch."""
    rng = random.Random(77)
    lines = []
    tag_sets = ["", "tp:A:P", "tp:A:S\tAS:i:-5", "AS:i:2147483646\ts1:i:9", "tl:i:3", "tl:i:1\tcn:i:12\ts1:i:-7", "tp:A:I\tAS:i:0\ttl:i:2\tcn:i:4000000000\ts1:i:1",
                "AS:i:-2147483648", "cn:i:0"]
    for k in range(1200):
        nq, nt = rng.choice((1, 2, 7, 15, 16, 17, 31, 32, 33, 40, 43, 44, 45, 46, 60)), rng.choice((1, 3, 12, 13, 16, 28, 29, 36, 37, 38, 39, 40, 41, 55))
        qname = "".join(rng.choice("abcXYZ_.|0123456789") for _ in range(nq))
        tname = "".join(rng.choice("abcXYZ_.|0123456789") for _ in range(nt))
        qlen, tlen = rng.choice((5000, 99_999, 100_000, 4_000_000_000, 250_000_000)), rng.choice((7000, 9_999_999, 10_000_000, 5_000_000_000))
        line = record(random_ops(rng, rng.choice((1, 3, 20, 90)), lens=(1, 5, 30), indel=(1, 4)), strand=rng.choice("+-"), qname=qname, tname=tname, qlen=qlen, tlen=tlen,
                      tags=rng.choice(tag_sets), rng=rng)
        mapq = rng.choice(("0", "7", "60", "255"))
        cols = line.split("\t")
        cols[11] = mapq
        lines.append("\t".join(c for c in cols if c != ""))

    def fits(line):  # both ways round (the invert swaps the names): name, tab, length, tab <= 48 and tab, strand, tab, name, tab, length, tab <= 48
        c = line.rstrip("\n").split("\t")
        tags = [t for t in c[12:-1] if not t.startswith("s1:")]
        if any(t.startswith("tl:") for t in tags) and not any(t.startswith("tp:") for t in tags):
            tags.append("tp:A:P")  # a tile level makes a type (impl/paf.c:343-348)
        len_c = 1 + len(c[11]) + sum(1 + len(t) for t in tags) + 7 + 6  # \t mapq tags \ts1:i:0 \tcg:Z:
        first_ok = min(len(c[0]) + len(c[1]), len(c[5]) + len(c[6])) + 2 >= 16  # the one-wave row writer wants a first piece of a whole 16 bytes
        return first_ok and len_c <= 48 and all(len(a) + len(b) + 5 <= 48 for a, b in ((c[0], c[1]), (c[5], c[6]), (c[0], c[6]), (c[5], c[1])))

    pipes = ([O.SHATTER], [O.INVERT, O.SHATTER], [O.INVERT, O.TRIM_IDENTITY, O.SHATTER])
    short = [l for l in lines if fits(l)]
    assert 100 < len(short) < len(lines)
    run_both(eng, "".join(short).encode(), pipes=pipes)
    assert kept_all(3)
    run_both(eng, "".join(lines).encode(), pipes=pipes)  # the longer names among them: those records take the general row writer


def test_stats_stage_from_the_summaries(eng):
    """paf_stats_calc (impl/paf.c:236-260; `paffy view -s`) of every record as a stage of a pipe the flat pass takes: the six sums come from
    the pieces' summaries (the I ops counted where a shatter pipe counts digits) -- of the whole record, of what an identity trim leaves of it
    (the cut ops summed by a walk), of an inverted record (inserts and deletes swap). Records with = and X ops go to the record kernels."""
    import paffy_amd

    rng = random.Random(5)
    lines = []
    for k in range(900):
        n = rng.choice((1, 2, 3, 9, 40, 150, 700, 2500))
        lines.append(record(random_ops(rng, n), strand=rng.choice("+-"), rng=rng))
    lines.append(record(random_ops(rng, 30_000), strand="-", rng=rng))  # more than 64 pieces
    plain = "".join(lines).encode()
    eqx = plain + record([(5, "="), (2, "X"), (3, "I"), (4, "="), (1, "D"), (6, "M")], rng=rng).encode() * 3

    def oracle_sums(out):
        acc = [0] * 6
        for ln in out.splitlines():
            f = ln.split(b"cg:Z:")
            if len(f) > 1:
                acc = O.cigar_stats(f[1].split(b"\t")[0].decode(), acc, zero=False)
        return acc

    for data, all_flat in ((plain, True), (eqx, False)):
        buf = eng.to_device(data)
        for pipe in ([], [O.INVERT], [O.TRIM_IDENTITY], [O.INVERT, O.TRIM_IDENTITY], [O.INVERT, O.INVERT]):
            want, werr = O.run([O.stage(k) for k in pipe] or [O.stage(O.PASS)], data)
            assert werr.code == 0
            info = eng.plan([paffy_amd.stage(k) for k in pipe] + [paffy_amd.stage(paffy_amd.STATS)], buf, len(data))
            assert info.error.code == 0 and list(eng.plan_stats()) == oracle_sums(want), pipe
            left, why = eng.flat_stats()
            assert (left == 0) == all_flat, (pipe, left, why)


def test_fixed_trim_as_the_last_stage(eng):
    """`paffy trim -f` (paf_trim_end_fraction + paf_trim_ends, impl/paf.c:578-598) as the last stage of a pipe the flat pass takes: the wave
    kernel finds the op either cut stops at (flat_find_aligned) and shortens it. Every fraction from nothing to everything, cuts that fall
    on an op boundary (the indels behind it go too), in the record's only op, in the same op from both ends, behind an invert and an
    identity trim, = and X ops (aligned bases like M), short and very long records (pieces in registers / scanned in HBM)."""
    rng = random.Random(21)
    lines = []
    for k in range(500):
        n = rng.choice((1, 1, 2, 3, 4, 9, 40, 150, 700, 2500))
        lines.append(record(random_ops(rng, n, lens=(1, 2, 3, 10, 11, 99, 100, 101, 999, 1000, 1001), indel=(1, 2, 3, 10)), strand=rng.choice("+-"), rng=rng))
    lines.append(record(random_ops(rng, 30_000), strand="-", rng=rng))
    lines.append(record([(10, "M")], rng=rng))
    lines.append(record([(10, "M"), (3, "I"), (10, "M")], strand="-", rng=rng))
    lines.append(record([(4, "M"), (2, "D"), (2, "I"), (4, "M"), (1, "I"), (4, "M")], rng=rng))
    lines.append(record([(5, "="), (2, "X"), (3, "I"), (4, "="), (1, "D"), (6, "M"), (1, "X")], rng=rng))
    data = "".join(lines).encode()
    for frac in (0.0, 0.05, 0.1, 0.3333, 0.5, 0.9, 1.0):
        params = {O.TRIM_FIXED: (0.05, frac)}
        run_both(eng, data, pipes=([O.TRIM_FIXED], [O.INVERT, O.TRIM_FIXED], [O.TRIM_IDENTITY, O.TRIM_FIXED], [O.INVERT, O.TRIM_IDENTITY, O.PASS, O.TRIM_FIXED]), params=params)
    left = [x for _, _, (x, _) in STATS[-28:]]
    assert all(x == 0 for x in left[:24]), left  # fractions up to 0.9: every record stays with the flat pass
    assert all(0 < x < 300 for x in left[24:]), left  # 1.0: the records it empties go to the record kernels (they write a cg tag without ops)
