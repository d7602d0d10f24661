"""Shatter row writers on the GPU vs the oracle: records built to reach every branch of the row kernel
(k_emit_rows) and its fallbacks -- piece lengths around the 16/32/48-byte limits, coordinates crossing powers
of ten and multiples of 10^4, long ops, adjacent M ops (two rows per lane), both strands, windows that do not
fit the LDS buffer at full size, sequences of 10^11 bases and more, the register parser's limits."""
import hashlib
import random

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def make_record(rng, qname, tname, qlen, tlen, n_ops, strand, len_choices, adjacent_m=0.0, tags=True, start_near=None):
    ops = []
    qspan = tspan = 0
    last_m = False
    for k in range(n_ops):
        if k % 2 == 0 or (last_m and rng.random() < adjacent_m):
            op = "M"
        else:
            op = rng.choice("ID")
        if last_m and op == "M" and rng.random() >= adjacent_m:
            op = rng.choice("ID")
        L = rng.choice(len_choices)
        ops.append(f"{L}{op}")
        if op != "D":
            qspan += L
        if op != "I":
            tspan += L
        last_m = op == "M"
    if ops[-1][-1] != "M":
        ops.append("3M"); qspan += 3; tspan += 3
    assert qspan < qlen and tspan < tlen
    if start_near is None:
        qs = rng.randrange(0, qlen - qspan)
        ts = rng.randrange(0, tlen - tspan)
    else:  # start just below a boundary so that the rows cross it
        qs = max(0, min(qlen - qspan - 1, start_near - rng.randrange(0, max(1, qspan))))
        ts = max(0, min(tlen - tspan - 1, start_near - rng.randrange(0, max(1, tspan))))
    t = []
    if tags:
        t = [f"tp:A:{rng.choice('PSI')}", f"AS:i:{rng.randrange(10**rng.randrange(1, 8))}", f"s1:i:{rng.randrange(1000)}"]
        if rng.random() < 0.3:
            t += [f"tl:i:{rng.randrange(1, 4)}", f"cn:i:{rng.randrange(100000)}"]
    return (f"{qname}\t{qlen}\t{qs}\t{qs + qspan}\t{strand}\t{tname}\t{tlen}\t{ts}\t{ts + tspan}\t{qspan}\t{qspan}\t{rng.randrange(256)}\t"
            + "\t".join(t + ["cg:Z:" + "".join(ops)]) + "\n")


def run_both(eng, data, pipes):
    import paffy_amd

    for pipe in pipes:
        want, werr = O.run([O.stage(k) for k in pipe], data)
        got, info = eng.run([paffy_amd.stage(k) for k in pipe], data, raise_on_error=False)
        assert info.error.code == werr.code, (pipe, info.error.code, werr.code, info.error.record, werr.record)
        assert len(got) == len(want) and hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest(), pipe


PIPES = ([O.SHATTER], [O.INVERT, O.SHATTER], [O.INVERT, O.TRIM_IDENTITY, O.SHATTER], [O.TRIM_FIXED, O.SHATTER])


@pytest.fixture(scope="module")
def eng():
    import paffy_amd

    e = paffy_amd.Engine()
    yield e
    e.close()


def test_piece_lengths_and_digit_boundaries(eng):
    rng = random.Random(41)
    recs = []
    for name_len in list(range(1, 40)) + [47, 48, 49, 60, 100]:  # lenA / lenB around 16, 32, 48
        for strand in "+-":
            qn = "q" * name_len
            tn = "T" * max(1, 41 - name_len if name_len < 40 else name_len)
            qlen = rng.choice([90000, 99999999, 100000123, 123456789012])
            tlen = rng.choice([150000, 20000000, 999999999, 98765432109])
            recs.append(make_record(rng, qn, tn, qlen, tlen, rng.randrange(1, 700), strand, [1, 2, 9, 10, 40, 99], tags=rng.random() < 0.8))
    # rows crossing 10^4 multiples and powers of ten on both strands
    for near in (10000, 20000, 99990, 100000, 9999999, 10000000, 99999999, 100000000, 999999999, 1000000000, 99999999999):
        for strand in "+-":
            recs.append(make_record(rng, "hs.chr1", "pt.chr1", near * 3 + 12345, near * 3 + 999, 600, strand, [1, 5, 30, 60], start_near=near))
    run_both(eng, "".join(recs).encode(), PIPES)


def test_long_ops_adjacent_matches_and_dense_windows(eng):
    rng = random.Random(42)
    recs = []
    for k in range(40):
        strand = "+-"[k & 1]
        # long ops: windows whose coordinates leave the shared-digit range, L with 3..7 digits
        recs.append(make_record(rng, "hs.chr7", "pt.chr12", 10**11 - 5, 10**11 - 7, rng.randrange(50, 900), strand, [1, 7, 150, 5000, 12345, 2000000]))
        # adjacent M ops: two rows per lane and pair
        recs.append(make_record(rng, "hs.chr7", "pt.chr12", 250000000, 250000000, rng.randrange(50, 900), strand, [1, 3, 25, 120], adjacent_m=0.6))
        # sequences of 10^11 bases and more: general row path
        recs.append(make_record(rng, "hs.chrBig", "pt.chrBig", 10**11 + 17, 3 * 10**12, rng.randrange(50, 400), strand, [1, 3, 25, 120]))
        # long names and many-digit lengths with one-digit rows: the full-size window would overflow the LDS buffer estimate
        recs.append(make_record(rng, "Q" * 45, "T" * 44, 99999999999, 99999999999, rng.randrange(300, 1200), strand, [1, 2], tags=True))
    run_both(eng, "".join(recs).encode(), PIPES)


def test_register_parser_limits(eng):
    """Cigars at the edges of the register parser: 8 KiB of text, 7- and 8-digit numbers, leading zeros, ops without
    digits, text offsets of every alignment, bad characters at every position class."""
    rng = random.Random(43)
    recs = []
    for pad in range(16):  # cigar text offset mod 16 through the name length
        recs.append(make_record(rng, "n" * (pad + 1), "t", 5000000, 5000000, rng.randrange(1, 2500), "+-"[pad & 1], [1, 12, 345]))
    base = "q\t900000000\t0\t{q}\t+\tt\t900000000\t0\t{t}\t1\t1\t60\tcg:Z:{cg}\n"
    for cg, q, t in (("9999999M", 9999999, 9999999), ("10000000M", 10000000, 10000000), ("0000012M", 12, 12), ("00000000012M", 12, 12),
                     ("5M0I5M", 10, 10), ("5MI5M", 10, 10), ("M", 0, 0), ("5M3", 5, 5), ("5M3S2M", 7, 7), ("5m", 5, 5)):
        recs.append(base.format(q=q, t=t, cg=cg))
    for n_ops in (2047, 2731, 2732, 4000):  # around 8 KiB of cigar text (3 bytes per op)
        recs.append(make_record(rng, "hs.chr2", "pt.chr2", 240000000, 240000000, n_ops, "+", [10, 99]))
    data = "".join(recs).encode()
    import paffy_amd

    for line in data.splitlines(keepends=True):  # one by one: error records must not hide the others
        want, werr = O.run([O.stage(O.SHATTER)], line)
        got, info = eng.run([paffy_amd.stage(paffy_amd.SHATTER)], line, raise_on_error=False)
        assert info.error.code == werr.code and got == want, line[:80]
        want, werr = O.run([O.stage(O.INVERT)], line)
        got, info = eng.run([paffy_amd.stage(paffy_amd.INVERT)], line, raise_on_error=False)
        assert info.error.code == werr.code and got == want, line[:80]


def test_fuzz_regressions(eng):
    """Inputs the soak test (tools/fuzz_gpu.py) once caught: case1 = a parent with target_start -1 whose first op is a
    deletion (valid rows, negative base for the row kernel's digit arithmetic)."""
    import glob
    import os

    from conftest import GOLDEN

    for path in sorted(glob.glob(os.path.join(GOLDEN, "fuzz", "*.paf"))):
        with open(path, "rb") as fh:
            data = fh.read()
        run_both(eng, data, ([O.SHATTER], [O.PASS], [O.INVERT, O.SHATTER]))


def test_trim_ends_stage_equals_paf_trim_ends():
    """PAFFY_TRIM_ENDS (the stage behind paf_trim_ends of the per-record API) against the oracle's paf_trim_ends on the
    reference's known answers (tests/paf_unit_test.c:413-457) and on fixture records, both strands, several counts."""
    import os

    import paffy_amd

    eng = paffy_amd.Engine()
    lines = [b"q\t100\t5\t15\t+\tt\t100\t5\t15\t10\t10\t60\tcg:Z:10M\n", b"q\t100\t0\t10\t+\tt\t100\t0\t10\t10\t10\t60\tcg:Z:10M\n",
             b"q\t100\t0\t8\t+\tt\t100\t0\t7\t7\t8\t60\tcg:Z:2M1I5M\n", b"q\t100\t0\t8\t-\tt\t100\t0\t7\t7\t8\t60\tcg:Z:2M1I5M\n",
             b"q\t100\t0\t13\t-\tt\t100\t0\t12\t10\t15\t60\tcg:Z:5M3I2D4=1X\n"]
    with open(os.path.join(os.path.dirname(__file__), "golden", "human_chimp.paf"), "rb") as fh:
        lines += fh.read().splitlines(keepends=True)[:60]
    checked = 0
    for count in (0, 1, 2, 3, 7, 40, 1000):
        want = []
        for ln in lines:
            rc, out = O.trim_ends_line(ln, count)
            want.append(out if rc == 0 else None)
        keep = [ln for ln, w in zip(lines, want) if w is not None]
        got, info = eng.run([paffy_amd.stage_trim_ends(count)], b"".join(keep), raise_on_error=False)
        # the stage also runs paf_check like the command loops: compare the records the oracle's check accepts too
        ok = [w for w in want if w is not None]
        if info.error.code == 0:
            assert got == b"".join(ok), count
            checked += len(ok)
    assert checked > 200
    eng.close()


def test_longer_first_store_level_and_its_way_back():
    """A context sizes cigars of up to 20 000 bytes at the first store level once a batch has shown that none of them holds more ops than
    that store (8 192); a cigar that then does (two bytes per op) goes through the arena class -- same bytes as the oracle's -- and the
    context returns to the safe bound. Every batch of the sequence is compared with the oracle."""
    import paffy_amd

    rng = random.Random(77)
    eng = paffy_amd.Engine()
    pipe_g = [paffy_amd.stage(paffy_amd.INVERT), paffy_amd.stage(paffy_amd.TRIM_IDENTITY), paffy_amd.stage(paffy_amd.SHATTER)]
    pipe_o = [O.stage(O.INVERT), O.stage(O.TRIM_IDENTITY), O.stage(O.SHATTER)]

    def usual(n):  # three-digit lengths: 4 bytes per op, cigars of 17 000 - 19 000 bytes among them (4 300 - 4 700 ops)
        return "".join(make_record(rng, f"q{k}", f"t{k % 3}", 10**9, 10**9, rng.choice([40, 900, 4300, 4700]), rng.choice("+-"), [100, 250, 999]) for k in range(n))

    def dense(n_ops):  # one-digit lengths: 2 bytes per op
        return make_record(rng, "qd", "td", 10**9, 10**9, n_ops, "+", [1, 2, 9])

    batches = [usual(40), usual(40), usual(20) + dense(9500) + usual(20), usual(40), dense(9000) + usual(10), usual(30)]
    for i, text in enumerate(batches):
        data = text.encode()
        want, werr = O.run(pipe_o, data)
        got, info = eng.run(pipe_g, data, raise_on_error=False)
        assert info.error.code == werr.code == 0, i
        assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest(), i
    eng.close()


def test_one_wave_writers_on_both_sides_of_their_op_limit(eng):
    """round-3 advice: PAFFY_ROWS_MAX_OPS went from 16 384 to 32 768 ops after the soak. Cigars of 16 385, 20 000, 32 768 and 32 769 view ops
    (and their neighbours) through a shatter pipe and a line-writing pipe the RECORD kernels size (the fixed trim keeps them off the flat
    pass) and through the flat pass's own pipes: the one-wave writers k_emit_rows / k_emit_line below the limit, the four-wave writers
    and the segments of the flat pass above it, byte for byte against the oracle."""
    rng = random.Random(45)
    recs = []
    for k, n_ops in enumerate([16383, 16385, 20001, 32767, 32769, 32771, 40001]):  # make_record ends on an M op: odd counts
        recs.append(make_record(rng, "hs.chr5", "pt.chr8", 240000000, 240000000, n_ops, "+-"[k & 1], [1, 7, 30, 120]))
    data = "".join(recs).encode()
    run_both(eng, data, ([O.TRIM_FIXED, O.SHATTER], [O.TRIM_FIXED], [O.REMOVE_MISMATCHES, O.INVERT], [O.SHATTER], [O.INVERT], [O.INVERT, O.TRIM_IDENTITY, O.SHATTER]))
    import re

    # the view's op count decides: the records really stand on both sides of 32 768
    counts = [len(re.findall(rb"[MID]", r.split(b"cg:Z:")[1])) for r in data.splitlines()]
    assert min(counts) < 16384 < sorted(counts)[1] and any(c == 32767 for c in counts) and any(c == 32769 for c in counts)
