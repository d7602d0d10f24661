"""Parity of the HIP path (through the C-ABI) with the CPU oracle: bit-exact output bytes."""
import hashlib
import json
import os

import pytest

import oracle_lib as O
import synth_lib
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

S = O.stage
PIPES = {
    "pass": [S(O.PASS)],
    "invert": [S(O.INVERT)],
    "invert|invert": [S(O.INVERT), S(O.INVERT)],
    "trim": [S(O.TRIM_IDENTITY)],
    "trim -r 0.95": [S(O.TRIM_IDENTITY, 0.95, 1.0)],
    "trim -r 0 -t 0.3": [S(O.TRIM_IDENTITY, 0.0, 0.3)],
    "trim -f -t 0.1": [S(O.TRIM_FIXED, 0.05, 0.1)],
    "trim -f -t 0": [S(O.TRIM_FIXED, 0.05, 0.0)],
    "trim -f -t 1": [S(O.TRIM_FIXED, 0.05, 1.0)],
    "shatter": [S(O.SHATTER)],
    "invert|trim|shatter": [S(O.INVERT), S(O.TRIM_IDENTITY), S(O.SHATTER)],
    "trim -f|invert|shatter": [S(O.TRIM_FIXED, 0.05, 0.2), S(O.INVERT), S(O.SHATTER)],
}


@pytest.fixture(scope="module")
def eng():
    import paffy_amd

    e = paffy_amd.Engine()
    yield e
    e.close()


def gpu_stages(stages):
    import paffy_amd

    return [paffy_amd.Stage(s.kind, s.p0, s.p1) for s in stages]


def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i, a[max(0, i - 60): i + 40], b[max(0, i - 60): i + 40]
    return n, a[n - 60: n + 40], b[n - 60: n + 40]


def check(eng, stages, data, name=""):
    want, werr = O.run(stages, data)
    got, info = eng.run(gpu_stages(stages), data, raise_on_error=False)
    assert info.error.code == werr.code, (name, info.error.code, werr.code, info.error.record, werr.record)
    if werr.code:
        assert (info.error.record, info.error.stage) == (werr.record, werr.stage), name
    assert len(got) == info.out_bytes
    assert got == want, (name, len(got), len(want), first_diff(got, want))
    return got, info


def kat_line(q, qlen, qs, qe, strand, t, tlen, ts, te, nm, nb, mq, cigar=None, tags=""):
    s = f"{q}\t{qlen}\t{qs}\t{qe}\t{strand}\t{t}\t{tlen}\t{ts}\t{te}\t{nm}\t{nb}\t{mq}"
    if tags:
        s += "\t" + tags
    if cigar is not None:
        s += "\tcg:Z:" + cigar
    return (s + "\n").encode()


KAT_RECORDS = [
    kat_line("query1", 100, 0, 50, "+", "target1", 200, 10, 60, 50, 50, 255),
    kat_line("q1", 100, 0, 8, "+", "t1", 200, 0, 7, 8, 10, 60, "5M3I2D"),
    kat_line("q1", 100, 0, 50, "+", "t1", 200, 0, 50, 50, 50, 60, None, "tp:A:P\tAS:i:42\ttl:i:2\tcn:i:5\ts1:i:100"),
    kat_line("q1", 100, 0, 50, "-", "t1", 200, 0, 50, 50, 50, 60, None, "s1:i:100\tcn:i:5\ttl:i:2\tAS:i:-42\ttp:A:I"),
    kat_line("query", 100, 10, 18, "+", "target", 200, 20, 27, 8, 10, 60, "5M3I2D"),
    kat_line("query", 100, 10, 18, "-", "target", 200, 20, 25, 5, 8, 60, "5M3I"),
    kat_line("q", 100, 5, 15, "+", "t", 100, 5, 15, 10, 10, 60, "10M"),
    kat_line("q", 100, 0, 8, "+", "t", 100, 0, 7, 7, 8, 60, "2M1I5M"),
    kat_line("q", 100, 0, 8, "-", "t", 100, 0, 7, 7, 8, 60, "2M1I5M"),
    kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M"),
    kat_line("q", 100, 0, 7, "+", "t", 100, 0, 9, 7, 9, 60, "3M2D4M"),
    kat_line("q", 100, 0, 7, "-", "t", 100, 0, 9, 7, 9, 60, "3M2D4M"),
    kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M", "tl:i:2\ts1:i:99\tcn:i:4\tAS:i:7\tNM:i:3\tde:f:0.01"),
    b"q\t\t100\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\t\tcg:Z:\n",
    kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, None, "AS:i:-12\tcg:Z:9M\tcg:Z:5M"),
    kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "0000000000000000000000000000000000000000005M"),
    kat_line("q", 10 ** 15, 10 ** 14, 10 ** 14 + 5, "+", "t", 10 ** 18, 123456789012345678, 123456789012345683, 5, 5, 60, "5M", "AS:i:-9223372036854775807"),
]
KAT_EQX = [  # records with =/X ops: not shatterable (assert), fine for invert / trim
    kat_line("q", 9, 0, 9, "-", "t", 9, 0, 9, 5, 9, 60, "2X5=2X"),
    kat_line("q", 9, 0, 9, "+", "t", 9, 0, 9, 5, 9, 60, "2X5=2X"),
    kat_line("q", 9, 0, 7, "-", "t", 9, 0, 7, 5, 7, 60, "2X5="),
    kat_line("q", 100, 0, 6, "+", "t", 100, 0, 7, 5, 8, 60, "3=2X1I2D"),
    kat_line("q", 100, 0, 13, "+", "t", 100, 0, 12, 10, 15, 60, "5M3I2D4=1X"),
]


def test_known_answer_records(eng):
    """The records of /root/reference/tests/paf_unit_test.c through every fused pipe."""
    data = b"".join(KAT_RECORDS)
    for name, stages in PIPES.items():
        check(eng, stages, data, name)
        for rec in KAT_RECORDS:
            check(eng, stages, rec, name)
    for name in ("pass", "invert", "trim", "trim -r 0 -t 0.3", "trim -f -t 0.1", "trim -f -t 1", "invert|invert"):
        check(eng, PIPES[name], b"".join(KAT_EQX), name)
    for thr in (0.0, 1.0):
        for rec in KAT_EQX:
            check(eng, [S(O.TRIM_IDENTITY, thr, 1.0)], rec, "kat trim")


def test_fixture_digests(eng, human_chimp):
    """tests/human_chimp.paf: 207 records, 1..20663 ops, both LDS and arena classes."""
    with open(os.path.join(GOLDEN, "human_chimp_digests.json")) as fh:
        golden = json.load(fh)
    for name, stages in PIPES.items():
        got, info = check(eng, stages, human_chimp, name)
        assert info.n_records == 207
        if name in golden:
            assert hashlib.sha256(got).hexdigest() == golden[name]["sha256"], name
            assert info.n_rows == golden[name]["lines"]


def test_chained_commands_equal_fused(eng, human_chimp):
    import paffy_amd

    st = gpu_stages(PIPES["invert|trim|shatter"])
    assert eng.run_chain(st, human_chimp) == eng.run(st, human_chimp)[0]
    # cfg 1 of BASELINE.json: shatter | invert, run as two commands
    want = O.run([S(O.SHATTER), S(O.INVERT)], human_chimp)[0]
    assert eng.run_chain([paffy_amd.stage(paffy_amd.SHATTER), paffy_amd.stage(paffy_amd.INVERT)], human_chimp) == want


@pytest.mark.parametrize("seed,mean_ops,n", [(0x5EED0002, 512, 700), (0x5EED0003, 2048, 300), (0x5EED0009, 3, 3000), (0x5EED000A, 9000, 40)])
def test_synthetic_records(eng, seed, mean_ops, n):
    data = synth_lib.generate(seed, mean_ops, 0, n)
    for name in ("invert|trim|shatter", "shatter", "invert", "trim", "trim -f -t 0.1"):
        check(eng, PIPES[name], data, name)


def test_device_generator_matches_host(eng):
    for seed, mean_ops, r0, n in [(0x5EED0003, 2048, 0, 200), (0x5EED0002, 512, 12345, 500), (7, 1, 0, 1000)]:
        buf, nbytes = eng.synth(seed, mean_ops, r0, n)
        assert bytes(buf[:nbytes].cpu().numpy().tobytes()) == synth_lib.generate(seed, mean_ops, r0, n)


def test_edges(eng):
    ok = kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M")
    for stages in (PIPES["pass"], PIPES["shatter"], PIPES["invert|trim|shatter"]):
        assert eng.run(gpu_stages(stages), b"")[0] == b""
        check(eng, stages, ok[:-1])           # final line without '\n'
        check(eng, stages, ok * 3 + ok[:-1])
    long_name = ("Q" * 700).encode()
    rec = long_name + b"\t100\t0\t5\t-\t" + b"T" * 333 + b"\t100\t0\t5\t5\t5\t60\tcg:Z:2M1D2M1I\n"
    rec = rec.replace(b"\t0\t5\t-", b"\t0\t5\t-", 1)
    for name in ("pass", "invert"):
        check(eng, PIPES[name], rec + ok, name)
    # names far longer than the LDS staging of the line pieces: the direct-to-HBM paths
    huge = b"Q" * 5000 + b"\t900\t3\t12\t-\t" + b"T" * 3000 + b"\t800\t5\t13\t7\t9\t60\ttp:A:S\tcg:Z:3M1D2M1I3M\n"
    for name in ("pass", "invert", "shatter", "invert|trim|shatter", "trim -f -t 0.1"):
        check(eng, PIPES[name], ok + huge + ok + huge, name)
    # many tiny records: every alignment of a line start modulo 16
    tiny = b"".join(kat_line("c%d" % i, 1000 + i, i, i + 5, "+-"[i & 1], "d", 2000, 7, 12, 5, 5, i, "2M1I2M1D1M" if i % 3 else "5M") for i in range(500))
    for name in ("pass", "invert", "shatter", "invert|trim|shatter", "trim -f -t 0.1"):
        check(eng, PIPES[name], tiny, name)


def test_header_lines_of_many_tokens(eng):
    """The header kernel takes 32 separators of a line per round (round 3): lines with dozens of tags, the recognised ones spread over
    several rounds and repeated (the last one of a kind wins), runs of tabs inside and between rounds, a bad tp / strand token in the
    second round (tokens behind it are ignored), short tokens, a cg tag that is not the last token."""
    import random

    rng = random.Random(3)
    junk = ["NM:i:%d" % i for i in range(40)] + ["de:f:0.01", "zd:Z:x", "ab", "x", "tp:B:P", "AS:f:1.0", "s1:Z:9", ""]
    lines = []
    for k in range(300):
        toks = [rng.choice(junk) for _ in range(rng.choice([0, 5, 19, 20, 21, 31, 32, 33, 64, 90]))]
        for kind in ("AS:i:%d", "tl:i:%d", "cn:i:%d", "s1:i:%d"):
            for _ in range(rng.choice([0, 1, 2, 3])):
                toks.insert(rng.randrange(len(toks) + 1), kind % rng.randrange(-5, 500))
        if rng.random() < 0.6:
            toks.insert(rng.randrange(len(toks) + 1), "tp:A:" + rng.choice("PSI"))
        cigs = ["cg:Z:5M"] * rng.choice([1, 1, 2]) + (["cg:Z:2M1I2M1D1M"] if rng.random() < 0.3 else [])
        for c in cigs:
            toks.insert(rng.randrange(len(toks) + 1), c)
        last_cg = next(t for t in reversed(toks) if t.startswith("cg:Z:"))  # the last cg tag is the record's cigar (impl/paf.c:193-198)
        qe, te = (6, 6) if last_cg == "cg:Z:2M1I2M1D1M" else (5, 5)
        head = ["q%d" % k, "100", "0", str(qe), rng.choice("+-"), "t", "100", "0", str(te), "5", "5", "60"]
        sep = lambda: "\t" * rng.choice([1, 1, 1, 2, 3])  # noqa: E731
        line = head[0]
        for t in head[1:] + toks:
            line += sep() + t
        lines.append(line.encode() + b"\n")
    data = b"".join(lines)
    for name in ("pass", "invert", "shatter"):
        check(eng, PIPES[name], data, name)
    # a token the reference aborts on, in the second round of its line: everything in front is written, the codes agree
    bad_tp = b"q\t100\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\t" + b"\t".join(b"NM:i:%d" % i for i in range(30)) + b"\ttp:A:X\tAS:i:7\tcg:Z:5M\n"
    check(eng, PIPES["pass"], lines[0] + lines[1] + bad_tp + lines[2], "bad tp in round 2")


def test_integer_columns_at_every_alignment(eng):
    """str_to_int64 (impl/paf.c:37-48) in the header kernel reads a column's digits from aligned 8-byte words (parse_i64_words): columns
    of 1 to 19 digits and a sign starting at every byte alignment, digits followed by other characters (the number stops there), a sign
    alone, a '+', columns of more than twenty characters (the byte-by-byte form), tag values of every length -- and the text's last
    line without a newline, its last column ending with the buffer (no word beyond the text is read)."""
    import random

    rng = random.Random(11)
    lines = []
    for k in range(400):
        qn = "q" * rng.randrange(1, 18)  # shifts the alignment of everything behind it
        nd = rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 18, 19])
        big = rng.randrange(10 ** (nd - 1), 10 ** nd) if nd > 1 else rng.randrange(1, 10)
        big = min(big, 2 ** 62)
        qlen = big + 10
        qs = big
        score = rng.choice([0, 5, -5, 123456789012, -(2 ** 62), 2 ** 62, 99999999, 100000000])
        s1 = rng.choice(["7", "-7", "0012", "12ab", "-", "+5", "", "1" * 25, "-" + "9" * 22, "4x", "18446744073709551615"])
        tl = rng.choice(["1", "2", "30"])
        cn = rng.choice(["0", "65535", "4294967296"])
        lines.append(f"{qn}\t{qlen}\t{qs}\t{qs + 5}\t{rng.choice('+-')}\tt\t{10 ** rng.randrange(3, 18)}\t7\t12\t5\t5\t{rng.choice([0, 9, 60, 255])}"
                     f"\tAS:i:{score}\ttl:i:{tl}\tcn:i:{cn}\tcg:Z:5M\ts1:i:{s1}\n".encode())
    data = b"".join(lines)
    for name in ("pass", "invert", "shatter"):
        check(eng, PIPES[name], data, name)
        for cut in (1, 2, 9):  # the last column ends with the buffer, at several alignments
            check(eng, PIPES[name], data[:-cut] if data[-cut - 1:-cut].isdigit() else data[:-1], name)
    tail = b"q\t100\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\tcg:Z:5M\ts1:i:1234567"
    for n in range(1, 8):
        check(eng, PIPES["pass"], lines[0] + tail[:len(tail) - 7 + n], "tag value of %d digits at the end of the text" % n)


def test_a_stats_stage_is_a_process_of_its_own(eng):
    """`paffy view -s` behind another command is a second process: what reaches it has been through `paf_write | paf_parse`, so a record whose
    cigar a fixed trim has emptied arrives without its cg tag (impl/paf.c:71-73: cigar_parse("") is NULL) -- the output of a pipe with a stats
    stage equals the oracle's with a pass stage in its place, not the pipe without it (found by the soak, round 4)."""
    import paffy_amd

    rec = kat_line("q", 100, 0, 5, "+", "t", 100, 0, 4, 4, 5, 60, "2M1I2M", "tp:A:S\tAS:i:35114")
    keep = kat_line("q2", 100, 0, 50, "-", "t", 100, 0, 48, 48, 50, 60, "20M1I10M1I18M")
    data = rec + keep
    trim_all = S(O.TRIM_FIXED, 0.05, 1.0)
    want_plain, _ = O.run([trim_all], data)
    want_pass, _ = O.run([trim_all, S(O.PASS)], data)
    assert want_plain != want_pass and b"cg:Z:\n" in want_plain  # the emptied cigar keeps its tag only without the boundary
    got, info = eng.run([paffy_amd.Stage(O.TRIM_FIXED, 0.05, 1.0), paffy_amd.stage(paffy_amd.STATS)], data, raise_on_error=False)
    assert info.error.code == 0 and got == want_pass
    got, info = eng.run([paffy_amd.Stage(O.TRIM_FIXED, 0.05, 1.0)], data, raise_on_error=False)
    assert info.error.code == 0 and got == want_plain


def test_one_pass_separator_index_follows_the_density(eng):
    """Round 3: from the second batch of a context on the separator index is ONE pass over the text whose buffers are sized by the batch
    before (k_sep_index: tile tickets, decoupled look-back); a denser batch is indexed again with exact sizes. One engine, batches of
    rising and falling separator density, lines without a cigar, runs of tabs, an unterminated last line, more tiles than a look-back
    window holds, a batch of one line -- all against the oracle."""
    import paffy_amd

    e = paffy_amd.Engine()
    try:
        ok = kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M")
        sparse = synth_lib.generate(0x5EED0003, 2048, 0, 300)           # 24 separators per 5.4 KB
        dense = b"".join(kat_line("c%d" % i, 1000 + i, i, i + 5, "+-"[i & 1], "d", 2000, 7, 12, 5, 5, i, None, "AS:i:%d\ttp:A:P" % i) for i in range(40000))  # a separator every 4 bytes
        tabs = b"".join(b"q%d\t\t\t100\t\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\t\t\tcg:Z:5M\n" % i for i in range(3000))
        big = synth_lib.generate(0x5EED0009, 3, 0, 120000)               # about 9 MB: 140 tiles of 64 KiB, short records
        seq = [sparse, dense, sparse, tabs, big, ok, dense[:-1], big + ok[:-1], ok * 7, sparse]
        for k, data in enumerate(seq):
            for name in ("pass", "invert"):
                got, info = check(e, PIPES[name], data, "%s #%d" % (name, k))
                assert info.n_records == data.count(b"\n") + (0 if data.endswith(b"\n") else 1)
        # the same batches through a fresh engine each (two-pass index) give the same bytes: already implied by the oracle
    finally:
        e.close()


def test_error_paths(eng):
    ok = kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M")
    cases = [
        (b"q\t100\t0\t5\t*\tt\t100\t0\t5\t5\t5\t60\n", PIPES["pass"]),
        (kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M", "tp:A:i"), PIPES["pass"]),
        (kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M2S"), PIPES["pass"]),
        (kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M3"), PIPES["pass"]),
        (kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5="), PIPES["shatter"]),
        (kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "0M5M"), PIPES["shatter"]),
        (kat_line("q", 100, 0, 6, "+", "t", 100, 0, 5, 5, 5, 60, "5M"), PIPES["shatter"]),
        (kat_line("q", 100, 0, 6, "+", "t", 100, 0, 5, 5, 5, 60, "5M"), PIPES["invert"]),
        (kat_line("q", 100, 98, 103, "+", "t", 100, 0, 5, 5, 5, 60, "5M"), PIPES["shatter"]),
        (kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60), PIPES["trim"]),
        (kat_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60), PIPES["trim -f -t 0.1"]),
        (b"q\t100\t0\t5\t+\tt\t100\t0\n", PIPES["pass"]),
        (b"\n", PIPES["pass"]),
    ]
    for bad, stages in cases:
        got, info = check(eng, stages, ok + bad + ok)
        assert info.error.code != 0 and info.error.record == 1
        assert got == O.run(stages, ok)[0]


def test_a_stream_closed_after_its_context_was_destroyed():
    """round-3 advice: paffy_hip_stream_close left the slots' device buffers with a context that could be gone. A context now knows its open
    streams and detaches them when it is destroyed; closing such a stream frees its own buffers."""
    import ctypes as C

    import paffy_amd
    from paffy_amd import engine

    L = engine.lib()
    ctx, st = C.c_void_p(), C.c_void_p()
    assert L.paffy_hip_create(C.byref(ctx), 0) == 0
    stages = (engine.Stage * 1)(paffy_amd.stage(paffy_amd.INVERT))
    assert L.paffy_hip_stream_open(ctx, stages, 1, 1 << 20, 1 << 20, C.byref(st)) == 0
    L.paffy_hip_destroy(ctx)
    L.paffy_hip_stream_close(st)  # must neither crash nor touch the freed context
    # and the usual order still leaves the buffers with the context for its next stream
    assert L.paffy_hip_create(C.byref(ctx), 0) == 0
    for _ in range(2):
        assert L.paffy_hip_stream_open(ctx, stages, 1, 1 << 20, 1 << 20, C.byref(st)) == 0
        L.paffy_hip_stream_close(st)
    L.paffy_hip_destroy(ctx)
