"""Pins the CPU oracle to the reference's own known-answer tests.

Every case restates one test of /root/reference/tests/paf_unit_test.c (lines cited) at the
text level: the record the reference test builds with make_paf()/parse_str() is written as a
PAF line, pushed through the oracle, and compared with the line paf_write would give for the
fields the reference test asserts (SURVEY.md Appendix B grammar: AS:i:<n> is always written).
"""
import oracle_lib as O

S = O.stage
M, I, D, EQ, X = 0, 1, 2, 3, 4


def line(q, qlen, qs, qe, strand, t, tlen, ts, te, nm, nb, mq, cigar=None, tags=""):
    s = f"{q}\t{qlen}\t{qs}\t{qe}\t{strand}\t{t}\t{tlen}\t{ts}\t{te}\t{nm}\t{nb}\t{mq}"
    if tags:
        s += "\t" + tags
    if cigar is not None:
        s += "\tcg:Z:" + cigar
    return (s + "\n").encode()


def out_line(q, qlen, qs, qe, strand, t, tlen, ts, te, nm, nb, mq, cigar=None, tags="AS:i:0"):
    return line(q, qlen, qs, qe, strand, t, tlen, ts, te, nm, nb, mq, cigar, tags)


def run1(stages, data, seqs=None):
    out, err = O.run(stages, data, seqs)
    assert err.code == 0, (err.code, err.stage, err.record)
    return out


# ---- 1/2. cigar parsing, paf_unit_test.c:51-106 ----
def test_cigar_parse():
    assert O.cigar_parse("") == -1  # NULL
    assert O.cigar_parse("10M") == [(M, 10)]
    assert O.cigar_parse("5M3I2D4=1X") == [(M, 5), (I, 3), (D, 2), (EQ, 4), (X, 1)]
    assert O.cigar_parse("1000000M") == [(M, 1000000)]
    assert O.cigar_parse("3M2I") == [(M, 3), (I, 2)]


# ---- 3. paf_parse, paf_unit_test.c:110-182 ----
def test_parse_minimal():
    src = b"query1\t100\t0\t50\t+\ttarget1\t200\t10\t60\t50\t50\t255"
    assert run1([S(O.PASS)], src) == src + b"\tAS:i:0\n"  # no cigar, no cg tag


def test_parse_with_cigar_and_tags():
    src = line("q1", 100, 0, 8, "+", "t1", 200, 0, 7, 8, 10, 60, "5M3I2D")
    assert run1([S(O.PASS)], src) == out_line("q1", 100, 0, 8, "+", "t1", 200, 0, 7, 8, 10, 60, "5M3I2D")
    src = line("q1", 100, 0, 50, "+", "t1", 200, 0, 50, 50, 50, 60, None, "tp:A:P\tAS:i:42\ttl:i:2\tcn:i:5\ts1:i:100")
    assert run1([S(O.PASS)], src) == src
    # tag order on output is fixed whatever the input order (impl/paf.c:343-385)
    shuffled = line("q1", 100, 0, 50, "+", "t1", 200, 0, 50, 50, 50, 60, None, "s1:i:100\tcn:i:5\ttl:i:2\tAS:i:42\ttp:A:P")
    assert run1([S(O.PASS)], shuffled) == src


def test_parse_strand():
    for st in "+-":
        src = line("q1", 100, 0, 50, st, "t1", 200, 0, 50, 50, 50, 60)
        assert run1([S(O.PASS)], src).split(b"\t")[4] == st.encode()


# ---- 4. print(parse(print(parse(x)))) == print(parse(x)), paf_unit_test.c:186-230 ----
def test_roundtrip_idempotent():
    for src in (b"query1\t100\t0\t50\t+\ttarget1\t200\t10\t60\t50\t50\t255\n",
                line("q1", 100, 0, 8, "+", "t1", 200, 0, 7, 8, 10, 60, "5M3I2D")):
        once = run1([S(O.PASS)], src)
        assert run1([S(O.PASS)], once) == once


# ---- 5. file I/O, paf_unit_test.c:234-291 ----
def test_read_three_records_then_eof():
    src = (b"q1\t100\t0\t50\t+\tt1\t200\t0\t50\t50\t50\t60\n"
           b"q2\t200\t10\t60\t-\tt2\t300\t20\t70\t50\t50\t30\n"
           b"q3\t150\t5\t55\t+\tt3\t250\t15\t65\t50\t50\t40\n")
    out = run1([S(O.PASS)], src).splitlines()
    assert len(out) == 3
    assert [l.split(b"\t")[0] for l in out] == [b"q1", b"q2", b"q3"]
    assert out[1].split(b"\t")[4] == b"-" and out[2].split(b"\t")[2] == b"5"
    # a final line without '\n' is still a record and is written with one (impl/paf.c:213,387)
    assert run1([S(O.PASS)], src[:-1]).splitlines() == out


# ---- 6. paf_stats_calc, paf_unit_test.c:295-330 ----
def test_stats():
    assert O.cigar_stats("10M") == [10, 0, 0, 0, 0, 0]
    assert O.cigar_stats("3=2X1I2D") == [3, 2, 1, 1, 1, 2]
    acc = O.cigar_stats("5M", O.cigar_stats("5M", zero=False), zero=False)
    assert acc[0] == 10
    assert O.cigar_stats("5M", acc, zero=True)[0] == 5


# ---- 7. paf_invert, paf_unit_test.c:334-393 ----
def test_invert():
    src = line("query", 100, 10, 18, "+", "target", 200, 20, 27, 8, 10, 60, "5M3I2D")
    assert run1([S(O.INVERT)], src) == out_line("target", 200, 20, 27, "+", "query", 100, 10, 18, 8, 10, 60, "5M3D2I")
    src = line("query", 100, 10, 18, "-", "target", 200, 20, 25, 5, 8, 60, "5M3I")
    assert run1([S(O.INVERT)], src) == out_line("target", 200, 20, 25, "-", "query", 100, 10, 18, 5, 8, 60, "3D5M")
    src = line("query", 100, 10, 18, "+", "target", 200, 20, 27, 8, 10, 60, "5M3I2D")
    assert run1([S(O.INVERT), S(O.INVERT)], src) == run1([S(O.PASS)], src)


# ---- 8. aligned bases, paf_unit_test.c:397-403 ----
def test_aligned_bases():
    assert O.aligned_bases("5M3I2D4=1X") == 10


# ---- 9. trimming, paf_unit_test.c:407-456 ----
def test_trim_ends():
    rc, out = O.trim_ends_line(line("q", 100, 5, 15, "+", "t", 100, 5, 15, 10, 10, 60, "10M"), 0)
    assert rc == 0 and out == out_line("q", 100, 5, 15, "+", "t", 100, 5, 15, 10, 10, 60, "10M")
    rc, out = O.trim_ends_line(line("q", 100, 0, 10, "+", "t", 100, 0, 10, 10, 10, 60, "10M"), 2)
    assert rc == 0 and out == out_line("q", 100, 2, 8, "+", "t", 100, 2, 8, 10, 10, 60, "6M")
    rc, out = O.trim_ends_line(line("q", 100, 0, 8, "+", "t", 100, 0, 7, 7, 8, 60, "2M1I5M"), 3)
    assert rc == 0 and out == out_line("q", 100, 4, 5, "+", "t", 100, 3, 4, 7, 8, 60, "1M")


def test_trim_end_fraction():
    src = line("q", 100, 0, 10, "+", "t", 100, 0, 10, 10, 10, 60, "10M")
    assert run1([S(O.TRIM_FIXED, 0.05, 0.4)], src) == out_line("q", 100, 2, 8, "+", "t", 100, 2, 8, 10, 10, 60, "6M")


# ---- 10. shatter, paf_unit_test.c:460-521 (+ Appendix A-12: children carry s1:i:0) ----
def test_shatter():
    kid = lambda qs, qe, st, ts, te, L: out_line("q", 100, qs, qe, st, "t", 100, ts, te, L, L, 60, f"{L}M", "AS:i:0\ts1:i:0")
    assert run1([S(O.SHATTER)], line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M")) == kid(0, 5, "+", 0, 5, 5)
    assert run1([S(O.SHATTER)], line("q", 100, 0, 7, "+", "t", 100, 0, 9, 7, 9, 60, "3M2D4M")) == \
        kid(0, 3, "+", 0, 3, 3) + kid(3, 7, "+", 5, 9, 4)
    assert run1([S(O.SHATTER)], line("q", 100, 0, 7, "-", "t", 100, 0, 9, 7, 9, 60, "3M2D4M")) == \
        kid(4, 7, "-", 0, 3, 3) + kid(0, 4, "-", 5, 9, 4)


# ---- 11. mismatch encoding, paf_unit_test.c:525-572 ----
def test_encode_and_remove_mismatches():
    enc = lambda q, t, nm, L: run1([S(O.ADD_MISMATCHES)], line("q", L, 0, L, "+", "t", L, 0, L, nm, L, 60, f"{L}M"), {"q": q, "t": t})
    assert enc("AAAAA", "AAAAA", 5, 5) == out_line("q", 5, 0, 5, "+", "t", 5, 0, 5, 5, 5, 60, "5=")
    assert enc("AAAAA", "CCCCC", 0, 5) == out_line("q", 5, 0, 5, "+", "t", 5, 0, 5, 0, 5, 60, "5X")
    assert enc("AATT", "AACC", 2, 4) == out_line("q", 4, 0, 4, "+", "t", 4, 0, 4, 2, 4, 60, "2=2X")
    src = line("q", 100, 0, 6, "+", "t", 100, 0, 5, 5, 6, 60, "3=2X1I")
    assert run1([S(O.REMOVE_MISMATCHES)], src) == out_line("q", 100, 0, 6, "+", "t", 100, 0, 5, 5, 6, 60, "5M1I")


# ---- 12. coverage counters, paf_unit_test.c:576-603 ----
# O.coverage_counts drives counts_for() + bump_counts() of oracle/paf_oracle.c: the very functions po_tile and po_to_bed call
# (one restatement of get_alignment_count_array / increase_alignment_level_counts, impl/paf.c:675-709), so this reference test
# pins what `tile` and `to_bed` run.
def test_coverage_counts():
    rec = line("seq1", 10, 2, 5, "+", "t", 100, 0, 3, 3, 3, 60, "3M")
    applied, counts = O.coverage_counts(rec, "seq1", 10)
    assert applied == 1 and counts == [0, 0, 1, 1, 1, 0, 0, 0, 0, 0]
    # "second call with same query name returns the same array" (:589-591): two records of one name share the counters
    applied, counts = O.coverage_counts(rec + rec, "seq1", 10)
    assert applied == 2 and counts == [0, 0, 2, 2, 2, 0, 0, 0, 0, 0]
    # ... and the tile / to_bed entry points show the same counters: to_bed prints them as runs, tile's level 1 then 2
    bed, err = O.to_bed(rec + rec)
    assert err.code == 0 and bed == b"seq1 0 2 0\nseq1 2 5 2\nseq1 5 10 0\n"
    tiled, err = O.tile(rec + rec)
    assert err.code == 0 and [l.split(b"\ttl:i:")[1][:1] for l in tiled.splitlines()] == [b"1", b"2"]
    # the length assert of impl/paf.c:685 and the end assert of :708
    assert O.coverage_counts(rec + line("seq1", 11, 2, 5, "+", "t", 100, 0, 3, 3, 3, 60, "3M"), "seq1", 10)[0] == -2
    assert O.coverage_counts(line("seq1", 10, 2, 6, "+", "t", 100, 0, 3, 3, 3, 60, "3M"), "seq1", 10)[0] == -1


# ---- 3b. cigar kept as a string (paf_parse(.., false)), paf_unit_test.c:148-160: tile and dedupe read records this way and write the
# cigar text back verbatim (impl/paf.c:381-385) -- even text cigar_parse would reject or normalise
def test_parse_cigar_string_mode():
    src = line("q1", 100, 0, 8, "+", "t1", 200, 0, 7, 8, 10, 60, "5M3I2D")
    out, err = O.dedupe(src)
    assert err.code == 0 and out == out_line("q1", 100, 0, 8, "+", "t1", 200, 0, 7, 8, 10, 60, "5M3I2D")
    odd = line("q1", 100, 0, 8, "+", "t1", 200, 0, 7, 8, 10, 60, "005M3I2D")
    out, err = O.dedupe(odd)
    assert err.code == 0 and out == out_line("q1", 100, 0, 8, "+", "t1", 200, 0, 7, 8, 10, 60, "005M3I2D")


# ---- 5b. read_pafs / write_pafs list round trip, paf_unit_test.c:269-291: three records in, the same three out, in order
def test_read_write_pafs_list():
    src = (line("q1", 100, 0, 50, "+", "t1", 200, 0, 50, 50, 50, 60) + line("q2", 200, 10, 60, "-", "t2", 300, 20, 70, 50, 50, 30) +
           line("q3", 150, 5, 55, "+", "t3", 250, 15, 65, 50, 50, 40))
    out, err = O.dedupe(src)  # read_pafs(.., 0)-style parse of every line, write in input order
    assert err.code == 0 and [l.split(b"\t")[0] for l in out.splitlines()] == [b"q1", b"q2", b"q3"]


# Reference tests of paf_unit_test.c:735-778 NOT restated here, and why:
#   test_decode_fasta_header (:607-616), test_cmp_intervals (:618-632)  -- FASTA-header helpers of dechunk/upconvert, outside SURVEY 8
#       (cmp_intervals is restated for the C API in tests/c/paf_api_kat.c)
#   test_paf_pretty_print_basic (:691-701) -- only asserts "output is non-empty"; the stats line is checked in tests/test_view_stats.py
# Every other test of the suite has its restatement above (numbers 1-12, 14, 16).


# ---- 14. paf_trim_unreliable_tails, paf_unit_test.c:634-687 ----
def test_trim_unreliable_tails():
    src = line("q", 9, 0, 9, "-", "t", 9, 0, 9, 5, 9, 60, "2X5=2X")
    assert run1([S(O.TRIM_IDENTITY, 0.0, 1.0)], src) == out_line("q", 9, 2, 7, "-", "t", 9, 2, 7, 5, 9, 60, "5=")
    src = line("q", 9, 0, 9, "+", "t", 9, 0, 9, 5, 9, 60, "2X5=2X")
    assert run1([S(O.TRIM_IDENTITY, 1.0, 1.0)], src) == out_line("q", 9, 0, 9, "+", "t", 9, 0, 9, 5, 9, 60, "2X5=2X")
    src = line("q", 9, 0, 7, "-", "t", 9, 0, 7, 5, 7, 60, "2X5=")
    assert run1([S(O.TRIM_IDENTITY, 0.0, 1.0)], src) == out_line("q", 9, 0, 5, "-", "t", 9, 2, 7, 5, 7, 60, "5=")
    # a '+' record has its prefix trimmed twice and its suffix never
    src = line("q", 9, 0, 9, "+", "t", 9, 0, 9, 5, 9, 60, "2X5=2X")
    assert run1([S(O.TRIM_IDENTITY, 0.0, 1.0)], src) == out_line("q", 9, 2, 9, "+", "t", 9, 2, 9, 5, 9, 60, "5=2X")


# ---- 16. paf_check positive path, paf_unit_test.c:705-731 (invert's driver calls paf_check) ----
def test_check_valid_records():
    for src in (line("q", 100, 0, 50, "+", "t", 200, 10, 60, 50, 50, 60),
                line("q", 100, 0, 50, "-", "t", 200, 10, 60, 50, 50, 60),
                line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5="),
                line("q", 100, 0, 6, "+", "t", 100, 0, 7, 5, 8, 60, "3=2X1I2D")):
        _, err = O.run([S(O.INVERT), S(O.INVERT)], src)
        assert err.code == 0


# ---- error behaviour read off the reference sources ----
def test_error_paths():
    ok = line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M")
    cases = [
        (b"q\t100\t0\t5\t*\tt\t100\t0\t5\t5\t5\t60\n", [S(O.PASS)], 2, 1),            # bad strand -> st_errAbort
        (line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M", "tp:A:i"), [S(O.PASS)], 3, 134),  # tp assert
        (line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M2S"), [S(O.PASS)], 4, 1),  # bad cigar char
        (line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M3"), [S(O.PASS)], 4, 1),   # trailing digits
        (line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5="), [S(O.SHATTER)], 12, 134),
        (line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "0M5M"), [S(O.SHATTER)], 11, 134),
        (line("q", 100, 0, 6, "+", "t", 100, 0, 5, 5, 5, 60, "5M"), [S(O.SHATTER)], 13, 134),
        (line("q", 100, 0, 6, "+", "t", 100, 0, 5, 5, 5, 60, "5M"), [S(O.INVERT)], 10, 1),
        (b"q\t100\t0\t5\t+\tt\t100\t0\n", [S(O.PASS)], 1, 139),
        (b"\n", [S(O.PASS)], 1, 139),
    ]
    for src, stages, code, status in cases:
        out, err = O.run(stages, ok + src + ok)
        assert err.code == code and err.record == 1, (src, err.code)
        assert O.exit_status(err.code) == status
        assert out == O.run(stages, ok)[0]  # the records before the failing one were written


def test_writer_quirks():
    # every unknown tag is dropped, AS:i:0 appears when AS is absent (Appendix A-4, A-8)
    src = line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M", "NM:i:3\tms:i:9\tde:f:0.01\tzd:i:3")
    assert run1([S(O.PASS)], src) == out_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M")
    # consecutive tabs collapse (strtok_r) and an empty cg:Z: yields no cg tag (A-1, A-6)
    src = b"q\t\t100\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\t\tcg:Z:\n"
    assert run1([S(O.PASS)], src) == out_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60)
    # tl without tp synthesises tp from the level (A-9); shatter children inherit tp/AS/tl/cn, not s1 (A-12)
    src = line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M", "tl:i:2\ts1:i:99\tcn:i:4\tAS:i:7")
    assert run1([S(O.SHATTER)], src) == out_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M", "tp:A:S\tAS:i:7\ttl:i:2\tcn:i:4\ts1:i:0")
    # negative and duplicate tags: last cg wins (A-7)
    src = line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, None, "AS:i:-12\tcg:Z:9M\tcg:Z:5M")
    assert run1([S(O.PASS)], src) == out_line("q", 100, 0, 5, "+", "t", 100, 0, 5, 5, 5, 60, "5M", "AS:i:-12")


def test_pretty_print_known_answer():
    """paf_pretty_print (impl/paf.c:262-316), columns derived by hand: 4= 1X 2I 2= 1D 1= of q[0:10) on t[1:10)."""
    line = b"q\t10\t0\t10\t+\tt\t12\t1\t10\t8\t10\t60\tAS:i:7\tcg:Z:4=1X2I2=1D1="
    rc, out = O.pretty_print(line, b"ACGTTGGACA", b"NACGTAACTAGG")
    assert rc == 0
    assert out == (b"Query:q\tQ-start:0\tQ-length:10\tTarget:t\tT-start:1\tT-length:9\tSame-strand:1\tScore:7\tIdentity:0.875000"
                   b"\tIdentity-with-gaps0.636364\tAligned-bases:8\tQuery-inserts:1\tQuery-deletes:1\n"
                   b"ACGTA--ACTA\nACGTTGGAC-A\n****   ** *\n")
    # - strand: column i pairs t[i] with the complement of q[qe - 1 - i]; the case of the bases is kept, the stars ignore it
    rc, out = O.pretty_print(b"q\t6\t0\t6\t-\tt\t6\t0\t6\t6\t6\t60\tcg:Z:6M", b"GTaCGn", b"NCGTAC")
    assert rc == 0 and out.split(b"\n")[1:4] == [b"NCGTAC", b"nCGtAC", b"******"]
    # windows of 150 columns: 310 columns -> 150, 150, 10
    rc, out = O.pretty_print(b"q\t310\t0\t310\t+\tt\t310\t0\t310\t310\t310\t60\tcg:Z:310M", b"A" * 310, b"A" * 310)
    rows = out.split(b"\n")[1:-1]
    assert [len(x) for x in rows] == [150] * 6 + [10] * 3
    # without the alignment: the stats line alone
    rc, out2 = O.pretty_print(b"q\t310\t0\t310\t+\tt\t310\t0\t310\t310\t310\t60\tcg:Z:310M", b"A" * 310, b"A" * 310, include_alignment=False)
    assert out2 == out.split(b"\n")[0] + b"\n"


def _chain_line(q, qs, qe, t, ts, te, score, strand=b"+"):
    return b"\t".join([q, b"1000", b"%d" % qs, b"%d" % qe, strand, t, b"1000", b"%d" % ts, b"%d" % te, b"10", b"20", b"60", b"AS:i:%d" % score, b"cg:Z:5M"]) + b"\n"


def test_chain_known_answer():
    """paf_chain (impl/chaining.c:136-343) worked by hand, gap cost 10 + 1 per base, max gap 1000, no trim:
    A q[0,100) t[0,100) 100; B q[110,200) t[120,200) 80: gap (10, 20) costs 40 < 80, chain score 80 + 100 - 40 = 140;
    C q[105,150) t[300,350) 50: gap (5, 200) from A costs 215 >= 50, stays alone; D q[210,300) t[210,300) 90: from B the gap
    (10, 10) costs 30, 90 + 140 - 30 = 200 (from A 230 >= 90). Chains: D-B-A = 90 + (80 - 30) + (100 - 40) = 200 (id 0), C = 50 (id 1);
    printed by descending alignment score."""
    data = (_chain_line(b"q", 0, 100, b"t", 0, 100, 100) + _chain_line(b"q", 110, 200, b"t", 120, 200, 80) + _chain_line(b"q", 105, 150, b"t", 300, 350, 50) +
            _chain_line(b"q", 210, 300, b"t", 210, 300, 90))
    out, err, fresh = O.chain(data, 10, 1, 1000, 0.0)
    assert err.code == 0 and fresh == 0
    tags = [(ln.split(b"\t")[12], ln.split(b"\t")[13], ln.split(b"\t")[14]) for ln in out.splitlines()]
    assert tags == [(b"AS:i:100", b"cn:i:0", b"s1:i:200"), (b"AS:i:90", b"cn:i:0", b"s1:i:200"), (b"AS:i:80", b"cn:i:0", b"s1:i:200"), (b"AS:i:50", b"cn:i:1", b"s1:i:50")]
    # the same four on the - strand, query mirrored (q' = 1000 - q): the + chains are numbered first
    neg = (_chain_line(b"q", 900, 1000, b"t", 0, 100, 100, b"-") + _chain_line(b"q", 800, 890, b"t", 120, 200, 80, b"-") +
           _chain_line(b"q", 700, 790, b"t", 210, 300, 90, b"-"))
    out2, err, fresh = O.chain(neg + data, 10, 1, 1000, 0.0)
    by_strand = {(ln.split(b"\t")[4], ln.split(b"\t")[12]): ln.split(b"\t")[13:15] for ln in out2.splitlines()}
    assert by_strand[(b"-", b"AS:i:90")] == [b"cn:i:2", b"s1:i:200"] and by_strand[(b"+", b"AS:i:90")] == [b"cn:i:0", b"s1:i:200"]
    # the default trim of 1.0 shrinks every alignment to its centre: B overlaps A by 10 on the query and still chains
    over = _chain_line(b"q", 0, 100, b"t", 0, 100, 9000) + _chain_line(b"q", 90, 200, b"t", 95, 200, 8000)
    out3, err, fresh = O.chain(over)
    assert [ln.split(b"\t")[13] for ln in out3.splitlines()] == [b"cn:i:0", b"cn:i:0"]
    assert O.chain(over, trim=0.0)[0].count(b"cn:i:1") == 1  # untrimmed they overlap: two chains
    # coordinates come back untrimmed
    assert out3.splitlines()[0].split(b"\t")[2:4] == [b"0", b"100"]


def _cov_line(q, ql, qs, qe, score, cg, extra=b""):
    return b"\t".join([q, b"%d" % ql, b"%d" % qs, b"%d" % qe, b"+", b"t", b"100", b"0", b"%d" % (qe - qs), b"5", b"5", b"60", b"AS:i:%d" % score]) + extra + b"\tcg:Z:" + cg + b"\n"


def test_tile_and_to_bed_known_answer():
    """paffy tile (impl/paf_tile.c:28-93,156-178) and to_bed (impl/paf_to_bed.c:33-55) worked by hand on three records of one query of
    20 bases: by descending score R1 q[0,10) 100, R2 q[5,15) 90, R3 q[8,12) 80. Counters after R1: 1 on [0,10) -> its ten bases all at
    level 1; after R2: [5,10) = 2, [10,15) = 1 -> five bases at 1, five at 2, half of ten is reached at level 1; after R3: [8,10) = 3,
    [10,12) = 2 -> two at 2, two at 3, half of four is reached at level 2. tp is P for level 1, S above (impl/paf.c:343-348)."""
    data = _cov_line(b"q", 20, 8, 12, 80, b"4M") + _cov_line(b"q", 20, 0, 10, 100, b"10M") + _cov_line(b"q", 20, 5, 15, 90, b"10M")
    out, err = O.tile(data)
    assert err.code == 0
    assert out == (b"q\t20\t0\t10\t+\tt\t100\t0\t10\t5\t5\t60\ttp:A:P\tAS:i:100\ttl:i:1\tcg:Z:10M\n"
                   b"q\t20\t5\t15\t+\tt\t100\t0\t10\t5\t5\t60\ttp:A:P\tAS:i:90\ttl:i:1\tcg:Z:10M\n"
                   b"q\t20\t8\t12\t+\tt\t100\t0\t4\t5\t5\t60\ttp:A:S\tAS:i:80\ttl:i:2\tcg:Z:4M\n")
    # a chain score outranks the alignment score (paf_cmp_by_descending_score): R3 first -> it sees an empty query: level 1
    data2 = _cov_line(b"q", 20, 8, 12, 80, b"4M", b"\ts1:i:7") + _cov_line(b"q", 20, 0, 10, 100, b"10M")
    out2, _ = O.tile(data2)
    assert [ln.split(b"\t")[14] for ln in out2.splitlines()] == [b"tl:i:1", b"tl:i:1"] and out2.startswith(b"q\t20\t8\t12")
    # insertions count, deletions do not (impl/paf.c:691-712): 2M1D2M1I1M covers q[0,6) at 0,1,2,3 and 5
    bed, err = O.to_bed(_cov_line(b"q", 8, 0, 6, 1, b"2M1D2M1I1M"))
    assert err.code == 0 and bed == b"q 0 4 1\nq 4 5 0\nq 5 6 1\nq 6 8 0\n"
    bed, _ = O.to_bed(data)
    assert bed == b"q 0 5 1\nq 5 8 2\nq 8 10 3\nq 10 12 2\nq 12 15 1\nq 15 20 0\n"
    assert O.to_bed(data, binary=True)[0] == b"q 0 15 1\nq 15 20 0\n"
    assert O.to_bed(data, exclude_unaligned=True, min_size=3)[0] == b"q 0 5 1\nq 5 8 2\nq 12 15 1\n"


def test_dedupe_known_answer():
    """paffy dedupe (impl/paf_dedupe.c:27-46,117-143): a record is dropped when an earlier written one has the same names, strand and
    four coordinates; with -a also when an earlier one equals it with query and target swapped."""
    a = b"q\t100\t0\t10\t+\tt\t200\t5\t15\t10\t10\t60\tcg:Z:10M\n"
    a_again = b"q\t100\t0\t10\t+\tt\t200\t5\t15\t9\t10\t7\tAS:i:5\tcg:Z:4M2I4M2D\n"  # same key, everything else differs
    a_minus = a.replace(b"\t+\t", b"\t-\t")
    a_inv = b"t\t200\t5\t15\t+\tq\t100\t0\t10\t10\t10\t60\tcg:Z:10M\n"
    data = a + a_again + a_minus + a_inv

    def written(x):  # the writer prints AS for every record, 0 when the line had none (score is calloc'ed, impl/paf.c:349)
        return x.rstrip(b"\n").replace(b"\tcg:Z:", b"\tAS:i:0\tcg:Z:")

    out, err = O.dedupe(data)
    assert err.code == 0 and out.splitlines() == [written(x) for x in (a, a_minus, a_inv)]
    out, err = O.dedupe(data, check_inverse=True)
    assert err.code == 0 and out.splitlines() == [written(x) for x in (a, a_minus)]


def test_chain_fresh_iterator_walk_is_the_only_switchable_part():
    """impl/chaining.c:74-76: when no active chain sorts <= the key the reference walks the set from a fresh iterator (libavl: from its
    last element). The oracle restates that walk and counts its candidates; with the walk switched off (what the GPU implements,
    DESIGN 5) the same input gives two chains of one record instead of one chain of two. Inputs without such candidates do not depend
    on the switch."""
    def ln(qs, qe, ts, te):
        return f"q\t1000\t{qs}\t{qe}\t+\tt\t1000\t{ts}\t{te}\t{qe - qs}\t{qe - qs}\t60\tAS:i:100\tcg:Z:{qe - qs}M\n".encode()
    a, b = ln(0, 100, 0, 100), ln(100, 200, 100, 200)
    out, err, fresh = O.chain(b + a, trim=0.0)
    assert err.code == 0 and fresh == 1 and out.count(b"cn:i:0\ts1:i:200") == 2
    out2, err2, fresh2 = O.chain(b + a, trim=0.0, fresh_walk=False)
    assert err2.code == 0 and fresh2 == 0 and out2.count(b"s1:i:100") == 2 and b"cn:i:1" in out2
    for fw in (True, False):  # A first: the ordinary search finds it
        out3, _, fresh3 = O.chain(a + b, trim=0.0, fresh_walk=fw)
        assert fresh3 == 0 and out3.count(b"s1:i:200") == 2
