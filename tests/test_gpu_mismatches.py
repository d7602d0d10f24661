"""add_mismatches (M -> =/X against FASTA) and add_mismatches -a (=/X -> M) on the GPU vs the oracle."""
import random

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
S = O.stage


@pytest.fixture(scope="module")
def eng():
    import paffy_amd

    e = paffy_amd.Engine()
    yield e
    e.close()


def gs(stages):
    import paffy_amd

    return [paffy_amd.Stage(s.kind, s.p0, s.p1) for s in stages]


def make_genomes(rng, n_contigs=3, length=6000):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    seqs = {}
    for c in range(n_contigs):
        t = [rng.choice("ACGT") for _ in range(length)]
        for _ in range(length // 200):
            t[rng.randrange(length)] = "N"
        q = list(t)
        for i in range(length):
            r = rng.random()
            if r < 0.03:
                q[i] = rng.choice("ACGT")
            elif r < 0.06:
                q[i] = q[i].lower()
        seqs[f"pt.chr{c + 1}"] = "".join(x.lower() if rng.random() < 0.02 else x for x in t)
        seqs[f"hs.chr{c + 1}"] = "".join(q)
        # a reverse-complement relative of the target, for '-' records on the diagonal
        seqs[f"rc.chr{c + 1}"] = "".join(comp.get(x.upper(), x) for x in reversed(q))
    return seqs


def make_records(rng, seqs, n, length=6000):
    out = []
    for r in range(n):
        c = rng.randrange(1, 4)
        ops, qspan, tspan = [], 0, 0
        for k in range(rng.choice([1, 2, 5, 20, 60])):
            L = rng.choice([1, 2, 7, 40, 150])
            ops.append(f"{L}M"); qspan += L; tspan += L
            if rng.random() < 0.7:
                g = rng.choice([1, 2, 5])
                if rng.random() < 0.5:
                    ops.append(f"{g}I"); qspan += g
                else:
                    ops.append(f"{g}D"); tspan += g
        ops.append("3M"); qspan += 3; tspan += 3
        mode = rng.randrange(3)
        ts = rng.randrange(0, length - tspan)
        if mode == 0:      # '+' on the diagonal: long '=' runs
            q, strand, qs = f"hs.chr{c}", "+", min(ts, length - qspan)
        elif mode == 1:    # '-' against the reverse-complement relative
            q, strand = f"rc.chr{c}", "-"
            qs = max(0, min(length - qspan, length - ts - qspan))
        else:              # anywhere: short runs
            q, strand, qs = f"hs.chr{rng.randrange(1, 4)}", rng.choice("+-"), rng.randrange(0, length - qspan)
        out.append(f"{q}\t{length}\t{qs}\t{qs + qspan}\t{strand}\tpt.chr{c}\t{length}\t{ts}\t{ts + tspan}\t{tspan}\t{tspan}\t60\ttp:A:P\tcg:Z:{''.join(ops)}\n")
    return "".join(out).encode()


def test_add_and_remove_mismatches(eng):
    rng = random.Random(7)
    seqs = make_genomes(rng)
    data = make_records(rng, seqs, 400)
    want, werr = O.run([S(O.ADD_MISMATCHES)], data, seqs)
    assert werr.code == 0 and b"=" in want and b"X" in want
    eng.set_sequences(seqs)
    got, info = eng.run(gs([S(O.ADD_MISMATCHES)]), data)
    assert got == want
    # =/X -> M again: both the LDS class and the original text
    back, _ = eng.run(gs([S(O.REMOVE_MISMATCHES)]), got)
    assert back == O.run([S(O.REMOVE_MISMATCHES)], want)[0]
    assert back == O.run([S(O.REMOVE_MISMATCHES)], data)[0]  # adjacent M ops of the input merge too
    # fused with other stages
    for stages in ([S(O.ADD_MISMATCHES), S(O.TRIM_IDENTITY)], [S(O.INVERT), S(O.ADD_MISMATCHES)],
                   [S(O.ADD_MISMATCHES), S(O.REMOVE_MISMATCHES), S(O.SHATTER)], [S(O.REMOVE_MISMATCHES), S(O.INVERT)]):
        src = got if stages[0].kind == O.REMOVE_MISMATCHES else data
        w, e = O.run(stages, src, seqs)
        g, i = eng.run(gs(stages), src, raise_on_error=False)
        assert i.error.code == e.code and g == w, [s.kind for s in stages]


def test_remove_mismatches_fixture(eng, human_chimp):
    assert eng.run(gs([S(O.REMOVE_MISMATCHES)]), human_chimp)[0] == O.run([S(O.REMOVE_MISMATCHES)], human_chimp)[0]
    kat = b"q\t100\t0\t6\t+\tt\t100\t0\t5\t5\t6\t60\tcg:Z:3=2X1I\n"
    assert eng.run(gs([S(O.REMOVE_MISMATCHES)]), kat)[0] == b"q\t100\t0\t6\t+\tt\t100\t0\t5\t5\t6\t60\tAS:i:0\tcg:Z:5M1I\n"


def test_known_answers_and_errors(eng):
    """paf_unit_test.c:525-559 and the missing-sequence exits of impl/paf_add_mismatches.c:117-127."""
    def enc(q, t, L):
        eng.set_sequences({"q": q, "t": t})
        rec = f"q\t{L}\t0\t{L}\t+\tt\t{L}\t0\t{L}\t{L}\t{L}\t60\tcg:Z:{L}M\n".encode()
        return eng.run(gs([S(O.ADD_MISMATCHES)]), rec)[0].split(b"cg:Z:")[1].strip()
    assert enc("AAAAA", "AAAAA", 5) == b"5=" and enc("AAAAA", "CCCCC", 5) == b"5X" and enc("AATT", "AACC", 4) == b"2=2X"
    ok = b"q\t5\t0\t5\t+\tt\t5\t0\t5\t5\t5\t60\tcg:Z:5M\n"
    eng.set_sequences({"q": "AAAAA", "t": "AAAAA"})
    for bad, code in ((ok.replace(b"q\t5", b"zz\t5"), 17), (ok.replace(b"\tt\t", b"\tzz\t"), 18)):
        got, info = eng.run(gs([S(O.ADD_MISMATCHES)]), ok + bad + ok, raise_on_error=False)
        w, e = O.run([S(O.ADD_MISMATCHES)], ok + bad + ok, {"q": "AAAAA", "t": "AAAAA"})
        assert (info.error.code, info.error.record) == (code, 1) == (e.code, e.record) and got == w


def test_cfg4_workload_device_equals_host_and_oracle(eng):
    """cfg4: the device generator writes the same records and genomes as the host build, the records sit on
    homologous bases (98 % identity on both strands), and add_mismatches over them equals the oracle."""
    import re

    import synth_lib

    args = dict(n_contigs=8, tlen_min=60_000, tlen_span=90_000)
    host = synth_lib.Synth4(0x5EED0004, 512, **args)
    want_recs = host.records(5, 600)
    seqs = host.genomes()
    eng.synth4_setup(0x5EED0004, 512, **args)
    buf, nbytes = eng.synth4(5, 600)
    assert bytes(buf[:nbytes].cpu().numpy().tobytes()) == want_recs
    # genomes as the device wrote them: add_mismatches against them must equal the oracle against the host genomes
    want, werr = O.run([S(O.ADD_MISMATCHES)], want_recs, seqs)
    assert werr.code == 0
    got, info = eng.run(gs([S(O.ADD_MISMATCHES)]), want_recs)
    assert got == want
    eq = sum(int(x) for x in re.findall(rb"(\d+)=", want))
    xx = sum(int(x) for x in re.findall(rb"(\d+)X", want))
    assert 0.975 < eq / (eq + xx) < 0.985
    assert any(b"\t-\t" in line for line in want_recs.split(b"\n"))
    # and through the host-loaded store too (same bytes either way)
    eng.set_sequences(seqs)
    assert eng.run(gs([S(O.ADD_MISMATCHES)]), want_recs)[0] == want
    for stages in ([S(O.ADD_MISMATCHES), S(O.SHATTER)], [S(O.INVERT), S(O.ADD_MISMATCHES), S(O.TRIM_IDENTITY)]):
        w, e = O.run(stages, want_recs, seqs)
        g, i = eng.run(gs(stages), want_recs, raise_on_error=False)
        assert i.error.code == e.code and g == w


def test_alternating_columns_grow_the_arena(eng):
    """Every column differs from the one before: sixteen ops per 16-column chunk, far more than the arena was sized for --
    the sizing pass must be repeated with a bigger arena, and records whose new ops outgrow every LDS store (a stage follows)
    take the arena class. Equal to the oracle either way."""
    n = 200_000
    seqs = {"t": "AC" * (n // 2), "q": "A" * n}
    recs = []
    for k in range(40):
        span = [50, 3000, 20_000, 70_000][k % 4]
        qs = 17 * k
        recs.append(f"q\t{n}\t{qs}\t{qs + span}\t+\tt\t{n}\t{qs + 1}\t{qs + 1 + span}\t{span}\t{span}\t60\tcg:Z:{span}M\n")
    data = "".join(recs).encode()
    e2 = type(eng)()  # a fresh context: its arena starts small
    e2.set_sequences(seqs)
    for stages in ([S(O.ADD_MISMATCHES)], [S(O.ADD_MISMATCHES), S(O.INVERT)]):
        want, werr = O.run(stages, data, seqs)
        got, info = e2.run(gs(stages), data, raise_on_error=False)
        assert werr.code == 0 and info.error.code == 0 and got == want
    assert want.count(b"1X") > 100_000
    e2.close()


def test_long_lines_and_a_scratch_demand_that_outgrows_the_first_guess(eng):
    """`add_mismatches` alone goes through the flat pass (paffy_amd/csrc/flat_add_kernel.h): pieces of cigar text encoded one wave each.
    Records of tens of thousands of ops become lines of more than 32 768 ops -- written as segments by the one-wave line writer -- on
    both strands; records whose M ops meet alternating columns become fifty times their ops -- the scratch and the new ops outgrow what
    the first try allowed and the batch is encoded again; a missing sequence and a record that leaves its sequence are the record
    kernels' to report, between records the flat pass keeps."""
    import random

    from test_gpu_flat import random_ops, record

    rng = random.Random(77)
    n = 3_000_000
    t = "".join(rng.choice("ACGT") for _ in range(n))
    q = list(t)
    for i in range(0, n, 37):  # a substitution every 37 bases, some lower case
        q[i] = "ACGT"["ACGT".index(q[i]) ^ 1]
    for i in range(0, n, 101):
        q[i] = q[i].lower()
    q = "".join(q)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a"}
    qrc = "".join(comp[c] for c in reversed(q))
    seqs = {"tt": t, "qf": q, "qr": qrc, "alt": "AC" * 200_000, "aaa": "A" * 400_000}
    eng.set_sequences(seqs)
    recs = []
    for k, n_ops in enumerate([300, 20_000, 45_000, 70_000, 1200]):
        ops = random_ops(rng, n_ops, lens=(1, 5, 30, 60, 110), indel=(1, 2, 3))
        span_q = sum(L for L, c in ops if c in "MI")
        span_t = sum(L for L, c in ops if c in "MD")
        s0 = rng.randrange(0, 1000)
        recs.append(record(ops, "+", qname="qf", tname="tt", qlen=n, tlen=n, qs=s0, ts=s0 + 3))
        recs.append(record(ops, "-", qname="qr", tname="tt", qlen=n, tlen=n, qs=n - s0 - span_q, ts=s0 + 3))
        assert s0 + 3 + span_t < n
    for k in range(3):  # fifty ops per M op: 2 500 M ops of 50 alternating columns each
        ops = [(50, "M") if i % 2 == 0 else (1, "I") for i in range(4999)]
        recs.append(record(ops, "+", qname="aaa", tname="alt", qlen=400_000, tlen=400_000, qs=10 + k, ts=20 + k))
    good = recs[0]
    missing = good.replace("qf\t", "nope\t", 1)
    outside = record([(40, "M")], "+", qname="qf", tname="tt", qlen=n + 100, tlen=n, qs=n + 10, ts=5)
    stages = [S(O.ADD_MISMATCHES)]
    data = "".join(recs).encode()
    e2 = type(eng)()  # a fresh context: its buffers start at the first guess
    e2.set_sequences(seqs)
    want, werr = O.run(stages, data, seqs)
    got, info = e2.run(gs(stages), data, raise_on_error=False)
    assert werr.code == 0 and info.error.code == 0
    assert got == want
    left, _ = e2.flat_stats()
    assert left == 0, left  # the flat pass took every record: long lines and the second try included
    assert max(line.count(b"=") + line.count(b"X") for line in want.split(b"\n")) > 100_000
    for bad in (missing, outside):
        d2 = (good + bad + good).encode()
        w, e = O.run(stages, d2, seqs)
        g, i = e2.run(gs(stages), d2, raise_on_error=False)
        assert (i.error.code, i.error.record) == (e.code, e.record) and e.code != 0 and g == w
    e2.close()
