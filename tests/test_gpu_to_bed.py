"""`paffy to_bed` (impl/paf_to_bed.c; tests/paf_tools_test.sh:70-99 runs it with -b, -e, -f, -n): coverage runs on the GPU against
the oracle, byte for byte (both write the sequences in order of first appearance; the reference's hash order is not defined)."""
import os
import random
import subprocess

import pytest

import oracle_lib as O
import synth_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAFFY = os.path.join(ROOT, "bin", "paffy")
OPTS = [dict(), dict(binary=True), dict(exclude_unaligned=True), dict(exclude_aligned=True), dict(include_inverted=True),
        dict(include_inverted=True, binary=True, min_size=50), dict(min_size=1000, exclude_unaligned=True)]


@pytest.fixture(scope="module")
def eng():
    import paffy_amd

    e = paffy_amd.Engine()
    yield e
    e.close()


def check(eng, data, **kw):
    want, werr = O.to_bed(data, **kw)
    got, info = eng.to_bed(data, raise_on_error=False, **kw)
    assert info.error.code == werr.code, (kw, info.error.code, werr.code)
    if werr.code:
        assert got == b"" and info.error.record == werr.record
    else:
        assert got == want, kw
    return want


def test_fixture_and_options(eng, human_chimp):
    for kw in OPTS:
        out = check(eng, human_chimp, **kw)
        # human and chimp chromosomes share their names: with -n the second role of "chr10" has another length -> the length assert
        assert out.count(b"\n") > 100 or kw.get("min_size", 1) > 1 or kw.get("include_inverted")
    assert O.to_bed(human_chimp, include_inverted=True)[1].code == 19


def test_synthetic_piles_both_strands(eng):
    host = synth_lib.Synth4(0x5EED0004, 512, n_contigs=5, tlen_min=1_500_000, tlen_span=1_000_000)  # sequences over several 1 Mi slices
    data = host.records(0, 3000)
    for kw in OPTS:
        check(eng, data, **kw)
    lines = data.splitlines(keepends=True)
    random.Random(3).shuffle(lines)
    check(eng, b"".join(lines[:700]), include_inverted=True)


def test_batches_give_the_same_bytes(eng, human_chimp):
    """to_bed over an input held as several text batches (any size: the counters are per sequence, not per batch)."""
    host = synth_lib.Synth4(0x5EED0004, 512, n_contigs=4, tlen_min=200_000, tlen_span=300_000)
    data = host.records(0, 2500)
    for kw in (dict(), dict(include_inverted=True), dict(binary=True, min_size=20)):
        want, err = O.to_bed(data, **kw)
        assert err.code == 0
        for batch_bytes in (1 << 20, 150_000):
            got, info = eng.to_bed(data, batch_bytes=batch_bytes, **kw)
            assert got == want and info.n_records == data.count(b"\n")
    bad = data + b"q\t30\t2\t13\t+\tt\t40\t5\t15\t10\t10\t60\tcg:Z:10M\n"
    got, info = eng.to_bed(bad, batch_bytes=150_000, raise_on_error=False)
    assert got == b"" and info.error.code == 19 and info.error.record == data.count(b"\n")


def test_edges_and_errors(eng):
    ok = b"q\t30\t2\t12\t+\tt\t40\t5\t15\t10\t10\t60\tcg:Z:10M\n"
    check(eng, b"")
    check(eng, ok)
    check(eng, ok[:-1])                                                      # no newline at the end
    check(eng, ok + b"q\t30\t8\t20\t-\tt\t40\t0\t10\t10\t12\t60\tcg:Z:4M2I6M\n", include_inverted=True)
    check(eng, ok + b"t\t40\t0\t10\t+\tq\t30\t0\t10\t10\t10\t60\tcg:Z:10M\n", include_inverted=True)  # a sequence in both roles
    check(eng, b"q\t30\t4\t4\t+\tt\t40\t5\t5\t0\t0\t60\n", include_inverted=True)                      # no cigar, empty ranges
    check(eng, ok * 40000)                                                    # counters saturate at 32766
    # failures: nothing is written, the first failing record is reported
    check(eng, ok + b"q\t31\t2\t12\t+\tt\t40\t5\t15\t10\t10\t60\tcg:Z:10M\n")                          # sequence length changes
    check(eng, ok + b"q\t30\t2\t13\t+\tt\t40\t5\t15\t10\t10\t60\tcg:Z:10M\n")                          # walk does not end at query_end
    check(eng, ok + b"q\t30\t2\t12\t+\tt\t40\t5\t16\t10\t10\t60\tcg:Z:10M\n" + ok, include_inverted=True)  # only the target side fails
    check(eng, ok + b"q\t30\t2\t12\t+\tt\t40\t5\t15\t10\t10\t60\tcg:Z:10Q\n")                          # bad cigar character
    check(eng, ok + b"q\t30\t2\t12\t*\tt\t40\t5\t15\t10\t10\t60\tcg:Z:10M\n")                          # bad strand
    check(eng, b"q\t30\t2\t12\t+\tt\t40\t5\t15\t10\t10\t60\n")                                        # no cigar but a non-empty range


def test_cli(tmp_path, human_chimp):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    src = tmp_path / "in.paf"
    src.write_bytes(human_chimp)
    for args, kw in ((["-i", str(src)], {}), (["-i", str(src), "-b"], dict(binary=True)), (["-i", str(src), "-e"], dict(exclude_unaligned=True)),
                     (["-i", str(src), "-f", "--minSize", "200"], dict(exclude_aligned=True, min_size=200))):
        r = subprocess.run([PAFFY, "to_bed"] + args, capture_output=True)
        assert r.returncode == 0, r.stderr[-500:]
        assert r.stdout == O.to_bed(human_chimp, **kw)[0], args
    r = subprocess.run([PAFFY, "to_bed", "-o", str(tmp_path / "o.bed")], input=human_chimp, capture_output=True)
    assert r.returncode == 0 and (tmp_path / "o.bed").read_bytes() == O.to_bed(human_chimp)[0]
    # -f -q: sequences of the FASTA that no alignment names are listed as wholly unaligned ("name 0 length<TAB>0", impl/paf_to_bed.c:63-67)
    fa = tmp_path / "q.fa"
    fa.write_bytes(b">chr10\nACGT\n>nowhere\nACGTACGTAC\nACG\n")
    r = subprocess.run([PAFFY, "to_bed", "-i", str(src), "-f", "-q", str(fa)], capture_output=True)
    assert r.returncode == 0 and r.stdout == O.to_bed(human_chimp, exclude_aligned=True)[0] + b"nowhere 0 13\t0\n"
    bad = human_chimp + b"q\t30\t2\t13\t+\tt\t40\t5\t15\t10\t10\t60\tcg:Z:10M\n"
    r = subprocess.run([PAFFY, "to_bed"], input=bad, capture_output=True)
    assert r.returncode == -6 and r.stdout == b""  # assert -> SIGABRT, nothing written
