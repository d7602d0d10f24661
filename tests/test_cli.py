"""`bin/paffy` CLI: dispatcher contract on CPU, byte parity with the oracle on the GPU."""
import os
import subprocess

import pytest

import oracle_lib as O
from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAFFY = os.path.join(ROOT, "bin", "paffy")
S = O.stage


@pytest.fixture(scope="module", autouse=True)
def built():
    import paffy_amd

    paffy_amd.build_library()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])


def run(args, data=b"", env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([PAFFY] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    return p.returncode, p.stdout, p.stderr


def test_dispatcher_contract():
    """paffy_main.c:46-84: no args -> usage rc 0; unknown command -> rc 1; -h -> rc 0; bad flag -> rc 1."""
    rc, out, err = run([])
    assert rc == 0 and out == b"" and b"usage: paffy <command>" in err
    for name in (b"add_mismatches", b"invert", b"shatter", b"tile", b"trim", b"chain", b"view", b"split_file", b"dedupe", b"filter"):
        assert name in err
    rc, _, err = run(["frobnicate"])
    assert rc == 1 and b"frobnicate is not a valid paffy command" in err
    for cmd in ("shatter", "invert", "trim", "tile", "add_mismatches", "filter", "dedupe"):
        rc, out, err = run([cmd, "-h"])
        assert rc == 0 and out == b"" and b"--inputFile" in err and b"--logLevel" in err
        assert run([cmd, "--help"])[0] == 0
        assert run([cmd, "-Z"])[0] == 1
    assert b"--trimIdentity" in run(["trim", "-h"])[2] and b"--fixedTrim" in run(["trim", "-h"])[2]
    assert b"--minIdentityWithGaps" in run(["filter", "-h"])[2] and b"--maxTileLevel" in run(["filter", "-h"])[2]
    assert run(["chain"])[0] == 1  # outside the hot path


@pytest.mark.gpu
def test_cli_matches_oracle(human_chimp, tmp_path):
    cases = [(["shatter"], [S(O.SHATTER)]), (["invert"], [S(O.INVERT)]), (["trim"], [S(O.TRIM_IDENTITY)]),
             (["trim", "-r", "0.2", "-t", "0.5"], [S(O.TRIM_IDENTITY, 0.2, 0.5)]),
             (["trim", "--fixedTrim", "--trimFraction", "0.1"], [S(O.TRIM_FIXED, 0.05, 0.1)])]
    for args, stages in cases:
        rc, out, err = run(args, human_chimp)
        assert rc == 0, err
        assert out == O.run(stages, human_chimp)[0], args
    # -i / -o files and small chunks (lines straddling chunk boundaries)
    src, dst = tmp_path / "in.paf", tmp_path / "out.paf"
    src.write_bytes(human_chimp)
    rc, out, err = run(["invert", "-i", str(src), "-o", str(dst), "-l", "INFO"], env={"PAFFY_CHUNK_MB": "1"})
    assert rc == 0 and out == b"" and b"Input file string" in err
    assert dst.read_bytes() == O.run([S(O.INVERT)], human_chimp)[0]


@pytest.mark.gpu
def test_cli_shell_pipe(human_chimp):
    """cfg 1 of BASELINE.json and the cfg-3 shape as real shell pipes of three processes."""
    p = subprocess.run(f"{PAFFY} shatter | {PAFFY} invert", shell=True, input=human_chimp, stdout=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == O.run([S(O.SHATTER), S(O.INVERT)], human_chimp)[0]
    p = subprocess.run(f"{PAFFY} invert | {PAFFY} trim | {PAFFY} shatter", shell=True, input=human_chimp, stdout=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == O.run([S(O.INVERT), S(O.TRIM_IDENTITY), S(O.SHATTER)], human_chimp)[0]


@pytest.mark.gpu
def test_cli_error_status():
    ok = b"q\t100\t0\t5\t+\tt\t100\t0\t5\t5\t5\t60\tcg:Z:5M\n"
    bad_strand = b"q\t100\t0\t5\t*\tt\t100\t0\t5\t5\t5\t60\tcg:Z:5M\n"
    rc, out, err = run(["invert"], ok + bad_strand + ok)
    assert rc == 1 and out == O.run([S(O.INVERT)], ok)[0] and b"unexpected strand character (*)" in err
    rc, out, err = run(["shatter"], ok + ok.replace(b"5M", b"5=") + ok)
    assert rc == -6 and out == O.run([S(O.SHATTER)], ok)[0]  # assert -> SIGABRT, as the reference


@pytest.mark.gpu
def test_cli_tile_and_mismatches(human_chimp, tmp_path):
    rc, out, err = run(["tile"], human_chimp)
    assert rc == 0 and out == O.tile(human_chimp)[0]
    rc, out, err = run(["add_mismatches", "-a"], human_chimp)
    assert rc == 0 and out == O.run([S(O.REMOVE_MISMATCHES)], human_chimp)[0]
    # FASTA files on the command line, as `paffy add_mismatches q.fa t.fa`
    seqs = {"q": "ACGTACGTAC" * 3, "t": "ACGTACGAAC" * 3}
    (tmp_path / "q.fa").write_text(">q\n" + seqs["q"][:17] + "\n" + seqs["q"][17:] + "\n")
    (tmp_path / "t.fa").write_text(">t\n" + seqs["t"] + "\n")
    rec = b"q\t30\t2\t27\t+\tt\t30\t2\t27\t25\t25\t60\tcg:Z:25M\n"
    rc, out, err = run(["add_mismatches", str(tmp_path / "q.fa"), str(tmp_path / "t.fa")], rec)
    assert rc == 0 and out == O.run([S(O.ADD_MISMATCHES)], rec, seqs)[0] and b"X" in out
    rc, out, err = run(["add_mismatches", str(tmp_path / "q.fa")], rec)  # target sequence missing: exit(1)
    assert rc == 1 and out == b"" and b"No target sequence" in err


@pytest.mark.gpu
def test_cli_filter(human_chimp):
    """paffy filter flags (impl/paf_filter.c:45-100) and the tile | filter -w 1 step of the reference pipeline
    (tests/paf_pipeline_test.sh:79)."""
    for args, kw in ((["-t", "5000"], dict(min_alignment_score=5000)), (["-t", "5000", "-x"], dict(min_alignment_score=5000, invert=True)),
                     (["--minIdentity", "0.9"], dict(min_identity=0.9)), (["-v", "0.85", "-s", "-1"], dict(min_identity_with_gaps=0.85))):
        rc, out, err = run(["filter"] + args, human_chimp)
        assert rc == 0, err
        assert out == O.filter(human_chimp, **kw)[0], args
    p = subprocess.run(f"{PAFFY} tile | {PAFFY} filter -w 1", shell=True, input=human_chimp, stdout=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == O.filter(O.tile(human_chimp)[0], max_tile_level=1)[0] and 0 < p.stdout.count(b"\n") < 207
    O.set_filter()
