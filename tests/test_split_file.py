"""paffy split_file (impl/paf_split_file.c:131-173): oracle known answers on CPU, the CLI (GPU
normalisation + host routing) against the oracle's files on the GPU."""
import os
import subprocess

import pytest

import oracle_lib as O
from test_cli import PAFFY

R = [b"qa\t900\t0\t5\t+\tchr/1\t5000\t0\t5\t5\t5\t60\tNM:i:0\tcg:Z:5M\n",
     b"qb\t100\t0\t5\t-\ttiny1\t40\t0\t5\t5\t5\t60\tcg:Z:5M\n",
     b"qa\t900\t9\t14\t+\ttiny2\t70\t1\t6\t5\t5\t60\tcg:Z:5X\n",
     b"qc\t100\t0\t5\t+\ttiny3\t50\t0\t5\t5\t5\t60\n",
     b"qb\t100\t0\t5\t+\ttiny1\t40\t2\t7\t5\t5\t60\tcg:Z:2M1D3M\n",
     b"qd\t100\t0\t5\t+\tchr/1\t5000\t7\t12\t5\t5\t60\tcg:Z:5M\n"]


def read_dir(d):
    return {n: open(os.path.join(d, n), "rb").read() for n in sorted(os.listdir(d))}


def norm(line):
    return O.dedupe(line)[0]  # paf_read(.., 0) -> paf_write of one record


def test_oracle_known_answers(tmp_path):
    data = b"".join(R)
    d = tmp_path / "a"
    d.mkdir()
    assert O.split_file(data, str(d) + "/split_").code == 0
    got = read_dir(d)  # by target; '/' in a name becomes '_'
    assert set(got) == {"split_chr_1.paf", "split_tiny1.paf", "split_tiny2.paf", "split_tiny3.paf"}
    assert got["split_chr_1.paf"] == norm(R[0]) + norm(R[5]) and got["split_tiny1.paf"] == norm(R[1]) + norm(R[4])
    d = tmp_path / "b"
    d.mkdir()
    assert O.split_file(data, str(d) + "/p.", min_length=100).code == 0
    got = read_dir(d)  # only the current small file is ever tried: tiny1 (40) | tiny2 (70, 110 > 100) | tiny3 (50, 120 > 100)
    assert set(got) == {"p.chr_1.paf", "p.small_0.paf", "p.small_1.paf", "p.small_2.paf"}
    assert got["p.small_0.paf"] == norm(R[1]) + norm(R[4]) and got["p.small_1.paf"] == norm(R[2]) and got["p.small_2.paf"] == norm(R[3])
    d = tmp_path / "b2"
    d.mkdir()
    assert O.split_file(data, str(d) + "/p.", min_length=120).code == 0
    got = read_dir(d)  # tiny1 + tiny2 = 110 fit one file, tiny3 opens the next
    assert got["p.small_0.paf"] == norm(R[1]) + norm(R[2]) + norm(R[4]) and got["p.small_1.paf"] == norm(R[3])
    d = tmp_path / "c"
    d.mkdir()
    assert O.split_file(data, str(d) + "/", by_query=True).code == 0
    assert set(read_dir(d)) == {"qa.paf", "qb.paf", "qc.paf", "qd.paf"}


@pytest.mark.gpu
def test_cli_split_file(human_chimp, tmp_path):
    cases = [([], dict()), (["-q"], dict(by_query=True)), (["-m", "100000000"], dict(min_length=100000000)),
             (["--query", "--minLength", "60000000"], dict(by_query=True, min_length=60000000))]
    for i, (args, kw) in enumerate(cases):
        for data in (b"".join(R), human_chimp):
            want_dir, got_dir = tmp_path / f"want{i}_{len(data)}", tmp_path / f"got{i}_{len(data)}"
            want_dir.mkdir()
            got_dir.mkdir()
            assert O.split_file(data, str(want_dir) + "/s_", **kw).code == 0
            p = subprocess.run([PAFFY, "split_file", "-p", str(got_dir) + "/s_"] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env={"PAFFY_CHUNK_MB": "1", "PATH": "/usr/bin:/bin"})
            assert p.returncode == 0 and p.stdout == b"", p.stderr
            assert read_dir(got_dir) == read_dir(want_dir), args
    assert subprocess.run([PAFFY, "split_file", "-h"], stderr=subprocess.PIPE).returncode == 0
