"""The N-GPU path behind the CLI on real hardware (one GPU: PAFFY_ONE_DEVICE=1 puts every worker on device 0): `PAFFY_GPUS=2 bin/paffy
<cmd>` must write what one GPU writes. The launcher's logic itself is covered without a GPU in tests/test_launcher.py."""
import os
import subprocess

import pytest

import oracle_lib as O
import synth_lib
from test_launcher import tile_records

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAFFY = os.path.join(ROOT, "bin", "paffy")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])


def paffy(args, gpus, data=None):
    env = {k: v for k, v in os.environ.items() if k not in ("PAFFY_WORKER", "PAFFY_GPUS")}
    if gpus > 1:
        env.update(PAFFY_GPUS=str(gpus), PAFFY_ONE_DEVICE="1")
    return subprocess.run([PAFFY] + args, input=data, env=env, capture_output=True, timeout=600)


def test_tile_two_workers_equals_one(tmp_path, human_chimp):
    data = human_chimp + tile_records(1500) + synth_lib.generate(0x5EED0005, 300, 0, 400)
    src = tmp_path / "in.paf"
    src.write_bytes(data)
    want, err = O.tile(data)
    assert err.code == 0
    one = paffy(["tile", "-i", str(src)], 1)
    assert one.returncode == 0 and one.stdout == want
    for n in (2, 3):
        p = paffy(["tile", "-i", str(src), "-o", str(tmp_path / f"out{n}.paf")], n)
        assert p.returncode == 0, p.stderr[-2000:]
        assert (tmp_path / f"out{n}.paf").read_bytes() == want


def test_stream_pipe_two_workers_equals_one(tmp_path):
    data = synth_lib.generate(0x5EED0003, 300, 0, 2000)
    want, err = O.run([O.stage(O.INVERT), O.stage(O.TRIM_IDENTITY), O.stage(O.SHATTER)], data)
    assert err.code == 0
    a = paffy(["invert"], 2, data=data)
    b = paffy(["trim"], 2, data=a.stdout)
    c = paffy(["shatter"], 3, data=b.stdout)
    assert a.returncode == 0 and b.returncode == 0 and c.returncode == 0, (a.stderr, b.stderr, c.stderr)
    assert c.stdout == want


def test_failing_record_with_two_workers(tmp_path):
    good = synth_lib.generate(0x5EED0003, 60, 0, 40).splitlines(keepends=True)
    data = b"".join(good[:25]) + b"q\t10\t0\t5\t*\tt\t10\t0\t5\t5\t5\t60\n" + b"".join(good[25:])
    want, err = O.run([O.stage(O.INVERT)], data)
    p = paffy(["invert"], 2, data=data)
    assert err.code == O.ERR_STRAND and p.returncode == 1 and p.stdout == want


def test_tile_with_one_query_name_and_with_no_input(tmp_path):
    """fewer query names than workers (a per-contig split has one) and an empty input: the workers without lines are not started; the
    output is one GPU's, an empty input gives an empty output and status 0 (round-3 advice: `worker %d left no output`)"""
    data = tile_records(400, contigs=1)
    src = tmp_path / "one.paf"
    src.write_bytes(data)
    want, err = O.tile(data)
    assert err.code == 0
    p = paffy(["tile", "-i", str(src)], 3)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout == want
    empty = tmp_path / "empty.paf"
    empty.write_bytes(b"")
    for n in (1, 3):
        p = paffy(["tile", "-i", str(empty)], n)
        assert p.returncode == 0 and p.stdout == b"", (n, p.stderr[-2000:])
    p = paffy(["trim", "-fi", str(src), "-t", "0.2"], 2)  # clustered flag in front of -i
    q = paffy(["trim", "-f", "-t", "0.2", "-i", str(src)], 1)
    assert p.returncode == 0 and q.returncode == 0 and p.stdout == q.stdout and len(p.stdout) > 0
