"""bin/paffy with PAFFY_GPUS=N (host/paffy_launch.c): the N-GPU path behind the kept CLI. These tests run without a GPU: the worker
binary is replaced by tests/standin_worker.py (PAFFY_WORKER), which honours the same contract with the CPU oracle, so the launcher's
own work is what is tested -- byte ranges cut at line ends, spools, rank-ordered concatenation, the routing of lines by query name, the
merge of the workers' outputs by (s1, AS, input order), exit statuses. tests/test_gpu_launcher.py runs the real worker on one GPU."""
import os
import random
import subprocess

import pytest

import oracle_lib as O
import synth_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAFFY = os.environ.get("PAFFY_LAUNCHER") or os.path.join(ROOT, "bin", "paffy")  # tests/test_sanitizers.py points this at the ASan + UBSan build
STANDIN = os.path.join(ROOT, "tests", "standin_worker.py")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s", "../bin/paffy"])


def run(args, n, data=None, tmp=None, **kw):
    env = dict(os.environ, PAFFY_GPUS=str(n), PAFFY_WORKER=STANDIN, PAFFY_ONE_DEVICE="1")
    if tmp:
        env["PAFFY_TMPDIR"] = str(tmp)
    return subprocess.run([PAFFY] + args, input=data, env=env, capture_output=True, timeout=300, **kw)


def tile_records(n, seed=5, contigs=7):
    rng = random.Random(seed)
    out = []
    for r in range(n):
        c = rng.randrange(contigs)
        L = rng.choice([5, 40, 300])
        qs = rng.randrange(0, 2000 - 2 * L - 10)
        tags = [f"AS:i:{rng.choice([5, 5, 80, 900])}"] + ([f"s1:i:{rng.choice([3, 3, 70])}"] if rng.random() < 0.6 else [])
        out.append(f"c{c}\t2000\t{qs}\t{qs + 2 * L + 3}\t{rng.choice('+-')}\tt\t9000\t10\t{10 + 2 * L}\t{L}\t{L}\t60\t" + "\t".join(tags) +
                   f"\tcg:Z:{L}M3I{L}M\n")
    return "".join(out).encode()


@pytest.mark.parametrize("n", [2, 3, 8])
def test_stream_command_over_ranks_is_the_single_process_output(tmp_path, n):
    data = synth_lib.generate(0x5EED0003, 200, 0, 300, threads=1)
    src = tmp_path / "in.paf"
    src.write_bytes(data)
    want, err = O.run([O.stage(O.SHATTER)], data)
    assert err.code == 0
    p = run(["shatter", "-i", str(src), "-o", str(tmp_path / "out.paf")], n, tmp=tmp_path)
    assert p.returncode == 0, p.stderr
    assert (tmp_path / "out.paf").read_bytes() == want
    # stdin -> stdout, long option spelling of nothing: the launcher spools stdin first
    p = run(["invert"], n, data=data, tmp=tmp_path)
    assert p.returncode == 0 and p.stdout == O.run([O.stage(O.INVERT)], data)[0]
    assert [f for f in os.listdir(tmp_path) if f.startswith("paffy.")] == []  # no spool left behind


def test_more_ranks_than_lines_and_empty_input(tmp_path):
    data = synth_lib.generate(0x5EED0003, 50, 0, 3, threads=1)
    p = run(["invert"], 8, data=data, tmp=tmp_path)
    assert p.returncode == 0 and p.stdout == O.run([O.stage(O.INVERT)], data)[0]
    p = run(["invert"], 4, data=b"", tmp=tmp_path)
    assert p.returncode == 0 and p.stdout == b""
    last = data[:-1]  # no newline at the end of the input
    p = run(["invert"], 2, data=last, tmp=tmp_path)
    assert p.returncode == 0 and p.stdout == O.run([O.stage(O.INVERT)], last)[0]


def test_failing_record_ends_the_run_like_one_process(tmp_path):
    good = synth_lib.generate(0x5EED0003, 60, 0, 40, threads=1).splitlines(keepends=True)
    bad = b"q\t10\t0\t5\t*\tt\t10\t0\t5\t5\t5\t60\n"  # strand '*': st_errAbort, exit 1 (impl/paf.c:154-158)
    data = b"".join(good[:25]) + bad + b"".join(good[25:])
    want, err = O.run([O.stage(O.INVERT)], data)
    assert err.code == O.ERR_STRAND
    p = run(["invert"], 4, data=data, tmp=tmp_path)
    assert p.returncode == 1 and p.stdout == want  # everything before the failing record, nothing after it
    assert [f for f in os.listdir(tmp_path) if f.startswith("paffy.")] == []


@pytest.mark.parametrize("n", [2, 3, 5])
def test_tile_over_ranks_is_the_single_process_output(tmp_path, n):
    data = tile_records(600)
    src = tmp_path / "in.paf"
    src.write_bytes(data)
    want, err = O.tile(data)
    assert err.code == 0
    p = run(["tile", "--inputFile", str(src), "--outputFile=" + str(tmp_path / "out.paf")], n, tmp=tmp_path)
    assert p.returncode == 0, p.stderr
    assert (tmp_path / "out.paf").read_bytes() == want
    p = run(["tile"], n, data=data[:-1], tmp=tmp_path)  # stdin, last line without newline
    assert p.returncode == 0 and p.stdout == O.tile(data[:-1])[0]


def test_tile_failure_writes_nothing(tmp_path):
    data = tile_records(100) + b"c1\t2000\t0\t5\t*\tt\t9000\t0\t5\t5\t5\t60\n"
    p = run(["tile"], 3, data=data, tmp=tmp_path)
    assert p.returncode == 1 and p.stdout == b""


def test_one_gpu_and_other_commands_become_the_worker(tmp_path):
    env = dict(os.environ, PAFFY_WORKER="/bin/echo")
    env.pop("PAFFY_GPUS", None)
    p = subprocess.run([PAFFY, "invert", "-i", "x"], env=env, capture_output=True, timeout=30)
    assert p.stdout == b"invert -i x\n"
    p = subprocess.run([PAFFY, "chain", "-i", "x"], env=dict(env, PAFFY_GPUS="4"), capture_output=True, timeout=30)
    assert p.stdout == b"chain -i x\n"  # chain does not shard here: one worker
    p = subprocess.run([PAFFY, "tile", "-h"], env=dict(env, PAFFY_GPUS="4"), capture_output=True, timeout=30)
    assert p.stdout == b"tile -h\n"


@pytest.mark.parametrize("n", [2, 3, 8])
def test_tile_with_fewer_query_names_than_workers(tmp_path, n):
    """one per-contig split of the input (the reference's own workflow, tests/paf_pipeline_test.sh:42-67) has ONE query name: all but one
    worker have nothing to do -- they are not started, and the output is the one worker's"""
    data = tile_records(200, contigs=1)
    src = tmp_path / "in.paf"
    src.write_bytes(data)
    want, err = O.tile(data)
    assert err.code == 0
    p = run(["tile", "-i", str(src)], n, tmp=tmp_path)
    assert p.returncode == 0, p.stderr
    assert p.stdout == want
    assert os.listdir(tmp_path) == ["in.paf"]  # the private spool directory is gone


def test_tile_of_an_empty_input_is_an_empty_output(tmp_path):
    src = tmp_path / "in.paf"
    src.write_bytes(b"")
    p = run(["tile", "-i", str(src), "-o", str(tmp_path / "out.paf")], 3, tmp=tmp_path)
    assert p.returncode == 0, p.stderr
    assert (tmp_path / "out.paf").read_bytes() == b""
    p = run(["tile"], 4, data=b"", tmp=tmp_path)
    assert p.returncode == 0 and p.stdout == b""


def test_option_spellings_mean_what_they_mean_to_the_worker(tmp_path):
    """the launcher parses with getopt_long and the subcommand's own tables: abbreviated long options, a value that looks like an option,
    attached values and clustered flags are what the single-GPU path makes of them"""
    data = synth_lib.generate(0x5EED0003, 100, 0, 120, threads=1)
    src = tmp_path / "in.paf"
    src.write_bytes(data)
    want = O.run([O.stage(O.INVERT)], data)[0]
    for args in (["invert", "--input", str(src)], ["invert", "-l", "-i", "-i" + str(src)], ["invert", "--logLevel=-o", "--inputF=" + str(src)],
                 ["invert", "-lDEBUG", "-i", str(src)]):
        p = run(args, 3, tmp=tmp_path)
        assert p.returncode == 0 and p.stdout == want, args
    # a clustered flag in front of -i: `trim -fi in.paf` (the stand-in knows the identity trim only; what counts here is that the
    # launcher found the input and handed -f on)
    env = dict(os.environ, PAFFY_GPUS="2", PAFFY_WORKER="/bin/echo", PAFFY_TMPDIR=str(tmp_path))
    p = subprocess.run([PAFFY, "trim", "-fi", str(src), "-t", "0.2"], env=env, capture_output=True, timeout=30)
    lines = p.stdout.decode().splitlines()
    assert len(lines) == 2 and all(l.startswith("trim -f -t 0.2 -i " + str(src) + " -o ") for l in lines), p.stdout
    # something getopt_long rejects: one worker gets the command line as it is
    p = subprocess.run([PAFFY, "invert", "-Z", "-i", str(src)], env=env, capture_output=True, timeout=30)
    assert p.stdout == ("invert -Z -i " + str(src) + "\n").encode()


def test_a_signal_takes_the_workers_and_the_spools_along(tmp_path):
    import signal
    import time

    slow = tmp_path / "slow_worker.sh"
    slow.write_text("#!/bin/sh\nsleep 60\n")
    slow.chmod(0o755)
    src = tmp_path / "in.paf"
    src.write_bytes(synth_lib.generate(0x5EED0003, 50, 0, 20, threads=1))
    env = dict(os.environ, PAFFY_GPUS="3", PAFFY_WORKER=str(slow), PAFFY_TMPDIR=str(tmp_path))
    p = subprocess.Popen([PAFFY, "invert", "-i", str(src)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for _ in range(100):
        if any(f.startswith("paffy.") for f in os.listdir(tmp_path)):
            break
        time.sleep(0.05)
    time.sleep(0.2)
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=10) == -signal.SIGTERM
    time.sleep(0.2)
    assert [f for f in os.listdir(tmp_path) if f.startswith("paffy.")] == []
    alive = subprocess.run(["pgrep", "-f", str(slow)], capture_output=True).stdout.split()
    assert alive == []
