"""`python bench.py --gpus N` starts the N ranks itself (the flag used to be ignored). Without a GPU the hot path
cannot run, so the rehearsal mode is used: rendezvous, barriers, the per-step size exchange of the ordered write and the
max-over-ranks reduction run over gloo; the JSON line must say n_gpus = 2."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def test_gpus_flag_starts_that_many_ranks():
    rc, lines, err = _run("--gpus", "2", "--rehearse", "--steps", "4", "--warmup", "0")
    assert rc == 0, err
    assert len(lines) == 1  # rank 0 alone prints
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["rehearsal"] is True
    # both ranks agreed on where every batch lands in the ordered output
    assert line["ordered_write"]["offsets_ok"] is True
    assert line["ordered_write"]["total_bytes"] == sum(1000 * (i + 1) + r for i in range(4) for r in range(2))


def test_three_ranks_and_single_rank():
    rc, lines, err = _run("--gpus", "3", "--rehearse", "--steps", "2")
    assert rc == 0 and lines[0]["n_gpus"] == 3 and lines[0]["ordered_write"]["offsets_ok"], err
    rc, lines, err = _run("--rehearse", "--steps", "2")
    assert rc == 0 and lines[0]["n_gpus"] == 1, err


def test_gpus_beyond_the_visible_devices_is_an_error():
    # no --rehearse / --one-device here: this container has no GPU, so two ranks cannot get a device each
    rc, lines, err = _run("--gpus", "2", "--steps", "1")
    assert rc != 0 and not lines and "device(s) visible" in err


def test_verify_gathers_the_ordered_output_on_rank_zero():
    """`--verify` for the stream workloads (cfg3): the ranks' batch outputs travel to rank 0 in batch order (shard.gather_to_writer) and
    are compared with what one process writes. Without a GPU the rehearsal sends stand-in outputs through the same calls over gloo."""
    for n in ("2", "3"):
        rc, lines, err = _run("--gpus", n, "--rehearse", "--verify", "--workload", "cfg3", "--steps", "3", "--warmup", "0")
        assert rc == 0, err
        assert lines[0]["ordered_write"]["verified_against_one_process"] is True and lines[0]["ordered_write"]["offsets_ok"] is True
    rc, lines, err = _run("--rehearse", "--verify", "--steps", "2")
    assert rc == 0 and lines[0]["ordered_write"]["verified_against_one_process"] is True, err
