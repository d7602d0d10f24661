"""The device side of `paffy tile` sharded by query sequence: name hashing, the regrouping of lines by owner (the send
buffer of the all-to-all), the scatter of lines to their place in the ordered output, and shard.tile_sharded with the GPU worker
against the one-process tile."""
import os
import random
import subprocess
import sys
import json

import pytest
import torch

import oracle_lib as O
import synth_lib
from paffy_amd import shard

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    import paffy_amd

    e = paffy_amd.Engine()
    yield e
    e.close()


def records(n, contigs=7, seed=3):
    rng = random.Random(seed)
    out = []
    for r in range(n):
        c = rng.randrange(contigs)
        L = rng.choice([5, 40, 300, 4000])
        qs = rng.randrange(0, 90000 - 2 * L - 10)
        tags = [f"AS:i:{rng.choice([5, 5, 80, 900])}"] + ([f"s1:i:{rng.choice([3, 3, 70])}"] if rng.random() < 0.6 else [])
        out.append(f"contig_{c}\t90000\t{qs}\t{qs + 2 * L + 3}\t{rng.choice('+-')}\tt{r % 3}\t900000\t10\t{10 + 2 * L}\t{L}\t{L}\t60\t" +
                   "\t".join(tags) + f"\tcg:Z:{L}M3I{L}M\n")
    return out


def test_names_split_scatter(eng):
    lines = [l.encode() for l in records(3000)]
    data = b"".join(lines)[:-1]  # the last line without its newline: it still is a record, and leaves with one
    d_in = eng.to_device(data)
    names = eng.query_names(d_in, len(data))
    want = {}
    for ln in lines:
        h = shard.name_hash(shard.query_name(ln))
        want[h] = want.get(h, 0) + len(ln)
    assert names == want
    owner_of = shard.owner_table(names, 3)
    out, pb, pr, ridx = eng.split_by_owner(d_in, len(data), 3, owner_of)
    parts = [[] for _ in range(3)]
    for i, ln in enumerate(lines):
        parts[owner_of[shard.name_hash(shard.query_name(ln))]].append((i, ln))
    assert pr == [len(p) for p in parts] and pb == [sum(len(ln) for _, ln in p) for p in parts]
    assert bytes(out[: sum(pb)].cpu().numpy().tobytes()) == b"".join(ln for p in parts for _, ln in p)
    assert ridx.cpu().tolist() == [i for p in parts for i, _ in p]
    # a name the table does not know goes to hash % parts; one part only = the input (with the last newline added)
    out1, pb1, pr1, ridx1 = eng.split_by_owner(d_in, len(data), 1, {})
    assert bytes(out1[: pb1[0]].cpu().numpy().tobytes()) == b"".join(lines) and ridx1.cpu().tolist() == list(range(len(lines)))
    # scatter: the lines back into input order
    sizes = torch.tensor([len(ln) for p in parts for _, ln in p], dtype=torch.int64, device=eng.device)
    src_off = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=eng.device)
    src_off[1:] = torch.cumsum(sizes, 0)
    starts, at = [], 0
    for ln in lines:
        starts.append(at)
        at += len(ln)
    dst_off = torch.tensor([starts[i] for p in parts for i, _ in p], dtype=torch.int64, device=eng.device)
    dst = eng.alloc_out(at)
    eng.scatter_lines(out, src_off, dst_off, dst)
    eng.sync()
    assert bytes(dst[:at].cpu().numpy().tobytes()) == b"".join(lines)


def test_tile_sharded_single_rank_equals_tile(eng, human_chimp):
    data = human_chimp + "".join(records(1500)).encode() + synth_lib.generate(0x5EED0005, 300, 0, 400)
    want, err = O.tile(data)
    assert err.code == 0
    worker = shard.GpuTileWorker(eng, batch_bytes=300_000)  # what the rank owns is tiled as several text batches
    pieces = [(eng.to_device(p), len(p)) for p in eng.split_lines(data, 500_000)]
    res = shard.tile_sharded(worker, None, 0, 1, pieces, 0, eng.device)
    out = worker.emit()
    eng.sync()
    assert res["total"] == len(want) and bytes(out.cpu().numpy().tobytes()) == want
    whole = shard.gather_ordered_output(worker, None, 0, 1, out, res["keys"][:, 3].contiguous(), res["offsets"], res["total"], eng.device)
    eng.sync()
    assert bytes(whole.cpu().numpy().tobytes()) == want


def test_two_ranks_on_one_gpu_through_bench():
    """bench.py --workload cfg5 with two ranks sharing this GPU (gloo carries the exchanges): the gathered ordered output of the
    sharded run must equal a one-process tile of the same records (--verify)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg5", "--gpus", "2", "--one-device", "--dist-backend", "gloo", "--batch", "3000",
                        "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--verify"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ordered_write"]["verified_against_one_process"] is True
    assert line["config"]["records_timed"] == 2 * 3000 * 2


def test_split_to_fills_one_send_buffer(eng):
    """paffy_hip_split_to: the parts of several batches land behind each other per destination in ONE buffer (what the all-to-all
    sends), the global record indices likewise; a destination range that is too small is refused before anything is written."""
    lines = [l.encode() for l in records(2000, contigs=9, seed=8)]
    batches = [lines[:700], lines[700:1500], lines[1500:]]
    bufs = [(eng.to_device(b"".join(b)), sum(len(x) for x in b)) for b in batches]
    worker = shard.GpuTileWorker(eng)
    names, per_batch = worker.query_names(bufs)
    owner_of = shard.owner_table(names, 3)
    held = list(bufs)
    send, nbytes, gidx, nrec = worker.split(held, owner_of, 3, 1000, per_batch, consume=True)
    eng.sync()
    assert held == []  # handed over and let go
    parts = [[] for _ in range(3)]
    for i, ln in enumerate(lines):
        parts[owner_of[shard.name_hash(shard.query_name(ln))]].append((1000 + i, ln))
    assert nbytes == [sum(len(ln) for _, ln in p) for p in parts] and nrec == [len(p) for p in parts]
    assert bytes(send.cpu().numpy().tobytes()) == b"".join(ln for p in parts for _, ln in p)
    assert gidx.cpu().tolist() == [g for p in parts for g, _ in p]
    # a destination one byte short: PAFFY_E_CAPACITY (-3), nothing kept behind
    d_in, n = bufs[0]
    arrays = eng.owner_arrays(owner_of)
    small = torch.empty(64, dtype=torch.uint8, device=eng.device)
    idx = torch.empty(4096, dtype=torch.int64, device=eng.device)
    with pytest.raises(RuntimeError):
        eng.split_to(d_in, n, 3, arrays, small, [0, 16, 32], idx, [0, 1000, 2000], 0)


def test_kept_index_is_dropped_on_request(eng):
    """query_names keeps the batch's line index for the split; a batch that is refilled instead must be re-indexed: drop_index (the
    contract of include/paffy_hip.h)."""
    lines = [l.encode() for l in records(300, contigs=4, seed=2)]
    data = b"".join(lines)
    d_in = eng.to_device(data)
    eng.query_names(d_in, len(data))
    other = b"".join(reversed(lines))  # same bytes in total, other line starts
    assert len(other) == len(data)
    d_in[: len(other)] = torch.frombuffer(bytearray(other), dtype=torch.uint8).to(eng.device)
    eng.drop_index(d_in)
    out, pb, pr, ridx = eng.split_by_owner(d_in, len(other), 1, {})
    assert bytes(out[: pb[0]].cpu().numpy().tobytes()) == other


def _bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_rccl_runs_at_world_size_one():
    """The collectives of the N-rank path over RCCL (torch's nccl backend), executed for real at world size 1 (--force-dist): the
    per-step all_gather_into_tensor of the stream commands on its side stream, and for tile the all-gathers of names and keys and the
    uneven all_to_all_single of the lines on device tensors. Two ranks cannot share one GPU under RCCL, so this is as far as one GPU goes."""
    line = _bench("--force-dist", "--workload", "cfg3", "--batch", "4096", "--steps", "3", "--warmup", "1", "--cpu-sample", "0")
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["ordered_write"]["total_bytes"] > 0
    assert line["ordered_write"]["rank0_first_offsets"][0] == 0 and line["ordered_write"]["rank0_first_offsets"][1] > 0
    line = _bench("--force-dist", "--workload", "cfg5", "--batch", "3000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--verify")
    assert line["n_gpus"] == 1 and line["ordered_write"]["verified_against_one_process"] is True
    # several text batches per rank (their lengths are no multiples of 16: the splitter pads between them): a one-rank group once treated
    # the padded send buffer as contiguous text (the 10 M-record run of round 3 failed on its last record)
    line = _bench("--force-dist", "--workload", "cfg5", "--batch", "3000", "--tile-text-batch", "700", "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--verify")
    assert line["n_gpus"] == 1 and line["ordered_write"]["verified_against_one_process"] is True
