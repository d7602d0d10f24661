"""N > 1 path on CPU: world_size-2 gloo processes shard a synthetic stream by batches, run the
CPU oracle as the stand-in worker, exchange only output sizes, and the ordered concatenation of
the per-batch outputs must equal the single-process result."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import synth_lib
from paffy_amd import shard

TOTAL, BATCH = 230, 32
STAGES = [(O.INVERT,), (O.TRIM_IDENTITY,), (O.SHATTER,)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stages = [O.stage(*s) for s in STAGES]
    local, n_batches = {}, (TOTAL + BATCH - 1) // BATCH
    for b, r0, n in shard.batches_of_rank(rank, world, TOTAL, BATCH):
        out, err = O.run(stages, synth_lib.generate(0x5EED0003, 300, r0, n, threads=1))
        assert err.code == 0
        local[b] = out
    sizes = shard.gather_batch_sizes(dist, {b: len(o) for b, o in local.items()}, n_batches)
    offs, total = shard.output_offsets(sizes)
    # every rank writes its own byte ranges of the ordered output (pwrite)
    path = os.path.join(tmpdir, "out.paf")
    if rank == 0:
        with open(path, "wb") as fh:
            fh.truncate(total)
    dist.barrier()
    fd = os.open(path, os.O_WRONLY)
    for b, o in local.items():
        os.pwrite(fd, o, offs[b])
    os.close(fd)
    # the other way to the ordered output: the batches travel to rank 0 in order (gatherv with per-batch sizes)
    chunks = []
    tens = {b: torch.frombuffer(bytearray(o), dtype=torch.uint8) for b, o in local.items()}
    shard.gather_to_writer(dist, rank, world, tens, sizes, lambda b, t: chunks.append((b, bytes(t.numpy().tobytes()))))
    if rank == 0:
        assert [b for b, _ in chunks] == [b for b in range(n_batches) if sizes[b] > 0]
        with open(os.path.join(tmpdir, "gathered.paf"), "wb") as fh:
            fh.write(b"".join(c for _, c in chunks))
    # the bench's timing reduction: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_is_order_preserving(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want, err = O.run([O.stage(*s) for s in STAGES], synth_lib.generate(0x5EED0003, 300, 0, TOTAL, threads=1))
    assert err.code == 0
    assert (tmp_path / "out.paf").read_bytes() == want
    assert (tmp_path / "gathered.paf").read_bytes() == want


def test_batch_partition_covers_stream_once():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            seen += shard.batches_of_rank(r, world, 1000, 64)
        seen.sort()
        assert [b for b, _, _ in seen] == list(range(16))
        assert sum(n for _, _, n in seen) == 1000 and seen[-1][2] == 1000 - 15 * 64


def test_contig_partition_is_balanced_and_total():
    w = {f"chr{i}": 250 - 7 * i for i in range(24)}
    owner = shard.contig_partition(w, 8)
    assert set(owner) == set(w) and set(owner.values()) == set(range(8))
    loads = [sum(w[k] for k in w if owner[k] == r) for r in range(8)]
    assert max(loads) - min(loads) <= max(w.values())


def _tile_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = open(os.path.join(tmpdir, "in.paf"), "rb").read()
    lines = data.splitlines(keepends=True)
    weights = {}
    for ln in lines:
        weights[shard.query_name(ln)] = weights.get(shard.query_name(ln), 0) + len(ln)
    owner = shard.contig_partition(weights, world)
    mine = shard.split_by_owner(lines, owner).get(rank, [])
    # this rank tiles only its contigs (the oracle stands in for the GPU worker); order of equal keys = input order
    out, err = O.tile(b"".join(ln for _, ln in mine))
    assert err.code == 0
    out_lines = out.splitlines(keepends=True)
    # keys of my output lines: my inputs sorted by key give the global indices in output order
    order = sorted(range(len(mine)), key=lambda k: shard.tile_key(mine[k][1], mine[k][0]))
    keyed = [(shard.tile_key(mine[k][1], mine[k][0]), out_lines[pos]) for pos, k in enumerate(order)]
    gathered = [None] * world
    dist.all_gather_object(gathered, keyed)
    if rank == 0:
        with open(os.path.join(tmpdir, "tiled.paf"), "wb") as fh:
            fh.write(b"".join(shard.merge_tiled(gathered)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_equals_single_process(tmp_path):
    import random

    rng = random.Random(5)
    recs = []
    for r in range(400):
        c = rng.randrange(6)
        L = rng.choice([5, 40, 300])
        qs = rng.randrange(0, 2000 - 2 * L - 10)
        tags = [f"AS:i:{rng.choice([5, 5, 80, 900])}"] + ([f"s1:i:{rng.choice([3, 3, 70])}"] if rng.random() < 0.6 else [])
        recs.append(f"c{c}\t2000\t{qs}\t{qs + 2 * L + 3}\t{rng.choice('+-')}\tt\t9000\t10\t{10 + 2 * L}\t{L}\t{L}\t60\t" +
                    "\t".join(tags) + f"\tcg:Z:{L}M3I{L}M\n")
    data = "".join(recs).encode()
    (tmp_path / "in.paf").write_bytes(data)
    mp.spawn(_tile_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want, err = O.tile(data)
    assert err.code == 0
    assert (tmp_path / "tiled.paf").read_bytes() == want
