"""Sharding of the stream commands across GPUs.

Records of shatter / invert / trim / add_mismatches are independent (each iteration of the
reference loop touches one Paf, impl/paf_invert.c:84-89), so the N-GPU path is a partition of
the record stream with no data-path collective: rank r owns batches r, r+N, r+2N, ... of
`batch` records each. Output order is restored by concatenating per-batch outputs in batch
order; the only exchange is an all-gather of per-batch output byte counts (8 bytes per batch) so
that every rank knows where its bytes go in the ordered output: each rank pwrites its ranges, or the batches travel to one
writer in order (gather_to_writer: point-to-point sends over xGMI with the nccl backend).
`tile` shards by query contig instead (its state is keyed by query name, impl/paf.c:675-688).
"""


def batches_of_rank(rank, world, total_records, batch):
    """[(batch_index, first_record, n_records)] owned by `rank`: round-robin over the stream's batches."""
    out = []
    n_batches = (total_records + batch - 1) // batch
    for b in range(rank, n_batches, world):
        r0 = b * batch
        out.append((b, r0, min(batch, total_records - r0)))
    return out


def output_offsets(sizes_by_batch):
    """Exclusive prefix sum of the per-batch output sizes (index = batch index) -> byte offset of every batch."""
    offs, acc = [], 0
    for s in sizes_by_batch:
        offs.append(acc)
        acc += s
    return offs, acc


def gather_batch_sizes(dist, local, n_batches, device="cpu"):
    """All-gather of {batch_index: out_bytes}: every rank ends with the full size table.

    `local` maps the batch indices this rank processed to their output byte counts. Works with
    any torch.distributed backend (nccl = RCCL on the GPU box, gloo in the CPU tests).
    """
    import torch

    t = torch.zeros(n_batches, dtype=torch.int64, device=device)
    for b, s in local.items():
        t[b] = s
    dist.all_reduce(t, op=dist.ReduceOp.SUM)  # disjoint ownership: a sum is a gather
    return [int(x) for x in t.tolist()]


def gather_to_writer(dist, rank, world, local, sizes_by_batch, write, device="cpu", writer=0):
    """The ordered write: batch outputs travel to the writer rank in batch order (a gatherv with per-batch sizes).

    `local` maps the batch indices this rank owns to uint8 tensors (on `device`; with the nccl backend they stay on the GPU and
    move over xGMI as point-to-point sends, RCCL send/recv); `sizes_by_batch` is the table from gather_batch_sizes. The writer
    calls `write(batch_index, tensor)` for batch 0, 1, 2, ... -- receives are posted a few batches ahead so that the links stay
    busy while the writer drains. Other ranks return once their sends are done.
    """
    import torch

    n_batches = len(sizes_by_batch)
    owner = [b % world for b in range(n_batches)]  # = batches_of_rank's round robin
    if rank != writer:
        reqs = [dist.isend(local[b].contiguous(), dst=writer, tag=b) for b in range(n_batches) if owner[b] == rank and sizes_by_batch[b] > 0]
        for r in reqs:
            r.wait()
        return
    ahead, pending = 4, {}

    def post(b):
        if b < n_batches and owner[b] != writer and sizes_by_batch[b] > 0:
            buf = torch.empty(sizes_by_batch[b], dtype=torch.uint8, device=device)
            pending[b] = (dist.irecv(buf, src=owner[b], tag=b), buf)

    for b in range(min(ahead, n_batches)):
        post(b)
    for b in range(n_batches):
        post(b + ahead)
        if sizes_by_batch[b] == 0:
            continue
        if owner[b] == writer:
            write(b, local[b])
        else:
            req, buf = pending.pop(b)
            req.wait()
            write(b, buf)


def contig_partition(weights, world):
    """`tile`: assign query contigs to ranks, heaviest first onto the lightest rank (weights: name -> work)."""
    loads = [0] * world
    owner = {}
    for name, w in sorted(weights.items(), key=lambda kv: (-kv[1], kv[0])):
        r = min(range(world), key=lambda i: (loads[i], i))
        owner[name] = r
        loads[r] += w
    return owner


# ---- tile across ranks -------------------------------------------------------------------------
#
# `paffy tile` visits all records in (s1 desc, AS desc, input order) order but its state is per
# QUERY sequence, so records of different query sequences never interact: rank r tiles the records
# of the contigs it owns (same relative order), and the single-process output is the merge of the
# per-rank outputs by the same key. Only the keys (24 bytes per record) are all-gathered to build
# the merge; the lines themselves go to the writer (or are pwritten at their offsets).


def query_name(line):
    return line.split(b"\t", 1)[0]


def _tag(line, tag, default):
    i = line.find(b"\t" + tag + b":i:")
    if i < 0:
        return default
    j = line.find(b"\t", i + 1)
    return int(line[i + 6: j if j >= 0 else len(line)])


def tile_key(line, index):
    """Sort key of paf_cmp_by_descending_score (impl/paf_tile.c:28-34) + input index (stable)."""
    return (-_tag(line, b"s1", -1), -_tag(line, b"AS", 0), index)


def split_by_owner(lines, owner):
    """lines: list of PAF lines of the whole input; owner: query name -> rank. Returns {rank: [(global index, line)]}."""
    out = {}
    for i, ln in enumerate(lines):
        out.setdefault(owner[query_name(ln)], []).append((i, ln))
    return out


def merge_tiled(per_rank):
    """per_rank: list over ranks of [(key, line)] in that rank's output order -> lines in global visiting order."""
    import heapq

    return [ln for _, ln in heapq.merge(*per_rank)]
