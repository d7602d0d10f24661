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
# `paffy tile` visits all records in (s1 desc, AS desc, input order) order but its state is per QUERY sequence, so records of
# different query sequences never interact (impl/paf_tile.c:164-175, impl/paf.c:675-709). Sharded over N ranks:
#   1. every rank holds a contiguous share of the input; the distinct query names and the bytes of their lines are merged over the
#      ranks (merge_name_weights: one all-gather of a few hundred bytes) and dealt to the ranks, heaviest first (owner_table);
#   2. exchange_lines: the lines travel to the owner of their query sequence -- ONE all-to-all of the text (RCCL over xGMI with the
#      nccl backend) plus one of the records' global input indices (8 bytes per record);
#   3. every rank tiles what it received (its sequences, complete, in input order);
#   4. gather_tile_keys: an all-gather of (chain_score, score, global input index, line bytes) -- 32 bytes per record -- lets every
#      rank order ALL records as the single process would and scan their sizes: global_line_offsets gives every local line its
#      byte offset in the ordered output. The lines themselves are written at those offsets (scatter) or stay where they are.
# Everything here is tensor plumbing over torch.distributed; the device work (hashing, regrouping, tiling, writing) is the C-ABI's.


def name_hash(name):
    """The 64-bit hash the device uses for a sequence name (cov_name_hash, coverage_kernel.h): FNV-1a over the bytes, then the length."""
    h, mask = 0xcbf29ce484222325, (1 << 64) - 1
    for b in name:
        h = ((h ^ b) * 0x100000001b3) & mask
    h = ((h ^ (0x100 + len(name))) * 0x100000001b3) & mask
    return h ^ (h >> 29)


def _comm(t, comm_device):
    return t if str(t.device) == str(comm_device) else t.to(comm_device)


def merge_name_weights(dist, local, comm_device="cpu"):
    """local: {name hash: weight} of this rank's share -> the same table summed over all ranks (identical on every rank)."""
    import torch

    if dist is None:
        return dict(local)
    world = dist.get_world_size()
    n = torch.tensor([len(local)], dtype=torch.int64, device=comm_device)
    counts = torch.zeros(world, dtype=torch.int64, device=comm_device)
    dist.all_gather_into_tensor(counts, n)
    cap = max(1, int(counts.max().item()))
    mine = torch.zeros(cap, 2, dtype=torch.int64, device=comm_device)
    if local:
        keys = sorted(local)
        mine[: len(keys), 0] = torch.tensor([k - (1 << 64) if k >= (1 << 63) else k for k in keys], dtype=torch.int64)  # hashes as two's complement
        mine[: len(keys), 1] = torch.tensor([local[k] for k in keys], dtype=torch.int64)
    everyone = torch.zeros(world * cap, 2, dtype=torch.int64, device=comm_device)
    dist.all_gather_into_tensor(everyone, mine)
    everyone = everyone.cpu().reshape(world, cap, 2)
    out = {}
    for r in range(world):
        for k, w in everyone[r, : int(counts[r].item())].tolist():
            k &= (1 << 64) - 1
            out[k] = out.get(k, 0) + w
    return out


def owner_table(weights, world):
    """{name hash: weight} -> {name hash: owning rank}: contig_partition's deal, the same on every rank."""
    return contig_partition(weights, world)


def exchange_lines(dist, send_buf, send_bytes, send_gidx, send_records, comm_device="cpu"):
    """The all-to-all of the partition. send_buf: uint8 tensor whose first sum(send_bytes) bytes are the lines grouped by destination
    rank; send_gidx: int64 tensor, the global input index of every line in the same order; send_bytes / send_records: per destination.
    Returns (recv_buf uint8, recv_gidx int64, recv_bytes per source, recv_records per source): what this rank owns, grouped by
    source rank -- and since every source sent its lines in input order and sources hold consecutive shares, in global input order."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    if dist is None:
        n = int(send_bytes[0])
        return send_buf[:n], send_gidx[: int(send_records[0])], [n], [int(send_records[0])]
    sizes = torch.tensor([list(send_bytes), list(send_records)], dtype=torch.int64, device=comm_device).t().contiguous()  # [world, 2]
    got = torch.zeros_like(sizes)
    dist.all_to_all_single(got, sizes)
    recv_bytes, recv_records = [int(x) for x in got[:, 0].tolist()], [int(x) for x in got[:, 1].tolist()]
    src = _comm(send_buf[: sum(send_bytes)], comm_device)
    recv_buf = torch.empty(sum(recv_bytes), dtype=torch.uint8, device=comm_device)
    dist.all_to_all_single(recv_buf, src, recv_bytes, list(send_bytes))
    gsrc = _comm(send_gidx[: sum(send_records)], comm_device)
    recv_gidx = torch.empty(sum(recv_records), dtype=torch.int64, device=comm_device)
    dist.all_to_all_single(recv_gidx, gsrc, recv_records, list(send_records))
    return recv_buf, recv_gidx, recv_bytes, recv_records


def gather_tile_keys(dist, keys, comm_device="cpu"):
    """keys: int64 [n_local, 4] = (chain_score, score, global input index, line bytes) of this rank's output lines, in its output
    order. Returns (all keys [N, 4], owner rank of every row [N]) on comm_device: an all-gather padded to the longest share."""
    import torch

    if dist is None:
        return keys, torch.zeros(keys.shape[0], dtype=torch.int64, device=keys.device)
    world = dist.get_world_size()
    keys = _comm(keys.contiguous(), comm_device)
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=comm_device)
    counts = torch.zeros(world, dtype=torch.int64, device=comm_device)
    dist.all_gather_into_tensor(counts, n)
    cap = max(1, int(counts.max().item()))
    mine = torch.zeros(cap, 4, dtype=torch.int64, device=comm_device)
    mine[: keys.shape[0]] = keys
    everyone = torch.zeros(world * cap, 4, dtype=torch.int64, device=comm_device)
    dist.all_gather_into_tensor(everyone, mine)
    everyone = everyone.reshape(world, cap, 4)
    rows = [everyone[r, : int(counts[r].item())] for r in range(world)]
    owner = [torch.full((int(counts[r].item()),), r, dtype=torch.int64, device=comm_device) for r in range(world)]
    return torch.cat(rows), torch.cat(owner)


def global_line_offsets(all_keys, owner, rank):
    """The single-process order of ALL records -- chain_score desc, score desc, input index (paf_cmp_by_descending_score,
    impl/paf_tile.c:28-34, ties in input order) -- and the scan of their line sizes. Returns (byte offset of every line of `rank`, in
    that rank's own output order; total bytes). Three stable sorts, least significant key first (torch.sort on whatever device the
    keys are on)."""
    import torch

    order = torch.sort(all_keys[:, 2], stable=True).indices
    order = order[torch.sort(all_keys[order, 1], descending=True, stable=True).indices]
    order = order[torch.sort(all_keys[order, 0], descending=True, stable=True).indices]
    sizes = all_keys[order, 3]
    ends = torch.cumsum(sizes, 0)
    offs = ends - sizes
    mine = owner[order] == rank  # a rank's lines keep their relative order in the global order: same comparator, disjoint sequences
    return offs[mine], int(ends[-1].item()) if ends.numel() else 0


def share_of_rank(rank, world, total):
    """Contiguous share [first, first + n) of `total` records held by `rank` before the partition."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def query_name(line):
    return line.split(b"\t", 1)[0]


def line_cuts(buf, max_bytes):
    """Cut points of a uint8 tensor of '\\n'-terminated lines into pieces of at most max_bytes that end on line boundaries
    (a longer single line stays whole). Returns [(start, end), ...]."""
    n = int(buf.numel())
    cuts, at = [], 0
    while at < n:
        end = min(n, at + max_bytes)
        if end < n:
            lo = end
            while True:  # last newline in front of `end`, looking back through windows of 64 KiB
                lo2 = max(at, lo - 65536)
                hits = (buf[lo2:lo] == 10).nonzero()
                if hits.numel():
                    end = lo2 + int(hits[-1].item()) + 1
                    break
                if lo2 == at:  # one line longer than max_bytes: take it whole
                    fwd = (buf[end:] == 10).nonzero()
                    end = end + int(fwd[0].item()) + 1 if fwd.numel() else n
                    break
                lo = lo2
        cuts.append((at, end))
        at = end
    return cuts


class GpuTileWorker:
    """The device side of a rank in tile_sharded(): everything through the C-ABI of libpaffy_hip (paffy_amd.Engine)."""

    def __init__(self, eng, batch_bytes=(1 << 30) + (1 << 29)):
        self.eng, self.batch_bytes, self.keep = eng, batch_bytes, []

    def query_names(self, batches):
        out = {}
        for buf, nbytes in batches:
            for h, w in self.eng.query_names(buf, nbytes).items():
                out[h] = out.get(h, 0) + w
        self._last_named = batches[-1][0].data_ptr() if batches else None
        return out

    def split(self, batches, owner_of, world, first_record):
        """-> (send buffer grouped by destination, bytes per destination, global input index of every line, records per destination)"""
        t = self.eng.torch
        pieces, idx = [[] for _ in range(world)], [[] for _ in range(world)]
        nbytes, nrec, base = [0] * world, [0] * world, first_record
        for buf, n in batches:
            out, pb, pr, ridx = self.eng.split_by_owner(buf, n, world, owner_of)
            b0 = r0 = 0
            for d in range(world):
                pieces[d].append(out[b0: b0 + pb[d]])
                idx[d].append(ridx[r0: r0 + pr[d]] + base)
                b0, r0 = b0 + pb[d], r0 + pr[d]
                nbytes[d] += pb[d]
                nrec[d] += pr[d]
            base += int(ridx.numel())
        flat = [p for d in range(world) for p in pieces[d]]
        flat_i = [p for d in range(world) for p in idx[d]]
        send = t.cat(flat) if flat else t.empty(0, dtype=t.uint8, device=self.eng.device)
        gidx = t.cat(flat_i) if flat_i else t.empty(0, dtype=t.int64, device=self.eng.device)
        return send, nbytes, gidx, nrec

    def tile(self, recv_buf):
        """paffy tile over the lines this rank owns -> int64 [n, 5] per output line: chain_score, score, local record, bytes, level"""
        t = self.eng.torch
        recv = recv_buf.to(self.eng.device)
        self.keep = []
        for a, b in line_cuts(recv, self.batch_bytes):
            piece = t.zeros((b - a + 15) // 16 * 16 + 16, dtype=t.uint8, device=self.eng.device)  # batches are 16-byte aligned
            piece[: b - a] = recv[a:b]
            self.keep.append((piece, b - a))
        self.info = self.eng.tile_batches(self.keep)
        if self.info.error.code:
            raise RuntimeError(f"tile failed on this rank: code {self.info.error.code} at local record {self.info.error.record}")
        return self.eng.tile_keys(self.info.n_rows)

    def emit(self):
        """-> uint8 tensor with this rank's output lines in its output order"""
        out = self.eng.alloc_out(self.info.out_bytes)
        if self.info.out_bytes:
            self.eng.emit(out)
        return out[: self.info.out_bytes]

    def scatter(self, src, src_off, dst_off, dst):
        self.eng.scatter_lines(src, src_off, dst_off, dst)


def tile_sharded(worker, dist, rank, world, batches, first_record, comm_device="cpu"):
    """`paffy tile` over an input spread over the ranks (this rank holds `batches`, whose first record is global record
    first_record). Returns {"offsets": byte offset of every local output line in the ordered output (int64 tensor), "total": bytes
    of the whole output, "keys": the worker's [n, 5] keys}; worker.emit() then gives the local lines."""
    import os
    import time

    import torch

    timing, t0 = {}, time.perf_counter()
    trace = bool(os.environ.get("PAFFY_SHARD_TIMING"))

    def lap(name):
        nonlocal t0
        if trace:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            timing[name] = timing.get(name, 0.0) + (t1 - t0) * 1e3
            t0 = t1

    weights = merge_name_weights(dist, worker.query_names(batches), comm_device)
    owner_of = owner_table(weights, world)
    lap("names_ms")
    send, send_bytes, send_gidx, send_records = worker.split(batches, owner_of, world, first_record)
    lap("split_ms")
    recv, recv_gidx, _, _ = exchange_lines(dist, send, send_bytes, send_gidx, send_records, comm_device)
    lap("exchange_ms")
    keys = worker.tile(recv)
    lap("tile_ms")
    gidx = recv_gidx.to(keys.device)
    k4 = torch.stack([keys[:, 0], keys[:, 1], gidx[keys[:, 2]], keys[:, 3]], dim=1) if keys.shape[0] else torch.zeros(0, 4, dtype=torch.int64, device=keys.device)
    all_keys, owner = gather_tile_keys(dist, k4, comm_device)
    offsets, total = global_line_offsets(all_keys, owner, rank)
    lap("keys_ms")
    return {"offsets": offsets, "total": total, "keys": keys, "owner_of": owner_of, "timing": timing}


def gather_ordered_output(worker, dist, rank, world, lines, line_bytes, offsets, total, comm_device="cpu", writer=0):
    """The ordered write for outputs that fit one rank: every rank's lines (uint8 tensor `lines`, sizes `line_bytes`, places
    `offsets`) travel to `writer`, which scatters them to their offsets. Returns the whole output on the writer, None elsewhere."""
    import torch

    if dist is None:
        src_off = torch.zeros(line_bytes.numel() + 1, dtype=torch.int64, device=lines.device)
        src_off[1:] = torch.cumsum(line_bytes, 0)
        out = torch.zeros(max(16, (total + 15) // 16 * 16), dtype=torch.uint8, device=lines.device)
        worker.scatter(lines, src_off, offsets.to(lines.device), out)
        return out[:total]
    meta = torch.tensor([int(lines.numel()), int(line_bytes.numel())], dtype=torch.int64, device=comm_device)
    metas = torch.zeros(world * 2, dtype=torch.int64, device=comm_device)
    dist.all_gather_into_tensor(metas, meta)
    metas = metas.reshape(world, 2)
    cap_b, cap_n = max(1, int(metas[:, 0].max().item())), max(1, int(metas[:, 1].max().item()))
    pad_b = torch.zeros(cap_b, dtype=torch.uint8, device=comm_device)
    pad_b[: lines.numel()] = _comm(lines, comm_device)
    pad_n = torch.zeros(cap_n, 2, dtype=torch.int64, device=comm_device)
    pad_n[: line_bytes.numel(), 0] = _comm(line_bytes, comm_device)
    pad_n[: offsets.numel(), 1] = _comm(offsets, comm_device)
    all_b = [torch.zeros(cap_b, dtype=torch.uint8, device=comm_device) for _ in range(world)] if rank == writer else None
    all_n = [torch.zeros(cap_n, 2, dtype=torch.int64, device=comm_device) for _ in range(world)] if rank == writer else None
    dist.gather(pad_b, all_b, dst=writer)
    dist.gather(pad_n, all_n, dst=writer)
    if rank != writer:
        return None
    dev = lines.device
    out = torch.zeros(max(16, (total + 15) // 16 * 16), dtype=torch.uint8, device=dev)
    for r in range(world):
        nb, nl = int(metas[r, 0].item()), int(metas[r, 1].item())
        if nl == 0:
            continue
        sizes = all_n[r][:nl, 0].to(dev)
        src_off = torch.zeros(nl + 1, dtype=torch.int64, device=dev)
        src_off[1:] = torch.cumsum(sizes, 0)
        worker.scatter(all_b[r][:nb].to(dev), src_off, all_n[r][:nl, 1].to(dev).contiguous(), out)
    return out[:total]
