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
    # No tags: the nccl backend ignores them. A send is matched with a receive by ORDER per peer alone -- every sender posts its batches
    # in ascending batch order and the writer posts its receives in ascending batch order, so the k-th send of rank r meets the k-th
    # receive from rank r on any backend.
    if rank != writer:
        mine = [b for b in range(n_batches) if owner[b] == rank and sizes_by_batch[b] > 0]
        assert mine == sorted(mine)
        reqs = [dist.isend(local[b].contiguous(), dst=writer) for b in mine]
        for r in reqs:
            r.wait()
        return
    ahead, pending = 4, {}

    def post(b):
        if b < n_batches and owner[b] != writer and sizes_by_batch[b] > 0:
            buf = torch.empty(sizes_by_batch[b], dtype=torch.uint8, device=device)
            pending[b] = (dist.irecv(buf, src=owner[b]), buf)  # post() is called with ascending b: ascending per peer too

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


EXCHANGE_CHUNK = 256 << 20  # bytes per send / receive operation


def big_buffer_bytes(n):
    """Size to allocate for a buffer that must hold n bytes of a rank's share (send, receive and output buffers): rounded up to 1/64 of
    its size in 2 MiB units. A second tile of an input of about the same size (another step of the bench, another file of a pipeline)
    then asks torch's caching allocator for the very block it let go -- a buffer a few kilobytes larger than the cached one cannot
    use it, and with a share of 54 GB per buffer the allocator then has to return everything it caches to the driver and allocate
    again (1.5 s per step in the 10 M-record run of round 3)."""
    n = int(n)
    if n < (64 << 20):
        return n
    unit = max(2 << 20, (n >> 6) >> 21 << 21)
    return (n + unit - 1) // unit * unit


def _pairwise_exchange(dist, rank, world, src, send_counts, dst, recv_counts, chunk, recv_offsets=None, verify=True):
    """dst <- what every rank holds for this rank in src (both flat, grouped by peer, counts in elements): the all-to-all as explicit
    sends and receives per peer in pieces of at most `chunk` elements, the rank's own part as a plain copy. Over the nccl backend
    these are RCCL send / recv pairs -- point-to-point over xGMI, which is what the fabric is. Why not all_to_all_single: one call
    with several GB per peer came back without its bytes (tools/probes/rccl_a2a_sizes.py: wrong from 768 MiB on at world size 1)."""
    so, ro = [0], [0]
    for r in range(world):
        so.append(so[-1] + int(send_counts[r]))
        ro.append(ro[-1] + int(recv_counts[r]))
    re = ro[1:]  # where every peer's part ends in dst
    if recv_offsets is not None:  # parts at given places (16-byte aligned segments) instead of back to back
        ro = [int(x) for x in recv_offsets]
        re = [ro[r] + int(recv_counts[r]) for r in range(world)]
    if send_counts[rank]:
        dst[ro[rank]: re[rank]].copy_(src[so[rank]: so[rank + 1]])
    if verify:
        sent = _edge_sums(src, so[:-1], so[1:])
    rounds = max([0] + [(int(c) + chunk - 1) // chunk for r, c in enumerate(send_counts) if r != rank] +
                 [(int(c) + chunk - 1) // chunk for r, c in enumerate(recv_counts) if r != rank])
    for k in range(rounds):
        ops = []
        for step in range(1, world):  # peer order: everyone sends "to the right by step", receives "from the left by step"
            to, frm = (rank + step) % world, (rank - step) % world
            a, b = so[to] + k * chunk, min(so[to] + (k + 1) * chunk, so[to + 1])
            if a < b:
                ops.append(dist.P2POp(dist.isend, src[a:b], to))
            a, b = ro[frm] + k * chunk, min(ro[frm] + (k + 1) * chunk, re[frm])
            if a < b:
                ops.append(dist.P2POp(dist.irecv, dst[a:b], frm))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    if verify:
        # every posted byte arrived: the sender's fingerprint of each part (element count, sums over its first and last 64 Ki elements)
        # travels as a 24-byte all-to-all and must equal what the receiver finds where the part landed. An RCCL collective that returns
        # without its bytes (tools/probes/rccl_a2a_sizes.py saw one do that from 768 MiB per peer on) fails here, loudly, not downstream.
        import torch

        theirs = torch.zeros_like(sent)
        dist.all_to_all_single(theirs, sent)
        got = _edge_sums(dst, ro[:world], re[:world])
        if not torch.equal(theirs.cpu(), got.cpu()):
            bad = [r for r in range(world) if not torch.equal(theirs[r].cpu(), got[r].cpu())]
            raise RuntimeError(f"_pairwise_exchange: rank {rank} did not receive what ranks {bad} sent (fingerprints differ)")


EDGE_ELEMENTS = 1 << 16


def _edge_sums(buf, starts, ends):
    """int64 [parts, 3]: (elements, sum of the first EDGE_ELEMENTS, sum of the last EDGE_ELEMENTS) of buf[start:end] for every part"""
    import torch

    out = torch.zeros(len(starts), 3, dtype=torch.int64, device=buf.device)
    for r, (a, b) in enumerate(zip(starts, ends)):
        a, b = int(a), int(b)
        if b > a:
            out[r, 0] = b - a
            out[r, 1] = buf[a: min(b, a + EDGE_ELEMENTS)].to(torch.int64).sum()
            out[r, 2] = buf[max(a, b - EDGE_ELEMENTS): b].to(torch.int64).sum()
    return out


def exchange_lines(dist, send_buf, send_bytes, send_gidx, send_records, comm_device="cpu", pieces=None):
    """The all-to-all of the partition. send_buf: uint8 tensor holding the lines grouped by destination rank; send_gidx: int64
    tensor, the global input index of every line in the same order; send_bytes / send_records: per destination.
    Returns (recv_buf uint8, recv_gidx int64, pieces): what this rank owns -- the lines of every source rank as one segment of
    recv_buf that starts at a multiple of 16 bytes, pieces = [(offset, bytes)] of the segments in source order. Every source sent its
    lines in input order and sources hold consecutive shares, so the pieces one after the other are in global input order -- and each
    is a whole number of lines at an aligned address: a text batch the library can take where it is, no copy.
    Without ranks nothing moves: the send buffer and the pieces the splitter laid out come back."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    if dist is not None:
        rank = dist.get_rank()
        sizes = torch.tensor([list(send_bytes), list(send_records)], dtype=torch.int64, device=comm_device).t().contiguous()  # [world, 2]
        got = torch.zeros_like(sizes)
        dist.all_to_all_single(got, sizes)  # 16 bytes per peer
        recv_bytes, recv_records = [int(x) for x in got[:, 0].tolist()], [int(x) for x in got[:, 1].tolist()]
    if world == 1:
        # One rank owns everything: the splitter laid every batch's lines out at a multiple of 16 bytes (`pieces`, with up to 15 bytes
        # of padding between them -- the buffer is NOT sum(send_bytes) contiguous bytes), and a rank's own part never travels anyway.
        # With a process group of one (bench.py --force-dist) the size exchange above still ran over the backend.
        n = int(send_bytes[0])
        if dist is not None and (recv_bytes[0] != n or recv_records[0] != int(send_records[0])):
            raise RuntimeError("exchange_lines: the size exchange of a one-rank group returned other sizes than were sent")
        return send_buf, send_gidx[: int(send_records[0])], (pieces if pieces is not None else [(0, n)])
    offs, at = [], 0
    for r in range(world):
        offs.append(at)
        at += (recv_bytes[r] + 15) // 16 * 16
    src = _comm(send_buf[: sum(send_bytes)], comm_device)
    recv_buf = torch.empty(big_buffer_bytes(at + 16), dtype=torch.uint8, device=comm_device)
    _pairwise_exchange(dist, rank, world, src, list(send_bytes), recv_buf, recv_bytes, EXCHANGE_CHUNK, offs)
    for r in range(world):  # the few bytes between a segment's end and the next multiple of 16 are read with the segment's last chunk
        if recv_bytes[r] % 16:
            recv_buf[offs[r] + recv_bytes[r]: offs[r] + (recv_bytes[r] + 15) // 16 * 16] = 0
    gsrc = _comm(send_gidx[: sum(send_records)], comm_device)
    recv_gidx = torch.empty(sum(recv_records), dtype=torch.int64, device=comm_device)
    _pairwise_exchange(dist, rank, world, gsrc, list(send_records), recv_gidx, recv_records, EXCHANGE_CHUNK // 8)
    return recv_buf, recv_gidx, [(offs[r], recv_bytes[r]) for r in range(world) if recv_bytes[r]]


KEY_ROWS_PER_GATHER = 1 << 22  # 128 MiB of keys per rank and collective


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
    counts = [int(x) for x in counts.tolist()]
    cap = max(1, max(counts))
    rows = [torch.empty(counts[r], 4, dtype=torch.int64, device=comm_device) for r in range(world)]
    step = KEY_ROWS_PER_GATHER  # rows of every rank per collective: bounded pieces instead of one multi-GB all-gather
    mine = torch.zeros(min(cap, step), 4, dtype=torch.int64, device=comm_device)
    everyone = torch.zeros(world * mine.shape[0], 4, dtype=torch.int64, device=comm_device)
    for lo in range(0, cap, step):
        k = min(step, cap - lo)
        have = max(0, min(keys.shape[0] - lo, k))
        if have:
            mine[:have] = keys[lo: lo + have]
        dist.all_gather_into_tensor(everyone[: world * k], mine[:k])
        got = everyone[: world * k].reshape(world, k, 4)
        for r in range(world):
            m = max(0, min(counts[r] - lo, k))
            if m:
                rows[r][lo: lo + m] = got[r, :m]
    owner = [torch.full((counts[r],), r, dtype=torch.int64, device=comm_device) for r in range(world)]
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
                if lo2 == at:  # one line longer than max_bytes: take it whole (forward through windows of 1 MiB, never a whole-buffer scan)
                    hi = end
                    while hi < n:
                        hi2 = min(n, hi + (1 << 20))
                        fwd = (buf[hi:hi2] == 10).nonzero()
                        if fwd.numel():
                            hi = hi + int(fwd[0].item()) + 1
                            break
                        hi = hi2
                    end = hi
                    break
                lo = lo2
        cuts.append((at, end))
        at = end
    return cuts


class GpuTileWorker:
    """The device side of a rank in tile_sharded(): everything through the C-ABI of libpaffy_hip (paffy_amd.Engine).

    Memory: a rank never holds its share of the text more than twice. split() writes every batch's lines straight into ONE send
    buffer (paffy_hip_split_to: no per-batch regrouped copy, no concatenation) and, when the caller hands its batches over
    (`consume`), lets go of each batch as soon as it has been split; after the exchange the send buffer is dropped before tile() cuts
    what was received into 16-byte aligned batches of at most batch_bytes, and the receive buffer is dropped once they are cut."""

    def __init__(self, eng, batch_bytes=(1 << 30) + (1 << 29)):
        self.eng, self.batch_bytes, self.keep = eng, batch_bytes, []
        self._recv = self.pieces = None

    def release(self):
        """Let go of the text of the last tile (the received lines stay referenced until then: emit() reads them)."""
        self.keep = []
        self._recv = None
        self.pieces = None

    def query_names(self, batches):
        """-> ({name hash: bytes of its lines} over all batches, [per batch: {name hash: (bytes, lines)}])"""
        self.release()  # a new input: the text of the tile before it is no longer needed (it would stand beside the new send buffer)
        out, per_batch = {}, []
        for buf, nbytes in batches:
            names = self.eng.query_names_counts(buf, nbytes)
            per_batch.append(names)
            for h, (w, _) in names.items():
                out[h] = out.get(h, 0) + w
        return out, per_batch

    def split(self, batches, owner_of, world, first_record, per_batch_names=None, consume=False):
        """-> (send buffer grouped by destination, bytes per destination, global input index of every line, records per destination).
        per_batch_names: what query_names returned for the same batches: the bytes and lines every batch sends to every destination
        are known beforehand, so the send buffer and the index array are allocated once at their exact size and every batch's lines go
        straight to their place (paffy_hip_split_to). consume: `batches` is emptied batch by batch -- the caller's last reference gone,
        a batch is freed as soon as it has been split."""
        t = self.eng.torch
        dev = self.eng.device
        if per_batch_names is None:
            per_batch_names = [self.eng.query_names_counts(buf, nbytes) for buf, nbytes in batches]
        n_batches = len(batches)
        need_b = [[0] * world for _ in range(n_batches)]  # bytes / lines of batch b for destination d
        need_r = [[0] * world for _ in range(n_batches)]
        for b, names in enumerate(per_batch_names):
            for h, (w, r) in names.items():
                d = owner_of.get(h)
                d = min(h % world if d is None else d, world - 1)
                need_b[b][d] += w
                need_r[b][d] += r
        dest_bytes = [sum(need_b[b][d] for b in range(n_batches)) for d in range(world)]
        dest_recs = [sum(need_r[b][d] for b in range(n_batches)) for d in range(world)]
        total_r = sum(dest_recs)
        # One rank: nothing will be exchanged, so every batch's lines start at a multiple of 16 bytes in the buffer -- they are tiled
        # right there as text batches (`pieces`). Several ranks: the parts of a destination stand back to back, as they are sent.
        pad = (lambda x: (x + 15) // 16 * 16) if world == 1 else (lambda x: x)
        total = sum(pad(need_b[b][d]) for b in range(n_batches) for d in range(world))
        send = t.empty(big_buffer_bytes((total + 15) // 16 * 16 + 16), dtype=t.uint8, device=dev)
        gidx = t.empty(max(1, total_r), dtype=t.int64, device=dev)
        at_b, at_r = [0] * world, [0] * world
        for d in range(1, world):
            at_b[d] = at_b[d - 1] + dest_bytes[d - 1]
            at_r[d] = at_r[d - 1] + dest_recs[d - 1]
        arrays = self.eng.owner_arrays(owner_of)
        base = first_record
        pieces = []
        for b in range(n_batches):
            buf, n = batches[0] if consume else batches[b]
            pb, pr, nrec = self.eng.split_to(buf, n, world, arrays, send, at_b, gidx, at_r, base)
            if pb != need_b[b] or pr != need_r[b]:
                raise RuntimeError("split: a batch changed between query_names and split (bytes / lines per destination differ)")
            if world == 1 and pb[0]:
                pieces.append((at_b[0], pb[0]))
                if pb[0] % 16:
                    send[at_b[0] + pb[0]: at_b[0] + pad(pb[0])] = 0
            for d in range(world):
                at_b[d] += pad(pb[d])
                at_r[d] += pr[d]
            base += nrec
            if consume:
                self.eng.sync()  # the copy kernel reads the batch
                del buf
                batches.pop(0)
        self.pieces = pieces if world == 1 else None
        return send[:total], dest_bytes, gidx[:total_r], dest_recs

    def tile(self, recv_buf, pieces=None):
        """paffy tile over the lines this rank owns -> int64 [n, 5] per output line: chain_score, score, local record, bytes, level.
        pieces: [(offset, bytes)] -- whole lines starting at multiples of 16 bytes of recv_buf (what exchange_lines / split lay
        out): they are tiled where they are; a piece too long for a batch, or a buffer without pieces, is cut at line ends into
        16-byte aligned copies. recv_buf may be a one-element list (tile_sharded hands over its only reference)."""
        t = self.eng.torch
        holder = recv_buf if isinstance(recv_buf, list) else [recv_buf]
        recv = holder[0].to(self.eng.device)
        if pieces is None:
            pieces = [(0, int(recv.numel()))]
        self.keep = []
        in_place = recv.data_ptr() % 16 == 0
        for off, n in pieces:
            room = (n + 15) // 16 * 16
            if n == 0:
                continue
            if in_place and off % 16 == 0 and n <= self.batch_bytes and off + room <= recv.numel():
                self.keep.append((recv[off: off + room], n))  # where it was received
                continue
            part = recv[off: off + n]
            for a, b in line_cuts(part, self.batch_bytes):
                piece = t.empty((b - a + 15) // 16 * 16 + 16, dtype=t.uint8, device=self.eng.device)  # batches are 16-byte aligned
                piece[: b - a] = part[a:b]
                piece[b - a:] = 0
                self.keep.append((piece, b - a))
        if isinstance(recv_buf, list):
            recv_buf.clear()
        self._recv = recv if any(x.data_ptr() >= recv.data_ptr() and x.data_ptr() < recv.data_ptr() + max(1, recv.numel()) for x, _ in self.keep) else None
        del recv, holder
        self.info = self.eng.tile_batches(self.keep)
        if self.info.error.code:
            raise RuntimeError(f"tile failed on this rank: code {self.info.error.code} at local record {self.info.error.record}")
        return self.eng.tile_keys(self.info.n_rows)

    def tile_direct(self, batches, consume=False):
        """paffy tile over text batches [(uint8 tensor, bytes)] as they are (16-byte aligned, whole lines, below 2 GiB each): what a
        single rank does, no partition. Same return value as tile(). consume: the list is emptied (the worker holds the batches until
        the lines have been written -- the text is read where it is)."""
        self.release()
        self.keep = [(buf, n) for buf, n in batches if n]
        if consume:
            del batches[:]
        self.info = self.eng.tile_batches(self.keep)
        if self.info.error.code:
            raise RuntimeError(f"tile failed on this rank: code {self.info.error.code} at local record {self.info.error.record}")
        return self.eng.tile_keys(self.info.n_rows)

    def emit(self):
        """-> uint8 tensor with this rank's output lines in its output order"""
        out = self.eng.alloc_out(big_buffer_bytes(self.info.out_bytes))
        if self.info.out_bytes:
            self.eng.emit(out)
        return out[: self.info.out_bytes]

    def scatter(self, src, src_off, dst_off, dst):
        self.eng.scatter_lines(src, src_off, dst_off, dst)


def tile_sharded(worker, dist, rank, world, batches, first_record, comm_device="cpu", consume=False):
    """`paffy tile` over an input spread over the ranks (this rank holds `batches`, whose first record is global record
    first_record). Returns {"offsets": byte offset of every local output line in the ordered output (int64 tensor), "total": bytes
    of the whole output, "keys": the worker's [n, 5] keys}; worker.emit() then gives the local lines.
    consume: the list `batches` is emptied as its batches are split (hand over the only reference and a rank holds its share of the
    text at most twice at any time: batches + send buffer, send + receive buffer, receive buffer + tile batches)."""
    import os
    import time

    import torch

    timing, t0 = {}, time.perf_counter()
    trace = bool(os.environ.get("PAFFY_SHARD_TIMING"))

    peak, live = [0], [0]

    def lap(name):
        nonlocal t0
        if trace:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
                free_b, total_b = torch.cuda.mem_get_info()
                peak[0] = max(peak[0], total_b - free_b)
                # live = in use minus what torch's caching allocator holds without a tensor in it
                live[0] = max(live[0], total_b - free_b - (torch.cuda.memory_reserved() - torch.cuda.memory_allocated()))
            t1 = time.perf_counter()
            timing[name] = timing.get(name, 0.0) + (t1 - t0) * 1e3
            t0 = t1

    if world == 1 and hasattr(worker, "tile_direct"):
        # One rank owns every query sequence: there is nothing to partition, the batches are tiled where they are (the reference's
        # single process, impl/paf_tile.c:156-178). The key exchange and the offsets below are the N-rank code with N = 1.
        keys = worker.tile_direct(batches, consume)
        lap("tile_ms")
        k4 = torch.stack([keys[:, 0], keys[:, 1], keys[:, 2] + first_record, keys[:, 3]], dim=1) if keys.shape[0] else torch.zeros(0, 4, dtype=torch.int64, device=keys.device)
        all_keys, owner = gather_tile_keys(dist, k4, comm_device)
        offsets, total = global_line_offsets(all_keys, owner, rank)
        lap("keys_ms")
        return {"offsets": offsets, "total": total, "keys": keys, "owner_of": None, "timing": timing, "hbm_peak": peak[0] or None, "hbm_live_peak": live[0] or None}
    local, per_batch = worker.query_names(batches)
    weights = merge_name_weights(dist, local, comm_device)
    owner_of = owner_table(weights, world)
    lap("names_ms")
    try:
        send, send_bytes, send_gidx, send_records = worker.split(batches, owner_of, world, first_record, per_batch, consume)
    except Exception:
        worker.eng.drop_index()  # no kept index outlives a failed partition
        raise
    lap("split_ms")
    recv, recv_gidx, pieces = exchange_lines(dist, send, send_bytes, send_gidx, send_records, comm_device, getattr(worker, "pieces", None))
    del send  # with ranks: the receive buffer holds this rank's lines now; without: recv is the send buffer itself
    lap("exchange_ms")
    holder = [recv]
    del recv
    keys = worker.tile(holder, pieces)
    lap("tile_ms")
    gidx = recv_gidx.to(keys.device)
    k4 = torch.stack([keys[:, 0], keys[:, 1], gidx[keys[:, 2]], keys[:, 3]], dim=1) if keys.shape[0] else torch.zeros(0, 4, dtype=torch.int64, device=keys.device)
    all_keys, owner = gather_tile_keys(dist, k4, comm_device)
    offsets, total = global_line_offsets(all_keys, owner, rank)
    lap("keys_ms")
    return {"offsets": offsets, "total": total, "keys": keys, "owner_of": owner_of, "timing": timing, "hbm_peak": peak[0] or None, "hbm_live_peak": live[0] or None}


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
