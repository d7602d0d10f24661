"""ctypes binding of libpaffy_hip.so plus helpers named after the reference commands.

Names follow the reference CLI (`paffy invert | trim | shatter`, impl/paf_<cmd>.c): a Stage is
one command of a pipe; `trim` takes the reference's -r/-t/-f options. torch is used only for
device buffers and the stream handle.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
_LIB = os.environ.get("PAFFY_HIP_LIB", os.path.join(HERE, "libpaffy_hip.so"))  # override for A/B experiments only

INVERT, TRIM_IDENTITY, TRIM_FIXED, SHATTER, ADD_MISMATCHES, REMOVE_MISMATCHES, PASS, FILTER, TRIM_ENDS, STATS = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10


class BedOpts(C.Structure):
    _fields_ = [("binary", C.c_int32), ("exclude_unaligned", C.c_int32), ("exclude_aligned", C.c_int32), ("include_inverted", C.c_int32),
                ("min_size", C.c_int64)]


class Stage(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p0", C.c_float), ("p1", C.c_float)]


class Filter(C.Structure):
    """Thresholds of `paffy filter` (impl/paf_filter.c:27-32): -s, -t, -u, -v, -w, -x."""
    _fields_ = [("min_chain_score", C.c_int64), ("min_alignment_score", C.c_int64), ("min_identity", C.c_double),
                ("min_identity_with_gaps", C.c_double), ("max_tile_level", C.c_int64), ("invert", C.c_int32)]


class _Error(C.Structure):
    _fields_ = [("code", C.c_int32), ("stage", C.c_int32), ("record", C.c_int64), ("aux", C.c_int64)]


class PlanInfo(C.Structure):
    _fields_ = [("n_records", C.c_int64), ("n_rows", C.c_int64), ("in_bytes", C.c_int64), ("out_bytes", C.c_int64),
                ("error", _Error)]


class ChainOpts(C.Structure):
    _fields_ = [("gap_open", C.c_int64), ("gap_extend", C.c_int64), ("max_gap_length", C.c_int64), ("trim_fraction", C.c_float)]


class PafError(RuntimeError):
    """A record the reference would abort on; .info holds the plan (records before it are emitted)."""

    def __init__(self, msg, info, exit_status):
        super().__init__(msg)
        self.info = info
        self.exit_status = exit_status


def stage(kind, trim_identity=0.05, trim_fraction=1.0):
    """One command of a pipe. trim_identity = `paffy trim -r`, trim_fraction = `-t` (impl/paf_trim.c:14-16)."""
    return Stage(kind, trim_identity, trim_fraction)


def stage_trim_ends(end_bases):
    """paf_trim_ends(paf, end_bases) (impl/paf.c:575-598): the int64 argument travels in the two float slots, bit for bit."""
    import struct

    p0, p1 = struct.unpack("<ff", struct.pack("<q", end_bases))
    st = Stage(TRIM_ENDS, 0.0, 0.0)
    C.memmove(C.addressof(st) + Stage.p0.offset, struct.pack("<ff", p0, p1), 8)  # no float round trip: NaN payloads must survive
    return st


def library_path():
    return _LIB


def build_library(force=False):
    """Compile the gfx950 shared library in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "paffy_hip.h"))
    stale = not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} is missing: run paffy_amd.build_library() (there is no CPU fallback)")
        L = C.CDLL(_LIB)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
        L.paffy_hip_create.argtypes = [C.POINTER(vp), C.c_int]
        L.paffy_hip_destroy.argtypes = [vp]
        L.paffy_hip_set_stream.argtypes = [vp, vp]
        L.paffy_hip_plan.argtypes = [vp, C.POINTER(Stage), i32, vp, i64, C.POINTER(PlanInfo)]
        L.paffy_hip_emit.argtypes = [vp, vp, i64]
        L.paffy_hip_tile_plan.argtypes = [vp, vp, i64, C.POINTER(PlanInfo)]
        L.paffy_hip_tile_begin.argtypes = [vp]
        L.paffy_hip_tile_add.argtypes = [vp, vp, i64]
        L.paffy_hip_tile_run.argtypes = [vp, C.POINTER(PlanInfo)]
        L.paffy_hip_chain_begin.argtypes = [vp]
        L.paffy_hip_chain_add.argtypes = [vp, vp, i64]
        L.paffy_hip_chain_run.argtypes = [vp, C.POINTER(ChainOpts), C.POINTER(PlanInfo)]
        L.paffy_hip_chain_tags.restype = i64
        L.paffy_hip_chain_tags.argtypes = [vp, i64, C.POINTER(i64), C.POINTER(i64)]
        L.paffy_hip_tile_keys.restype = i64
        L.paffy_hip_tile_keys.argtypes = [vp, i64, vp]
        L.paffy_hip_emit_lines.argtypes = [vp, i64, i64, vp, i64, C.POINTER(i64)]
        L.paffy_hip_bed_begin.argtypes = [vp, C.POINTER(BedOpts)]
        L.paffy_hip_bed_add.argtypes = [vp, vp, i64]
        L.paffy_hip_bed_run.argtypes = [vp, C.POINTER(BedOpts), C.POINTER(PlanInfo)]
        L.paffy_hip_query_names.restype = i64
        L.paffy_hip_query_names.argtypes = [vp, vp, i64, i64, C.POINTER(C.c_uint64), C.POINTER(i64)]
        L.paffy_hip_query_names_counts.restype = i64
        L.paffy_hip_query_names_counts.argtypes = [vp, vp, i64, i64, C.POINTER(C.c_uint64), C.POINTER(i64), C.POINTER(i64)]
        L.paffy_hip_split_by_owner.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), i64, vp, i64, C.POINTER(i64), C.POINTER(i64), vp, i64,
                                               C.POINTER(i64)]
        L.paffy_hip_split_to.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), i64, vp, i64, C.POINTER(i64), C.POINTER(i64), i64, C.POINTER(i64),
                                         C.POINTER(i64), vp, i64, C.POINTER(i64)]
        L.paffy_hip_drop_index.argtypes = [vp, vp]
        L.paffy_hip_scatter_lines.argtypes = [vp, vp, vp, vp, i64, vp]
        L.paffy_hip_stream_open.argtypes = [vp, C.POINTER(Stage), i32, i64, i64, C.POINTER(vp)]
        L.paffy_hip_stream_input.restype = vp
        L.paffy_hip_stream_input.argtypes = [vp, i64, i64, C.POINTER(i64)]
        L.paffy_hip_stream_submit.argtypes = [vp, i64, C.POINTER(PlanInfo)]
        L.paffy_hip_stream_read.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
        L.paffy_hip_stream_close.argtypes = [vp]
        L.paffy_hip_sync.argtypes = [vp]
        L.paffy_hip_dedupe_plan.argtypes = [vp, vp, i64, C.c_int, C.POINTER(PlanInfo)]
        L.paffy_hip_dedupe_reset.argtypes = [vp]
        L.paffy_hip_set_sequences.argtypes = [vp, i64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(i64)]
        L.paffy_hip_set_filter.argtypes = [vp, C.POINTER(Filter)]
        L.paffy_hip_error_exit_status.argtypes = [i32]
        L.paffy_hip_error_string.restype = C.c_char_p
        L.paffy_hip_error_string.argtypes = [i32]
        L.paffy_hip_last_error.restype = C.c_char_p
        L.paffy_hip_last_error.argtypes = [vp]
        L.paffy_hip_profile_enable.argtypes = [vp, C.c_int]
        L.paffy_hip_profile_reset.argtypes = [vp]
        L.paffy_hip_profile_only.argtypes = [vp, C.c_char_p]
        L.paffy_hip_stream_trim.argtypes = [vp]
        L.paffy_hip_profile_read.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(i64), C.c_int]
        L.paffy_hip_synth.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, vp, i64, C.POINTER(i64)]
        L.paffy_hip_synth_contigs.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp, i64, C.POINTER(i64)]
        L.paffy_hip_plan_stats.argtypes = [vp, C.POINTER(i64)]
        L.paffy_hip_flat_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
        L.paffy_hip_bed_plan.argtypes = [vp, vp, i64, C.POINTER(BedOpts), C.POINTER(PlanInfo)]
        L.paffy_hip_synth4_setup.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, i64, i64, C.c_int]
        L.paffy_hip_synth4.argtypes = [vp, C.c_uint64, C.c_uint64, vp, i64, C.POINTER(i64)]
        L.paffy_hip_device_count.restype = C.c_int
        _lib = L
    return _lib


def _pad16(n):
    return (n + 15) // 16 * 16 + 16


class Engine:
    """One HIP context (workspace + stream) on the current torch device."""

    def __init__(self, device=None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("paffy_amd needs a GPU: the hot path has no CPU implementation")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._ctx = C.c_void_p()
        rc = lib().paffy_hip_create(C.byref(self._ctx), self.device.index)
        if rc:
            raise RuntimeError(f"paffy_hip_create failed ({rc})")
        self.use_stream(torch.cuda.current_stream(self.device))

    def close(self):
        if self._ctx:
            lib().paffy_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_stream(self, stream):
        self.stream = stream
        lib().paffy_hip_set_stream(self._ctx, C.c_void_p(stream.cuda_stream))

    def _check(self, rc, what):
        if rc:
            raise RuntimeError(f"{what} failed ({rc}): {lib().paffy_hip_last_error(self._ctx).decode()}")

    def set_filter(self, min_chain_score=-1, min_alignment_score=-1, min_identity=-1.0, min_identity_with_gaps=-1.0, max_tile_level=-1,
                   invert=False):
        """Thresholds used by FILTER stages of later plans (`paffy filter -s -t -u -v -w -x`)."""
        f = Filter(min_chain_score, min_alignment_score, min_identity, min_identity_with_gaps, max_tile_level, 1 if invert else 0)
        self._check(lib().paffy_hip_set_filter(self._ctx, C.byref(f)), "paffy_hip_set_filter")

    # ---- device-buffer level (what bench.py times) ----
    def to_device(self, data):
        """bytes -> padded uint8 device tensor (the library reads up to the next multiple of 16)."""
        t = self.torch
        buf = t.zeros(_pad16(len(data)), dtype=t.uint8, device=self.device)
        if len(data):
            buf[: len(data)] = t.frombuffer(bytearray(data), dtype=t.uint8).to(self.device)
        return buf

    def plan(self, stages, d_in, in_len):
        arr = (Stage * max(1, len(stages)))(*stages)
        info = PlanInfo()
        rc = lib().paffy_hip_plan(self._ctx, arr, len(stages), C.c_void_p(d_in.data_ptr()), in_len, C.byref(info))
        self._check(rc, "paffy_hip_plan")
        return info

    def tile_plan(self, d_in, in_len):
        info = PlanInfo()
        self._check(lib().paffy_hip_tile_plan(self._ctx, C.c_void_p(d_in.data_ptr()), in_len, C.byref(info)), "paffy_hip_tile_plan")
        return info

    def tile_batches(self, bufs):
        """paffy tile over an input held as several device batches [(uint8 tensor, nbytes)], each a whole number of lines and
        below 2 GiB; they must stay alive until the output has been emitted. Returns the PlanInfo."""
        self._check(lib().paffy_hip_tile_begin(self._ctx), "paffy_hip_tile_begin")
        for buf, nbytes in bufs:
            self._check(lib().paffy_hip_tile_add(self._ctx, C.c_void_p(buf.data_ptr()), nbytes), "paffy_hip_tile_add")
        info = PlanInfo()
        self._check(lib().paffy_hip_tile_run(self._ctx, C.byref(info)), "paffy_hip_tile_run")
        return info

    def tile_keys(self, n_lines):
        """After a tile plan: int64 tensor [n_lines, 5] on the device -- chain_score, score, input record, line bytes, tile level
        of every output line, in output order (what the ranks of a sharded tile exchange)."""
        keys = self.torch.empty((max(1, n_lines), 5), dtype=self.torch.int64, device=self.device)
        n = lib().paffy_hip_tile_keys(self._ctx, n_lines, C.c_void_p(keys.data_ptr()))
        if n < 0:
            raise RuntimeError(f"paffy_hip_tile_keys failed ({n})")
        return keys[:n]

    # ---- `paffy tile` sharded by query sequence: the device side of the partition and of the ordered write ----
    def query_names(self, d_in, in_len, cap=1 << 20):
        """Distinct query names of a device batch as ({hash: bytes of its lines}): what the partitioner balances."""
        h, w = (C.c_uint64 * cap)(), (C.c_int64 * cap)()
        n = lib().paffy_hip_query_names(self._ctx, C.c_void_p(d_in.data_ptr()), in_len, cap, h, w)
        if n < 0:
            raise RuntimeError(f"paffy_hip_query_names failed ({n}): {lib().paffy_hip_last_error(self._ctx).decode()}")
        return {int(h[i]): int(w[i]) for i in range(n)}

    def query_names_counts(self, d_in, in_len, cap=1 << 20):
        """Distinct query names of a device batch as {hash: (bytes of its lines, number of its lines)}."""
        cap0 = 4096  # the usual input has a few dozen sequences; the host arrays are kept and grown on demand
        while True:
            arrs = getattr(self, "_name_arrays", None)
            if arrs is None or len(arrs[0]) < cap0:
                arrs = self._name_arrays = ((C.c_uint64 * cap0)(), (C.c_int64 * cap0)(), (C.c_int64 * cap0)())
            h, w, r = arrs
            n = lib().paffy_hip_query_names_counts(self._ctx, C.c_void_p(d_in.data_ptr()), in_len, len(h), h, w, r)
            if n == -4 and len(h) < cap:  # PAFFY_E_CAPACITY: more names than the arrays hold
                cap0 = min(cap, len(h) * 16)
                continue
            break
        if n < 0:
            raise RuntimeError(f"paffy_hip_query_names_counts failed ({n}): {lib().paffy_hip_last_error(self._ctx).decode()}")
        return {int(h[i]): (int(w[i]), int(r[i])) for i in range(n)}

    def split_by_owner(self, d_in, in_len, n_parts, owner_of):
        """Lines of a device batch regrouped by owner_of[hash of the query name] (input order inside a part). Returns (uint8 tensor,
        bytes per part, records per part, int64 tensor: batch index of every output line)."""
        t = self.torch
        items = sorted(owner_of.items())
        nt = len(items)
        th = (C.c_uint64 * max(1, nt))(*[k for k, _ in items])
        to = (C.c_uint32 * max(1, nt))(*[v for _, v in items])
        out = t.empty(_pad16(in_len + 1), dtype=t.uint8, device=self.device)
        pb, pr, nrec = (C.c_int64 * n_parts)(), (C.c_int64 * n_parts)(), C.c_int64()
        idx = t.empty(max(1, in_len // 24 + 16), dtype=t.int64, device=self.device)  # a PAF line that parses has at least 24 bytes (the call checks the capacity)
        self._check(lib().paffy_hip_split_by_owner(self._ctx, C.c_void_p(d_in.data_ptr()), in_len, n_parts, th, to, nt, C.c_void_p(out.data_ptr()), out.numel(), pb, pr,
                                                   C.c_void_p(idx.data_ptr()), idx.numel(), C.byref(nrec)), "paffy_hip_split_by_owner")
        return out, list(pb), list(pr), idx[: nrec.value]

    def owner_arrays(self, owner_of):
        """{name hash: part} as the two ctypes arrays the split calls take (ascending hashes), built once per partition."""
        items = sorted(owner_of.items())
        nt = len(items)
        return (C.c_uint64 * max(1, nt))(*[k for k, _ in items]), (C.c_uint32 * max(1, nt))(*[v for _, v in items]), nt

    def split_to(self, d_in, in_len, n_parts, owner_arrays, d_out, part_dst, d_rec_index, rec_dst, rec_base):
        """split_by_owner straight into a send buffer: part p's lines go to d_out[part_dst[p]:], the global index (batch index +
        rec_base) of every one of them to d_rec_index[rec_dst[p]:]. Returns (bytes per part, records per part, records of the batch)."""
        th, to, nt = owner_arrays
        pb, pr, nrec = (C.c_int64 * n_parts)(), (C.c_int64 * n_parts)(), C.c_int64()
        pd, rd = (C.c_int64 * n_parts)(*part_dst), (C.c_int64 * n_parts)(*rec_dst)
        self._check(lib().paffy_hip_split_to(self._ctx, C.c_void_p(d_in.data_ptr()), in_len, n_parts, th, to, nt, C.c_void_p(d_out.data_ptr()), d_out.numel(), pd, rd, rec_base,
                                             pb, pr, C.c_void_p(d_rec_index.data_ptr()), d_rec_index.numel(), C.byref(nrec)), "paffy_hip_split_to")
        return list(pb), list(pr), nrec.value

    def drop_index(self, d_in=None):
        """Forget the line index query_names kept for a batch that will not be split (None: for every batch)."""
        lib().paffy_hip_drop_index(self._ctx, C.c_void_p(d_in.data_ptr()) if d_in is not None else None)

    def scatter_lines(self, d_src, src_off, dst_off, d_dst):
        """Line k = d_src[src_off[k] : src_off[k + 1]] to d_dst[dst_off[k]:] (int64 device tensors)."""
        n = dst_off.numel()
        self._check(lib().paffy_hip_scatter_lines(self._ctx, C.c_void_p(d_src.data_ptr()), C.c_void_p(src_off.data_ptr()), C.c_void_p(dst_off.data_ptr()), n,
                                                  C.c_void_p(d_dst.data_ptr())), "paffy_hip_scatter_lines")

    def emit_lines(self, first, n, d_out):
        """Lines [first, first + n) of a tile / dedupe plan into d_out (from its first byte); returns the bytes written."""
        nbytes = C.c_int64()
        self._check(lib().paffy_hip_emit_lines(self._ctx, first, n, C.c_void_p(d_out.data_ptr()), d_out.numel(), C.byref(nbytes)), "paffy_hip_emit_lines")
        return nbytes.value

    def split_lines(self, data, max_bytes):
        """Cut PAF text into pieces of at most max_bytes that end on line boundaries (one line may exceed it)."""
        out, at = [], 0
        while at < len(data):
            end = min(len(data), at + max_bytes)
            if end < len(data):
                nl = data.rfind(b"\n", at, end)
                end = nl + 1 if nl >= at else (data.find(b"\n", end) + 1 or len(data))
            out.append(data[at:end])
            at = end
        return out

    def tile(self, data, raise_on_error=True, batch_bytes=None):
        """paffy tile (impl/paf_tile.c) over PAF text; returns (output bytes, PlanInfo). With batch_bytes the text goes to the
        device in pieces of at most that size (inputs of 2 GiB and more must)."""
        if batch_bytes:
            bufs = [(self.to_device(p), len(p)) for p in self.split_lines(data, batch_bytes)]
            info = self.tile_batches(bufs)
        else:
            d_in = self.to_device(data)
            info = self.tile_plan(d_in, len(data))
        out = b""
        if info.out_bytes:
            d_out = self.alloc_out(info.out_bytes)
            self.emit(d_out)
            self.sync()
            out = bytes(d_out[: info.out_bytes].cpu().numpy().tobytes())
        if info.error.code and raise_on_error:
            L = lib()
            raise PafError(f"record {info.error.record}: {L.paffy_hip_error_string(info.error.code).decode()}", info,
                           L.paffy_hip_error_exit_status(info.error.code))
        return out, info

    def chain(self, data, gap_open=5000, gap_extend=1, max_gap=1000000, trim=1.0, raise_on_error=True, batch_bytes=None):
        """paffy chain (impl/paf_chain.c, impl/chaining.c) over PAF text; returns (output bytes, PlanInfo). The keyword defaults are
        the command's (-d, -e, -g, -t)."""
        pieces = self.split_lines(data, batch_bytes) if batch_bytes else [data]
        bufs = [(self.to_device(p), len(p)) for p in pieces if len(p)]
        self._check(lib().paffy_hip_chain_begin(self._ctx), "paffy_hip_chain_begin")
        for buf, nbytes in bufs:
            self._check(lib().paffy_hip_chain_add(self._ctx, C.c_void_p(buf.data_ptr()), nbytes), "paffy_hip_chain_add")
        info, opts = PlanInfo(), ChainOpts(gap_open, gap_extend, max_gap, trim)
        self._check(lib().paffy_hip_chain_run(self._ctx, C.byref(opts), C.byref(info)), "paffy_hip_chain_run")
        out = b""
        if info.out_bytes and not info.error.code:
            d_out = self.alloc_out(info.out_bytes)
            self.emit(d_out)
            self.sync()
            out = bytes(d_out[: info.out_bytes].cpu().numpy().tobytes())
        del bufs  # the batches had to stay in place until the lines were written
        if info.error.code and raise_on_error:
            L = lib()
            raise PafError(f"record {info.error.record}: {L.paffy_hip_error_string(info.error.code).decode()}", info,
                           L.paffy_hip_error_exit_status(info.error.code))
        return out, info

    def chain_tags(self, n):
        """(chain ids, chain scores) of the n output lines of the last chain run."""
        ids, scores = (C.c_int64 * max(n, 1))(), (C.c_int64 * max(n, 1))()
        got = lib().paffy_hip_chain_tags(self._ctx, n, ids, scores)
        if got < 0:
            self._check(int(got), "paffy_hip_chain_tags")
        return list(ids[:got]), list(scores[:got])

    def dedupe_plan(self, d_in, in_len, check_inverse=False):
        info = PlanInfo()
        self._check(lib().paffy_hip_dedupe_plan(self._ctx, C.c_void_p(d_in.data_ptr()), in_len, 1 if check_inverse else 0, C.byref(info)),
                    "paffy_hip_dedupe_plan")
        return info

    def dedupe(self, data, check_inverse=False, reset=True, raise_on_error=True):
        """paffy dedupe [-a] (impl/paf_dedupe.c) over PAF text; with reset=False the records written by earlier calls count too."""
        if reset:
            lib().paffy_hip_dedupe_reset(self._ctx)
        d_in = self.to_device(data)
        info = self.dedupe_plan(d_in, len(data), check_inverse)
        out = b""
        if info.out_bytes:
            d_out = self.alloc_out(info.out_bytes)
            self.emit(d_out)
            self.sync()
            out = bytes(d_out[: info.out_bytes].cpu().numpy().tobytes())
        if info.error.code and raise_on_error:
            L = lib()
            raise PafError(f"record {info.error.record}: {L.paffy_hip_error_string(info.error.code).decode()}", info,
                           L.paffy_hip_error_exit_status(info.error.code))
        return out, info

    def emit(self, d_out):
        rc = lib().paffy_hip_emit(self._ctx, C.c_void_p(d_out.data_ptr()), d_out.numel())
        self._check(rc, "paffy_hip_emit")

    def set_sequences(self, seqs):
        """FASTA sequences for add_mismatches: {header: bases}, as `paffy add_mismatches a.fa b.fa` would load them."""
        names = [k if isinstance(k, bytes) else k.encode() for k in seqs]
        vals = [v if isinstance(v, bytes) else v.encode() for v in seqs.values()]
        n = len(names)
        a = (C.c_char_p * max(1, n))(*names)
        b = (C.c_char_p * max(1, n))(*vals)
        ln = (C.c_int64 * max(1, n))(*[len(v) for v in vals])
        self._check(lib().paffy_hip_set_sequences(self._ctx, n, a, b, ln), "paffy_hip_set_sequences")

    def sync(self):
        self._check(lib().paffy_hip_sync(self._ctx), "paffy_hip_sync")

    def alloc_out(self, nbytes):
        return self.torch.empty(_pad16(nbytes), dtype=self.torch.uint8, device=self.device)

    # ---- bytes level (tests, small inputs) ----
    def run(self, stages, data, raise_on_error=True):
        """Apply a pipe of commands to PAF text; returns (output bytes, PlanInfo)."""
        d_in = self.to_device(data)
        info = self.plan(stages, d_in, len(data))
        out = b""
        if info.out_bytes:
            d_out = self.alloc_out(info.out_bytes)
            self.emit(d_out)
            self.sync()
            out = bytes(d_out[: info.out_bytes].cpu().numpy().tobytes())
        if info.error.code and raise_on_error:
            L = lib()
            raise PafError(f"record {info.error.record}: {L.paffy_hip_error_string(info.error.code).decode()}", info,
                           L.paffy_hip_error_exit_status(info.error.code))
        return out, info

    def run_chain(self, stages, data):
        """Run the commands one by one over text, as separate `paffy` processes in a shell pipe would."""
        for st in stages:
            data, _ = self.run([st], data)
        return data

    def synth(self, seed, mean_ops, r0, n, n_contigs=24):
        """Synthetic PAF records [r0, r0+n) generated on the device; returns (tensor, nbytes)."""
        nbytes = C.c_int64()
        self._check(lib().paffy_hip_synth_contigs(self._ctx, seed, mean_ops, n_contigs, r0, n, None, 0, C.byref(nbytes)), "paffy_hip_synth(size)")
        buf = self.torch.zeros(_pad16(nbytes.value), dtype=self.torch.uint8, device=self.device)
        self._check(lib().paffy_hip_synth_contigs(self._ctx, seed, mean_ops, n_contigs, r0, n, C.c_void_p(buf.data_ptr()), buf.numel(), C.byref(nbytes)),
                    "paffy_hip_synth(fill)")
        return buf, nbytes.value

    def to_bed(self, data, binary=False, exclude_unaligned=False, exclude_aligned=False, min_size=1, include_inverted=False, raise_on_error=True,
               batch_bytes=None):
        """paffy to_bed [-b -e -f -m -n] (impl/paf_to_bed.c) over PAF text; returns (BED bytes, PlanInfo). With batch_bytes the text
        goes to the device in pieces of at most that size."""
        info = PlanInfo()
        opts = BedOpts(int(binary), int(exclude_unaligned), int(exclude_aligned), int(include_inverted), min_size)
        if batch_bytes:
            bufs = [(self.to_device(p), len(p)) for p in self.split_lines(data, batch_bytes)]
            self._check(lib().paffy_hip_bed_begin(self._ctx, C.byref(opts)), "paffy_hip_bed_begin")
            for buf, nbytes in bufs:
                self._check(lib().paffy_hip_bed_add(self._ctx, C.c_void_p(buf.data_ptr()), nbytes), "paffy_hip_bed_add")
            self._check(lib().paffy_hip_bed_run(self._ctx, C.byref(opts), C.byref(info)), "paffy_hip_bed_run")
        else:
            d_in = self.to_device(data)
            self._check(lib().paffy_hip_bed_plan(self._ctx, C.c_void_p(d_in.data_ptr()), len(data), C.byref(opts), C.byref(info)), "paffy_hip_bed_plan")
        out = b""
        if info.out_bytes:
            d_out = self.alloc_out(info.out_bytes)
            self.emit(d_out)
            self.sync()
            out = bytes(d_out[: info.out_bytes].cpu().numpy().tobytes())
        if info.error.code and raise_on_error:
            L = lib()
            raise PafError(f"record {info.error.record}: {L.paffy_hip_error_string(info.error.code).decode()}", info,
                           L.paffy_hip_error_exit_status(info.error.code))
        return out, info

    def flat_stats(self):
        """(records the flat sizing pass left to the record kernels in the last plan, or -1 when the plan did not take it; counts per reason)"""
        left, why = C.c_int64(), (C.c_int64 * 16)()
        self._check(lib().paffy_hip_flat_stats(self._ctx, C.byref(left), why), "paffy_hip_flat_stats")
        return left.value, list(why)

    def plan_stats(self):
        """Sums of the STATS stage of the last plan (the last one, should a pipe hold several): (matches, mismatches, inserts, deletes, insert bases, delete bases)."""
        out = (C.c_int64 * 6)()
        self._check(lib().paffy_hip_plan_stats(self._ctx, out), "paffy_hip_plan_stats")
        return tuple(out)

    def synth4_setup(self, seed, mean_ops, n_contigs=24, tlen_min=50_000_000, tlen_span=200_000_000, genomes=True):
        """cfg4 workload: master alignments of n_contigs contig pairs and, with `genomes`, both genomes
        written into the sequence store of this engine (what set_sequences would hold)."""
        self._check(lib().paffy_hip_synth4_setup(self._ctx, seed, mean_ops, n_contigs, tlen_min, tlen_span, 1 if genomes else 0),
                    "paffy_hip_synth4_setup")

    def synth4(self, r0, n):
        """cfg4 records [r0, r0+n) generated on the device; returns (tensor, nbytes)."""
        nbytes = C.c_int64()
        self._check(lib().paffy_hip_synth4(self._ctx, r0, n, None, 0, C.byref(nbytes)), "paffy_hip_synth4(size)")
        buf = self.torch.zeros(_pad16(nbytes.value), dtype=self.torch.uint8, device=self.device)
        self._check(lib().paffy_hip_synth4(self._ctx, r0, n, C.c_void_p(buf.data_ptr()), buf.numel(), C.byref(nbytes)), "paffy_hip_synth4(fill)")
        return buf, nbytes.value

    # ---- host buffers in, host buffers out: the streaming runtime of the CLI ----
    def stream_host(self, stages, chunks, sink=None):
        """Push host chunks (bytes objects of whole lines) through paffy_hip_stream_*: pinned staging, H2D / kernels / D2H
        overlapped. sink(piece_bytes) gets the output pieces in order (default: they are only counted). Returns (records, output bytes)."""
        import time

        L = lib()
        arr = (Stage * max(1, len(stages)))(*stages)
        st = C.c_void_p()
        cap0 = max(4096, max((len(c) for c in chunks), default=4096))
        t_open = time.perf_counter()
        self._check(L.paffy_hip_stream_open(self._ctx, arr, len(stages), cap0, getattr(self, "stream_piece_bytes", 64 << 20), C.byref(st)), "paffy_hip_stream_open")
        # where the time of the call went: opening the stream pins its host buffers (two input slots, three output pieces) and allocates the
        # device buffers -- once per process in the CLI, and seconds on some hosts; the host's copy of a chunk into its pinned slot
        self.stream_seconds = {"open": time.perf_counter() - t_open, "input_copy": 0.0, "run": 0.0, "close": 0.0}
        t_run = time.perf_counter()
        records = out_bytes = 0

        def drain():
            nonlocal out_bytes
            while True:
                piece, n = C.c_void_p(), C.c_int64()
                self._check(L.paffy_hip_stream_read(st, C.byref(piece), C.byref(n)), "paffy_hip_stream_read")
                if n.value == 0:
                    return
                out_bytes += n.value
                if sink:
                    sink(C.string_at(piece.value, n.value))

        try:
            pending = False
            for chunk in chunks:
                cap = C.c_int64()
                buf = L.paffy_hip_stream_input(st, len(chunk), 0, C.byref(cap))
                if not buf:
                    raise RuntimeError("paffy_hip_stream_input: no free slot")
                t_in = time.perf_counter()
                C.memmove(buf, chunk, len(chunk))
                self.stream_seconds["input_copy"] += time.perf_counter() - t_in
                info = PlanInfo()
                self._check(L.paffy_hip_stream_submit(st, len(chunk), C.byref(info)), "paffy_hip_stream_submit")
                if info.error.code:
                    raise PafError(f"record {info.error.record}: {L.paffy_hip_error_string(info.error.code).decode()}", info, L.paffy_hip_error_exit_status(info.error.code))
                records += info.n_records
                if pending:
                    drain()  # the chunk before, while the GPU works on this one
                pending = True
            if pending:
                drain()
            self.stream_seconds["run"] = time.perf_counter() - t_run
        finally:
            t_close = time.perf_counter()
            L.paffy_hip_stream_close(st)
            self.stream_seconds["close"] = time.perf_counter() - t_close
        return records, out_bytes

    # ---- per-kernel HIP-event timing ----
    def profile(self, on=True, only=None):
        """HIP events around every kernel launch, or around the launches of the kernel `only` alone."""
        lib().paffy_hip_profile_only(self._ctx, only.encode() if only else None)
        lib().paffy_hip_profile_enable(self._ctx, 1 if on else 0)
        lib().paffy_hip_profile_reset(self._ctx)

    def profile_read(self):
        cap = 32
        names, ms, cnt = (C.c_char_p * cap)(), (C.c_double * cap)(), (C.c_int64 * cap)()
        n = lib().paffy_hip_profile_read(self._ctx, names, ms, cnt, cap)
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(min(n, cap))}


_default = None


def _engine():
    global _default
    if _default is None:
        _default = Engine()
    return _default


def pipe(stages, data):
    """`paffy a | paffy b | ...` over PAF text (bytes in, bytes out)."""
    return _engine().run(stages, data)[0]


def invert(data):
    """paffy invert (impl/paf_invert.c)."""
    return pipe([stage(INVERT)], data)


def shatter(data):
    """paffy shatter (impl/paf_shatter.c)."""
    return pipe([stage(SHATTER)], data)


def add_mismatches(data, seqs=None, remove=False):
    """paffy add_mismatches [fasta...] / paffy add_mismatches -a (impl/paf_add_mismatches.c)."""
    if remove:
        return pipe([stage(REMOVE_MISMATCHES)], data)
    e = _engine()
    e.set_sequences(seqs)
    return e.run([stage(ADD_MISMATCHES)], data)[0]


def tile(data):
    """paffy tile (impl/paf_tile.c)."""
    return _engine().tile(data)[0]


def chain(data, gap_open=5000, gap_extend=1, max_gap=1000000, trim=1.0):
    """paffy chain [-d gap_open] [-e gap_extend] [-g max_gap] [-t trim] (impl/paf_chain.c)."""
    return _engine().chain(data, gap_open, gap_extend, max_gap, trim)[0]


def trim(data, trim_identity=0.05, trim_fraction=1.0, fixed_trim=False):
    """paffy trim [-r trim_identity] [-t trim_fraction] [-f] (impl/paf_trim.c)."""
    return pipe([stage(TRIM_FIXED if fixed_trim else TRIM_IDENTITY, trim_identity, trim_fraction)], data)


def filter(data, **thresholds):  # noqa: A001 -- named after the reference command
    """paffy filter [-s -t -u -v -w -x] (impl/paf_filter.c); keyword arguments as in Engine.set_filter."""
    e = _engine()
    e.set_filter(**thresholds)
    return e.run([stage(FILTER)], data)[0]


def dedupe(data, check_inverse=False):
    """paffy dedupe [-a] (impl/paf_dedupe.c)."""
    return _engine().dedupe(data, check_inverse)[0]
