"""ctypes binding of oracle/libpaf_oracle.so -- the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libpaf_oracle.so")

INVERT, TRIM_IDENTITY, TRIM_FIXED, SHATTER, ADD_MISMATCHES, REMOVE_MISMATCHES, PASS, FILTER = 1, 2, 3, 4, 5, 6, 7, 8
ERR_STRAND, ERR_CHECK_QSTART, ERR_CHECK_QEND, ERR_CHAIN_ASSERT = 2, 5, 6, 22  # PO_ERR_* of oracle/paf_oracle.h


class Stage(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p0", C.c_float), ("p1", C.c_float)]


class Seq(C.Structure):
    _fields_ = [("name", C.c_char_p), ("seq", C.c_char_p), ("len", C.c_int64)]


class Filter(C.Structure):
    _fields_ = [("min_chain_score", C.c_int64), ("min_alignment_score", C.c_int64), ("min_identity", C.c_double),
                ("min_identity_with_gaps", C.c_double), ("max_tile_level", C.c_int64), ("invert", C.c_int32)]


class Error(C.Structure):
    _fields_ = [("code", C.c_int32), ("stage", C.c_int32), ("record", C.c_int64), ("aux", C.c_int64)]


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("paf_oracle.c", "paf_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.po_run.restype = C.c_int
        L.po_run.argtypes = [C.POINTER(Stage), C.c_int32, C.c_char_p, C.c_int64, C.POINTER(Seq), C.c_int64,
                             C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(Error)]
        L.po_tile.restype = C.c_int
        L.po_tile.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(Error)]
        L.po_to_bed.restype = C.c_int
        L.po_to_bed.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(Error)]
        L.po_free.argtypes = [C.c_void_p]
        L.po_split_file.restype = C.c_int
        L.po_split_file.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_int, C.c_int64, C.POINTER(Error)]
        L.po_dedupe.restype = C.c_int
        L.po_dedupe.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(Error)]
        L.po_set_filter.argtypes = [C.POINTER(Filter)]
        L.po_set_filter.restype = None
        L.po_error_exit_status.argtypes = [C.c_int32]
        L.po_cigar_parse.restype = C.c_int64
        L.po_cigar_parse.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int64]
        L.po_cigar_stats.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int]
        L.po_cigar_aligned_bases.restype = C.c_int64
        L.po_cigar_aligned_bases.argtypes = [C.c_char_p]
        L.po_trim_ends_line.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.po_pretty_print.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.po_chain.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(Error)]
        L.po_coverage_counts.restype = C.c_int64
        L.po_coverage_counts.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, C.POINTER(C.c_uint16), C.c_int64]
        _lib = L
    return _lib


def stage(kind, p0=0.05, p1=1.0):
    """p0 = trim -r (trimIdentity, default 0.05), p1 = trim -t (trimFraction, default 1.0)."""
    return Stage(kind, p0, p1)


def _take(ptr, n):
    # string_at takes a C int; shatter outputs pass 2 GiB
    data = bytes((C.c_char * n.value).from_address(ptr.value)) if ptr.value and n.value else b""
    if ptr.value:
        lib().po_free(ptr)
    return data


def run(stages, data, seqs=None):
    """Apply a chain of stream commands; returns (output bytes, Error)."""
    L = lib()
    arr = (Stage * max(1, len(stages)))(*stages)
    sarr, ns = None, 0
    keep = []
    if seqs:
        ns = len(seqs)
        sarr = (Seq * ns)()
        for i, (name, seq) in enumerate(seqs.items()):
            nb = name if isinstance(name, bytes) else name.encode()
            sb = seq if isinstance(seq, bytes) else seq.encode()
            keep += [nb, sb]
            sarr[i] = Seq(nb, sb, len(sb))
    out, n, err = C.c_void_p(), C.c_int64(), Error()
    L.po_run(arr, len(stages), data, len(data), sarr, ns, C.byref(out), C.byref(n), C.byref(err))
    return _take(out, n), err


def set_filter(min_chain_score=-1, min_alignment_score=-1, min_identity=-1.0, min_identity_with_gaps=-1.0, max_tile_level=-1, invert=False):
    """Thresholds of the FILTER stages of later run() calls (`paffy filter -s -t -u -v -w -x`)."""
    f = Filter(min_chain_score, min_alignment_score, min_identity, min_identity_with_gaps, max_tile_level, 1 if invert else 0)
    lib().po_set_filter(C.byref(f))


def filter(data, **thresholds):  # noqa: A001
    set_filter(**thresholds)
    return run([stage(FILTER)], data)


def tile(data):
    L = lib()
    out, n, err = C.c_void_p(), C.c_int64(), Error()
    L.po_tile(data, len(data), C.byref(out), C.byref(n), C.byref(err))
    return _take(out, n), err


def to_bed(data, binary=False, exclude_unaligned=False, exclude_aligned=False, min_size=1, include_inverted=False):
    out, n, err = C.c_void_p(), C.c_int64(), Error()
    lib().po_to_bed(data, len(data), int(binary), int(exclude_unaligned), int(exclude_aligned), min_size, int(include_inverted), C.byref(out), C.byref(n), C.byref(err))
    return _take(out, n), err


def dedupe(data, check_inverse=False):
    L = lib()
    out, n, err = C.c_void_p(), C.c_int64(), Error()
    L.po_dedupe(data, len(data), 1 if check_inverse else 0, C.byref(out), C.byref(n), C.byref(err))
    return _take(out, n), err


def split_file(data, prefix, by_query=False, min_length=0):
    """paffy split_file into files "<prefix>...paf" (the prefix may include a directory); returns the Error."""
    err = Error()
    lib().po_split_file(data, len(data), prefix.encode(), 1 if by_query else 0, min_length, C.byref(err))
    return err


def exit_status(code):
    return lib().po_error_exit_status(code)


def cigar_parse(text):
    cap = max(16, len(text))
    lens, ops = (C.c_int64 * cap)(), (C.c_int32 * cap)()
    n = lib().po_cigar_parse(text.encode() if isinstance(text, str) else text, lens, ops, cap)
    if n < 0:
        return n
    return [(ops[i], lens[i]) for i in range(n)]


def cigar_stats(text, acc=None, zero=True):
    s = (C.c_int64 * 6)(*(acc or [0] * 6))
    rc = lib().po_cigar_stats(text.encode(), s, 1 if zero else 0)
    assert rc == 0
    return list(s)


def aligned_bases(text):
    return lib().po_cigar_aligned_bases(text.encode())


def trim_ends_line(line, end_bases):
    out, n = C.c_void_p(), C.c_int64()
    rc = lib().po_trim_ends_line(line, len(line), end_bases, C.byref(out), C.byref(n))
    return rc, _take(out, n)


def coverage_counts(data, name, length):
    counts = (C.c_uint16 * length)()
    applied = lib().po_coverage_counts(data, len(data), name.encode(), counts, length)
    return applied, list(counts)


def pretty_print(line, query_seq, target_seq, include_alignment=True):
    """paf_pretty_print (impl/paf.c:262-316) of one PAF line."""
    out, n = C.c_void_p(), C.c_int64()
    rc = lib().po_pretty_print(line, len(line), query_seq, target_seq, 1 if include_alignment else 0, C.byref(out), C.byref(n))
    return rc, _take(out, n)


def chain(data, gap_open=5000, gap_extend=1, max_gap=1000000, trim=1.0, fresh_walk=True):
    """`paffy chain` (impl/paf_chain.c defaults). Returns (output, error, fresh_hits). fresh_walk=False: the search without the walk from
    a fresh iterator (impl/chaining.c:74-76) -- the GPU's behaviour on the inputs where that walk sees a candidate."""
    out, n, fresh, err = C.c_void_p(), C.c_int64(), C.c_int64(), Error()
    lib().po_set_chain_fresh_walk(1 if fresh_walk else 0)
    try:
        lib().po_chain(data, len(data), gap_open, gap_extend, max_gap, trim, C.byref(out), C.byref(n), C.byref(fresh), C.byref(err))
    finally:
        lib().po_set_chain_fresh_walk(1)
    return _take(out, n), err, fresh.value
