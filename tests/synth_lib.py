"""ctypes binding of tools/libpaf_synth.so: host build of the synthetic workload generator."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")
LIB = os.path.join(TOOLS, "libpaf_synth.so")


class Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("mean_ops", C.c_uint32), ("n_contigs", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", TOOLS, "-s"])
        L = C.CDLL(LIB)
        L.psynth_generate.restype = C.c_int64
        L.psynth_generate.argtypes = [C.POINTER(Cfg), C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def generate(seed, mean_ops, r0, n, threads=4, n_contigs=24):
    cfg = Cfg(seed, mean_ops, n_contigs)
    total = lib().psynth_generate(cfg, r0, n, None, 0, None, threads)
    buf = C.create_string_buffer(total)
    lib().psynth_generate(cfg, r0, n, buf, total, None, threads)
    return buf.raw


class Cfg4(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("mean_ops", C.c_uint32), ("n_contigs", C.c_uint32), ("tlen_min", C.c_int64), ("tlen_span", C.c_int64)]


class Synth4:
    """Host build of the cfg4 workload (records on master alignments + the two genomes)."""

    def __init__(self, seed, mean_ops, n_contigs=24, tlen_min=50_000_000, tlen_span=200_000_000):
        L = lib()
        L.psynth4_create.restype = C.c_void_p
        L.psynth4_create.argtypes = [C.POINTER(Cfg4)]
        L.psynth4_destroy.argtypes = [C.c_void_p]
        L.psynth4_contig_len.restype = C.c_int64
        L.psynth4_contig_len.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.psynth4_genome.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        L.psynth4_generate.restype = C.c_int64
        L.psynth4_generate.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_int]
        self.n_contigs = n_contigs
        self._h = L.psynth4_create(Cfg4(seed, mean_ops, n_contigs, tlen_min, tlen_span))
        if not self._h:
            raise ValueError("bad cfg4 parameters")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().psynth4_destroy(self._h)
            self._h = None

    def records(self, r0, n, threads=4):
        total = lib().psynth4_generate(self._h, r0, n, None, 0, threads)
        buf = C.create_string_buffer(total)
        lib().psynth4_generate(self._h, r0, n, buf, total, threads)
        return buf.raw

    def genomes(self, threads=1):
        """{name: bytes} for hs.chr1.. (query) and pt.chr1.. (target). threads > 1: the contigs are generated side by side (the C calls
        release the GIL; a 250 Mb contig takes about 1.4 s on one core, the two 3.6 Gb genomes of cfg4 about 40 s)."""
        jobs = [(g, c, f"{prefix}{c + 1}") for g, prefix in ((0, "hs.chr"), (1, "pt.chr")) for c in range(self.n_contigs)]

        def one(job):
            g, c, name = job
            n = lib().psynth4_contig_len(self._h, g, c)
            buf = (C.c_char * n)()
            lib().psynth4_genome(self._h, g, c, buf)
            return name, bytes(memoryview(buf))

        if threads > 1:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(threads) as pool:
                return dict(pool.map(one, jobs))
        return dict(one(j) for j in jobs)
