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


def generate(seed, mean_ops, r0, n, threads=4):
    cfg = Cfg(seed, mean_ops, 24)
    total = lib().psynth_generate(cfg, r0, n, None, 0, None, threads)
    buf = C.create_string_buffer(total)
    lib().psynth_generate(cfg, r0, n, buf, total, None, threads)
    return buf.raw
