"""The per-record C API (include/paf.h, lib/libstPaf_hip.so): symbols on the CPU, known answers and file round trips on the GPU."""
import os
import re
import subprocess

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "lib", "libstPaf_hip.so")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "paffy_amd", "csrc"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])


def test_library_exports_the_declared_api():
    """Every function include/paf.h declares outside PAFFY_WITH_SONLIB is defined by the library (no compute here)."""
    _build()
    hdr = open(os.path.join(ROOT, "include", "paf.h")).read()
    hdr = hdr[: hdr.index("#ifdef PAFFY_WITH_SONLIB")] + hdr[hdr.index("#endif", hdr.index("#ifdef PAFFY_WITH_SONLIB")):]
    names = set(re.findall(r"^[A-Za-z_][\w \*]*?\b(\w+)\(", hdr, flags=re.M)) - {"cigar_count", "cigar_get"}
    assert {"paf_parse", "paf_invert", "paf_shatter_array", "paf_encode_mismatches", "paf_trim_unreliable_tails", "cigar_parse"} <= names
    syms = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    defined = {line.split()[-1] for line in syms.splitlines() if " T " in line}
    assert names <= defined, names - defined
    # the API file holds no record logic of its own: it must not link or include the oracle
    src = open(os.path.join(ROOT, "host", "paf_api.c")).read()
    assert "oracle" not in src


@pytest.mark.gpu
def test_known_answers_and_file_round_trip(tmp_path, human_chimp):
    _build()
    exe = tmp_path / "paf_api_kat"
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-Wall", "-o", str(exe), os.path.join(ROOT, "tests", "c", "paf_api_kat.c"),
                           "-L" + os.path.join(ROOT, "lib"), "-lstPaf_hip", "-Wl,-rpath," + os.path.join(ROOT, "lib"),
                           "-Wl,-rpath," + os.path.join(ROOT, "paffy_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    out = tmp_path / "out.paf"
    r = subprocess.run([str(exe), os.path.join(ROOT, "tests", "golden", "human_chimp.paf"), str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    want = O.run([O.stage(O.PASS)], human_chimp)[0]  # what paf_read -> paf_write gives for every record
    lines = want.splitlines(keepends=True)
    assert out.read_bytes() == want + b"".join(lines[:20])
    # records the reference aborts on end this process the same way (status 1 for st_errAbort, SIGABRT for assert)
    for which in (1, 2, 3, 4):
        r = subprocess.run([str(exe), "--fail", str(which)], capture_output=True, text=True, timeout=120)
        assert r.returncode in (1, -6), (which, r.returncode, r.stderr[-500:])
        assert r.stderr.strip(), which
