"""Host sanitizers: the two host programs that run without a GPU under AddressSanitizer + UndefinedBehaviorSanitizer.
  * `make -C host asan` -> bin/paffy_asan, the N-GPU launcher: every case of tests/test_launcher.py runs again through it (stand-in worker);
  * `make -C oracle asan` -> oracle/_san/oracle_asan, the CPU oracle behind a small driver: the reference's fixture, a synthetic stream and the
    fuzz fixtures through the stream pipes, tile, dedupe, chain and to_bed, the bytes compared with the plain build's.
A sanitizer report ends the process with a non-zero status (-fno-sanitize-recover, ASan's default), which fails the comparison."""
import os
import subprocess
import sys

import pytest

import oracle_lib as O
import synth_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_ENV = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def test_launcher_cases_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s", "asan"])
    env = dict(SAN_ENV, PAFFY_LAUNCHER=os.path.join(ROOT, "bin", "paffy_asan"))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_launcher.py"), "-x", "-q", "-p", "no:cacheprovider"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "passed" in p.stdout


@pytest.fixture(scope="module")
def oracle_asan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    return os.path.join(ROOT, "oracle", "_san", "oracle_asan")


def inputs(human_chimp):
    yield "fixture", human_chimp
    yield "synthetic", synth_lib.generate(0x5EED0003, 400, 0, 300, threads=1)
    fuzz = os.path.join(ROOT, "tests", "golden", "fuzz")
    for name in sorted(os.listdir(fuzz)):
        with open(os.path.join(fuzz, name), "rb") as fh:
            yield name, fh.read()
    # records the reference aborts on: the output is what came before them
    good = synth_lib.generate(0x5EED0003, 60, 0, 30, threads=1).splitlines(keepends=True)
    yield "bad strand", b"".join(good[:10]) + b"q\t10\t0\t5\t*\tt\t10\t0\t5\t5\t5\t60\n" + b"".join(good[10:])
    yield "bad cigar", b"".join(good[:5]) + b"q\t10\t0\t5\t+\tt\t10\t0\t5\t5\t5\t60\tcg:Z:3M2Q\n" + b"".join(good[5:])
    yield "empty", b""
    yield "no newline", b"".join(good[:3])[:-1]


def test_oracle_under_asan_and_ubsan_gives_the_same_bytes(tmp_path, oracle_asan, human_chimp):
    S = O.stage
    pipes = {"4": [S(O.SHATTER)], "1,2,4": [S(O.INVERT), S(O.TRIM_IDENTITY), S(O.SHATTER)], "3": [S(O.TRIM_FIXED)], "6,1": [S(O.REMOVE_MISMATCHES), S(O.INVERT)],
             "8,7": [S(O.FILTER), S(O.PASS)]}
    for name, data in inputs(human_chimp):
        src = tmp_path / "in.paf"
        src.write_bytes(data)
        out = tmp_path / "out"
        for spec, stages in pipes.items():
            want, err = O.run(stages, data)
            p = subprocess.run([oracle_asan, "run", spec, str(src), str(out)], env=SAN_ENV, capture_output=True, timeout=600)
            assert p.returncode == (100 + err.code if err.code else 0), (name, spec, p.returncode, p.stderr[-3000:])
            assert out.read_bytes() == want, (name, spec)
        for cmd, fn in (("tile", O.tile), ("dedupe-a", lambda d: O.dedupe(d, True)), ("to_bed", lambda d: O.to_bed(d, include_inverted=True)),
                        ("chain", lambda d: O.chain(d)[:2])):
            if name == "case1.paf" and cmd in ("tile", "to_bed"):
                continue  # sequences of gigabases: minutes of counter sweeps under the sanitizer
            want, err = fn(data)
            p = subprocess.run([oracle_asan, cmd, str(src), str(out)], env=SAN_ENV, capture_output=True, timeout=600)
            assert p.returncode == (100 + err.code if err.code else 0), (name, cmd, p.returncode, p.stderr[-3000:])
            if not err.code:
                assert out.read_bytes() == want, (name, cmd)
