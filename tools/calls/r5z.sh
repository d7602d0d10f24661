#!/bin/bash
set -o pipefail
out=gpurun_out/r5z
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_dedupe.py tests/test_split_file.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/fuzz_gpu.py 40 $((RANDOM)) dedupe 2>&1 | tail -1
timeout -k 10 200 python tools/bench_extra.py --cmd dedupe 2>/dev/null | tail -1 | cut -c1-330
