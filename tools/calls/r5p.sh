#!/bin/bash
set -o pipefail
out=gpurun_out/r5p
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_dedupe.py tests/test_split_file.py tests/test_paf_tools_script.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -25 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_extra.py --cmd dedupe 2>&1 | tail -3
