#!/bin/bash
set -o pipefail
out=gpurun_out/r4d
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
timeout -k 10 200 python tools/fuzz_gpu.py 90 $((RANDOM)) pipe 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) mism 2>&1 | tail -1
