#!/bin/bash
set -o pipefail
out=gpurun_out/r5j
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_properties.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -15 $out/tests.log
