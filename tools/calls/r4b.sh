#!/bin/bash
set -o pipefail
out=gpurun_out/r4b
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
for mode in pipe mism chain pipe tile bed pipe; do
  timeout -k 10 300 python tools/fuzz_gpu.py 100 $((RANDOM)) $mode > $out/fuzz_${mode}_$RANDOM.txt 2>&1; echo "$mode rc=$? $(tail -1 $out/fuzz_${mode}_*.txt | tail -1 | cut -c1-200)"
done
