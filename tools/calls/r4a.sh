#!/bin/bash
set -o pipefail
out=gpurun_out/r4a
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_tile.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_prev.so timeout -k 10 300 python -m pytest tests/test_gpu_tile.py -m gpu -x -q -k overflow > $out/tests_prev.log 2>&1; echo "prev lib on the new test rc=$? (expected to fail)"; tail -2 $out/tests_prev.log
for mode in tile tile tile; do
  timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) $mode > $out/fuzz_${mode}_$RANDOM.txt 2>&1; echo "$mode rc=$?"
done
tail -q -n 1 $out/fuzz_tile_*.txt
timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) bed 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) pipe 2>&1 | tail -1
