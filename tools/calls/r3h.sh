#!/bin/bash
set -o pipefail
out=gpurun_out/r3h
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_chain.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
for lib in prev new; do
  if [ $lib = prev ]; then export PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_prev.so; else unset PAFFY_HIP_LIB; fi
  timeout -k 10 300 python tools/bench_extra.py --cmd chain --records 1000000 --mean-ops 64 --contigs 1 > $out/chain_big_$lib.txt 2>&1; echo "$lib rc=$?"; tail -1 $out/chain_big_$lib.txt | cut -c1-900
  timeout -k 10 300 python tools/bench_extra.py --cmd chain --records 300000 > $out/chain_300k_$lib.txt 2>&1; echo "$lib rc=$?"; tail -1 $out/chain_300k_$lib.txt | cut -c1-300
done
