#!/bin/bash
# soak of the round's new code against the oracle
set -o pipefail
out=gpurun_out/r3u
mkdir -p $out
export TMPDIR=/tmp
for mode in pipe tile mism bed chain; do
  timeout -k 10 200 python tools/fuzz_gpu.py 75 $((RANDOM)) $mode > $out/fuzz_$mode.txt 2>&1; echo "$mode rc=$?"; tail -2 $out/fuzz_$mode.txt | cut -c1-300
done
