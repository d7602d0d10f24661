#!/bin/bash
set -o pipefail
out=gpurun_out/r3c
mkdir -p $out
for n in 4000000 7000000 10000000; do
  timeout -k 10 300 python tools/dbg_cfg5_big.py $n > $out/dbg_$n.txt 2>&1; echo "n=$n rc=$?"; tail -4 $out/dbg_$n.txt | cut -c1-400
done
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_21.so timeout -k 10 200 python bench.py --steps 4 --cpu-sample 0 > $out/abl21.txt 2>&1; echo "abl21 rc=$?"
