#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r5s
for i in 1 2 3; do REPS=6 timeout -k 10 300 python tools/probes/d2h_pieces.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee -a gpurun_out/r5s/d2h_pieces_stalls.txt; done
