#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r5s
timeout -k 10 300 python tools/probes/d2h_pieces.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5s/d2h_pieces_gaps.txt
