#!/bin/bash
f=tests/golden/fuzz/tile_r3_fail.paf
mkdir -p gpurun_out/r3w
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_r2.so python tools/dbg_tile_file.py $f 2>&1 | grep -v amdgpu.ids | head -4 | cut -c1-300
