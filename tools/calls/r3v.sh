#!/bin/bash
set -o pipefail
out=gpurun_out/r3v
mkdir -p $out
f=tests/golden/fuzz/tile_r3_fail.paf
python tools/dbg_tile_file.py $f > $out/new.txt 2>&1; cat $out/new.txt | grep -v amdgpu.ids | cut -c1-400
PAFFY_TWO_PASS_INDEX=1 python tools/dbg_tile_file.py $f > $out/twopass.txt 2>&1; grep -v amdgpu.ids $out/twopass.txt | head -3 | cut -c1-300
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_notext.so python tools/dbg_tile_file.py $f > $out/notext.txt 2>&1; grep -v amdgpu.ids $out/notext.txt | head -3 | cut -c1-300
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_prev.so python tools/dbg_tile_file.py $f > $out/prev.txt 2>&1; grep -v amdgpu.ids $out/prev.txt | head -3 | cut -c1-300
