#!/bin/bash
f=tests/golden/fuzz/tile_r3_fail.paf
for l in r2 22c0d76 e325ed8 prev; do
  python tools/dbg_tile_lib.py paffy_amd/abl/libpaffy_hip_$l.so $f 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
done
python tools/dbg_tile_lib.py paffy_amd/libpaffy_hip.so $f 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
