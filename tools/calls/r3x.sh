#!/bin/bash
mkdir -p gpurun_out/r3x
timeout -k 10 500 python tools/dbg_tile_reduce.py tests/golden/fuzz/tile_r3_fail.paf gpurun_out/r3x/reduced.paf 400 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-400
