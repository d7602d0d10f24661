#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r5u
timeout -k 10 300 python bench.py --gpus 2 --one-device --dist-backend gloo --steps 6 --cpu-sample 0 > gpurun_out/r5u/b2.json 2> gpurun_out/r5u/b2.err; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r5u/b2.err | tail -5; cut -c1-400 gpurun_out/r5u/b2.json
timeout -k 10 300 python bench.py --force-dist --steps 10 --cpu-sample 0 > gpurun_out/r5u/b1.json 2> gpurun_out/r5u/b1.err; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r5u/b1.err | tail -5; cut -c1-300 gpurun_out/r5u/b1.json
