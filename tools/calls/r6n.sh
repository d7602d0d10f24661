#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
for m in bed tile dedupe pipe bed tile; do timeout -k 10 300 python tools/fuzz_gpu.py 140 $((RANDOM)) $m 2>&1 | grep -v amdgpu.ids | tail -1; done
