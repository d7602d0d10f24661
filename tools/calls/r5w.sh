#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
for m in dedupe pipe mism tile bed chain dedupe; do timeout -k 10 300 python tools/fuzz_gpu.py 110 $((RANDOM)) $m 2>&1 | grep -v amdgpu.ids | tail -1; done
