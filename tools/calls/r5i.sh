#!/bin/bash
set -o pipefail
out=gpurun_out/r5i
mkdir -p $out
export TMPDIR=/tmp
for v in "" "--interleave 1" "" "--interleave 1" "--pipeline 1"; do
  timeout -k 10 300 python bench.py --steps 40 --cpu-sample 0 $v > $out/b.json 2> $out/b.err; echo "rc=$? [$v]"; grep -v amdgpu.ids $out/b.err | tail -3
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5i/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print('   ', d['value'], d['ms_per_step'], r['frac'], r['kernel_event_sum_ms'], r['dominant_avg_kernel_ms'])
PY
done
