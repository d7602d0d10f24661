#!/bin/bash
set -o pipefail
out=gpurun_out/r6i
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; grep -v amdgpu.ids $out/bench_default.err | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6i/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['dominant_avg_kernel_ms'])
print(d['end_to_end'])
PY
