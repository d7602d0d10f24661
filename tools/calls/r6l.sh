#!/bin/bash
set -o pipefail
out=gpurun_out/r6l
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > $out/r03_g_bench_cfg3.json 2> $out/b.err; echo "bench rc=$?"; grep -v amdgpu.ids $out/b.err | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6l/r03_g_bench_cfg3.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['dominant_avg_kernel_ms'], r['dominant_frac'])
e=d['end_to_end']; print(e['value'], e['GBps_out'], e['runs_GBps_out'])
print(d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'])
PY
