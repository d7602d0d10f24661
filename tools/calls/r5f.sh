#!/bin/bash
set -o pipefail
out=gpurun_out/r5f
mkdir -p $out
export TMPDIR=/tmp
for v in "" "--pipeline 1" "--batch 262144 --steps 20" "--batch 262144 --steps 20 --pipeline 1" "--kernel-events all --pipeline 1" "--pipeline 3"; do
  timeout -k 10 300 python bench.py --steps 40 --cpu-sample 0 $v > $out/b.json 2> $out/b.err; echo "rc=$? [$v]"; tail -2 $out/b.err
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5f/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print('   ', d['value'], d['ms_per_step'], d['config']['records_per_step_per_gpu'], r['frac'], r['kernel_event_sum_ms'], r['dominant_avg_kernel_ms'], r['dominant_frac'], r['dominant_launches_timed'])
PY
done
cp $out/b.json $out/last.json
