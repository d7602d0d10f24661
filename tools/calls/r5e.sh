#!/bin/bash
set -o pipefail
out=gpurun_out/r5e
mkdir -p $out
export TMPDIR=/tmp
for v in "" "--no-kernel-events" "--pipeline 2 --no-kernel-events" "--batch 262144 --steps 20" "--batch 262144 --steps 20 --no-kernel-events" "--batch 262144 --steps 20 --pipeline 2"; do
  timeout -k 10 300 python bench.py --steps 40 --cpu-sample 0 $v > $out/b.json 2> $out/b.err; echo "rc=$? [$v]"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5e/b.json').read().strip().splitlines()[-1])
print('   ', d['value'], d['ms_per_step'], d['config']['records_per_step_per_gpu'])
PY
done
