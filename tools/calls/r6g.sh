#!/bin/bash
set -o pipefail
out=gpurun_out/r6g
mkdir -p $out
export TMPDIR=/tmp
for tb in 200000 350000; do
  timeout -k 10 400 python bench.py --workload cfg5 --cpu-sample 0 --tile-text-batch $tb > $out/b.json 2> $out/b.err; echo "rc=$? [$tb]"; grep -v amdgpu.ids $out/b.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6g/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print('   ', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), 'sep', k.get('k_sep_index'), 'hdr', k.get('k_header'))
PY
done
