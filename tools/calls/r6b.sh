#!/bin/bash
set -o pipefail
out=gpurun_out/r6b
mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout -k 10 400 python bench.py --workload cfg5 --cpu-sample 0 > $out/b.json 2> $out/b.err; echo "rc=$? [$1]"; grep -v amdgpu.ids $out/b.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6b/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print('   ', d['value'], d['ms_per_step'], d['roofline']['frac'], 'bitmap', k.get('k_cov_bitmap'), 'walk', k.get('k_cov_walk'), 'merge', k.get('k_cov_merge'), d.get('hbm'))
PY
}
run "4 GiB (default)"
PAFFY_COV_BITMAP_MB=8192 run "8 GiB"
PAFFY_COV_BITMAP_MB=16384 run "16 GiB"
PAFFY_COV_BITMAP_MB=32768 run "32 GiB"
