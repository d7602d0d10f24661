#!/bin/bash
set -o pipefail
out=gpurun_out/r5k
mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --workload cfg4 --steps 12 --cpu-sample 0 > $out/b.json 2> $out/b.err; echo "rc=$? [$1]"; grep -v amdgpu.ids $out/b.err | tail -3
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5k/b.json').read().strip().splitlines()[-1])
print('   ', d['value'], d['ms_per_step'], d['kernel_ms'].get('k_size_wave'), d['kernel_ms'].get('k_size_lds'))
PY
}
run default
PAFFY_WAVE_OPS=1536 PAFFY_WAVE_BYTES=4608 run "ops 1536 bytes 4608"
PAFFY_WAVE_OPS=1536 PAFFY_WAVE_BYTES=4300 run "ops 1536 bytes 4300"
PAFFY_WAVE_OPS=1792 PAFFY_WAVE_BYTES=5200 run "ops 1792 bytes 5200"
