#!/bin/bash
set -o pipefail
out=gpurun_out/r6o
mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --steps 30 --cpu-sample 0 > $out/b.json 2> $out/b.err; echo "rc=$? [$1]"; grep -v amdgpu.ids $out/b.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6o/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print('   ', d['value'], d['ms_per_step'], 'wave', k.get('k_size_wave'), 'lds', k.get('k_size_lds'))
PY
}
for rep in 1 2; do
unset PAFFY_HIP_LIB PAFFY_WAVE_OPS PAFFY_WAVE_BYTES; run base
export PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_occ5.so
PAFFY_WAVE_OPS=1536 PAFFY_WAVE_BYTES=4600 run "occ5 1536/4600"
PAFFY_WAVE_OPS=1472 PAFFY_WAVE_BYTES=4400 run "occ5 1472/4400"
done
