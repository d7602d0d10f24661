#!/bin/bash
set -o pipefail
out=gpurun_out/r5x
mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --steps 30 --cpu-sample 0 $2 > $out/b.json 2> $out/b.err; echo "rc=$? [$1 $2]"; grep -v amdgpu.ids $out/b.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5x/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print('   ', d['value'], d['ms_per_step'], d['roofline']['frac'], 'rows', k.get('k_emit_rows'), 'live', d['roofline']['dominant_avg_kernel_ms'])
PY
}
for rep in 1 2; do
unset PAFFY_HIP_LIB; run base ""
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_prio3.so run prio3 ""
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_prio3hold.so run prio3hold ""
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_prio1.so run prio1 ""
done
