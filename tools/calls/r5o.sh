#!/bin/bash
set -o pipefail
out=gpurun_out/r5o
mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --steps 30 --cpu-sample 0 $2 > $out/b.json 2> $out/b.err; echo "rc=$? [$1 $2]"; grep -v amdgpu.ids $out/b.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5o/b.json').read().strip().splitlines()[-1])
print('   ', d['value'], d['ms_per_step'], 'sep', d['kernel_ms'].get('k_sep_index'))
PY
}
for rep in 1 2; do
unset PAFFY_HIP_LIB; run look4 ""
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_look1.so run look1 ""
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_look2.so run look2 ""
done
