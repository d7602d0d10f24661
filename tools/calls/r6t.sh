#!/bin/bash
set -o pipefail
out=gpurun_out/r6t
mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --workload $2 --steps 20 --cpu-sample 1024 > $out/b.json 2> $out/b.err; echo -n "[$1 $2] "
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6t/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print(d['value'], d['ms_per_step'], 'line', k.get('k_emit_line'), 'lds<line>', k.get('k_emit_lds<line>'), 'rows', k.get('k_emit_rows'), (d.get('cpu_baseline') or {}).get('gpu_output_matches'))
PY
}
for rep in 1 2; do
unset PAFFY_HIP_LIB; run base cfg4
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_rows32k.so run rows32k cfg4
done
unset PAFFY_HIP_LIB; run base cfg3
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_rows32k.so run rows32k cfg3
