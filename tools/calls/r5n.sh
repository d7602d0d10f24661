#!/bin/bash
set -o pipefail
out=gpurun_out/r5n
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_edges.py tests/test_gpu_properties.py tests/test_cli.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
run() {
  timeout -k 10 300 python bench.py --steps 40 $2 > $out/b.json 2> $out/b.err; echo "rc=$? [$1 $2]"; grep -v amdgpu.ids $out/b.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5n/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print('   ', d['value'], d['ms_per_step'], r['frac'], 'sep', d['kernel_ms'].get('k_sep_index'), 'hdr', d['kernel_ms'].get('k_header'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
}
run w4 "--cpu-sample 16384"
run w4 "--cpu-sample 0"
run w4 "--workload cfg2 --cpu-sample 0"
