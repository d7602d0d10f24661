#!/bin/bash
set -o pipefail
out=gpurun_out/r5v
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_edges.py tests/test_gpu_mismatches.py tests/test_gpu_properties.py tests/test_filter.py tests/test_view_stats.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/fuzz_gpu.py 40 $((RANDOM)) pipe 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py 30 $((RANDOM)) mism 2>&1 | tail -1
run() {
  timeout -k 10 300 python bench.py --steps 30 --cpu-sample 0 $2 > $out/b.json 2> $out/b.err; echo "rc=$? [$1 $2]"; grep -v amdgpu.ids $out/b.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5v/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print('   ', d['value'], d['ms_per_step'], d['roofline']['frac'], 'wave', k.get('k_size_wave'), 'mid', k.get('k_size_mid'), 'lds', k.get('k_size_lds'))
PY
}
for rep in 1 2; do
run g128 ""
PAFFY_MID_BYTES=0 run off ""
done
run g128 "--workload cfg4"
PAFFY_MID_BYTES=0 run off "--workload cfg4"
PAFFY_MID_BYTES=9000 run mid9000 ""
