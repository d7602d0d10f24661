#!/bin/bash
set -o pipefail
out=gpurun_out/r6e
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_tile.py tests/test_gpu_to_bed.py tests/test_gpu_shard.py tests/test_paf_api.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) tile 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py 40 $((RANDOM)) bed 2>&1 | tail -1
run() {
  timeout -k 10 400 python bench.py --workload cfg5 --cpu-sample 0 > $out/b.json 2> $out/b.err; echo "rc=$? [$1]"; grep -v amdgpu.ids $out/b.err | tail -2
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6e/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print('   ', d['value'], d['ms_per_step'], d['roofline']['frac'], 'bitmap', k.get('k_cov_bitmap_wave'), k.get('k_cov_bitmap'), 'walk', k.get('k_cov_walk'))
PY
}
run "two shapes"
PAFFY_COV_WAVE_BYTES=0 run "four waves only"
run "two shapes"
PAFFY_COV_WAVE_BYTES=9000 run "wave up to 9000"
