#!/bin/bash
set -o pipefail
out=gpurun_out/r6r
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_rows_edges.py tests/test_gpu_parity.py tests/test_gpu_mismatches.py tests/test_gpu_properties.py tests/test_filter.py tests/test_cli.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) pipe 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py 40 $((RANDOM)) mism 2>&1 | tail -1
for w in cfg3 cfg4; do
timeout -k 10 300 python bench.py --workload $w --steps 30 --cpu-sample 2048 > $out/b.json 2> $out/b.err; echo -n "[$w] "
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6r/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'wave', k.get('k_size_wave'), 'lds', k.get('k_size_lds'), 'arena', k.get('k_arena_size'), (d.get('cpu_baseline') or {}).get('gpu_output_matches'))
PY
done
