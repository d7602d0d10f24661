#!/bin/bash
set -o pipefail
out=gpurun_out/r5c
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_edges.py tests/test_gpu_properties.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 40 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c/bench_cfg3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('roofline_by_kernel',{}).items(): print(k, v)
PY
timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) pipe 2>&1 | tail -1
