#!/bin/bash
set -o pipefail
out=gpurun_out/r5d
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_mismatches.py tests/test_gpu_parity.py tests/test_paf_api.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/fuzz_gpu.py 45 $((RANDOM)) mism 2>&1 | tail -1
timeout -k 10 300 python bench.py --workload cfg4 --steps 20 --cpu-sample 0 > $out/bench_cfg4.json 2> $out/bench_cfg4.err; echo "bench4 rc=$?"
timeout -k 10 300 python bench.py --steps 40 --cpu-sample 0 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "bench3 rc=$?"
python - <<'PY'
import json
for w in ('cfg4','cfg3'):
    d=json.loads(open(f'gpurun_out/r5d/bench_{w}.json').read().strip().splitlines()[-1])
    print(w, d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms'])
PY
