#!/bin/bash
set -o pipefail
out=gpurun_out/r6u
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) pipe 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py 40 $((RANDOM)) mism 2>&1 | tail -1
timeout -k 10 300 python bench.py > $out/r03_h_bench_cfg3.json 2> $out/b3.err; echo "cfg3 rc=$?"
timeout -k 10 400 python bench.py --workload cfg4 > $out/r03_h_bench_cfg4.json 2> $out/b4.err; echo "cfg4 rc=$?"
python - <<'PY'
import json
for w in ('cfg3','cfg4'):
    d=json.loads(open(f'gpurun_out/r6u/r03_h_bench_{w}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print(w, d['value'], d['ms_per_step'], r['frac'], r['dominant_avg_kernel_ms'], r['dominant_frac'])
    print('   cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('gpu_output_matches'), (d.get('end_to_end') or {}).get('runs_GBps_out'), d['kernel_ms'])
PY
