#!/bin/bash
set -o pipefail
out=gpurun_out/r6v
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --workload cfg2 > $out/r03_h_bench_cfg2.json 2> $out/b2.err; echo "cfg2 rc=$?"
timeout -k 10 400 python bench.py --workload cfg5 > $out/r03_h_bench_cfg5.json 2> $out/b5.err; echo "cfg5 rc=$?"
python - <<'PY'
import json
for w in ('cfg2','cfg5'):
    d=json.loads(open(f'gpurun_out/r6v/r03_h_bench_{w}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print(w, d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))
PY
