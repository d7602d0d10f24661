#!/bin/bash
set -o pipefail
out=gpurun_out/r6a
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6a/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['dominant_avg_kernel_ms'], r['dominant_frac'], d['kernel_ms'])
print(d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'], d['end_to_end']['value'], d['end_to_end']['GBps_out'])
PY
