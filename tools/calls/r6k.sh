#!/bin/bash
set -o pipefail
out=gpurun_out/r6k
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_cli.py tests/test_gpu_parity.py tests/test_paf_tools_script.py tests/test_gpu_launcher.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do REPS=8 timeout -k 10 300 python tools/probes/d2h_pieces.py 2>&1 | grep -E "^run" | cut -c1-60; done
timeout -k 10 300 python bench.py --steps 10 --cpu-sample 2048 > $out/b.json 2> $out/b.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6k/b.json').read().strip().splitlines()[-1])
e=d['end_to_end']; print('   ', e['value'], e['GBps_out'], e['runs_GBps_out'], e['seconds'])
PY
