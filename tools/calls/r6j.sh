#!/bin/bash
set -o pipefail
out=gpurun_out/r6j
mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --steps 10 --cpu-sample 2048 > $out/b.json 2> $out/b.err; echo "rc=$? [$1]"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6j/b.json').read().strip().splitlines()[-1])
e=d['end_to_end']; print('   ', e['GBps_out'], e['runs_GBps_out'], e['link'])
PY
}
for rep in 1 2 3; do
PAFFY_D2H_INFLIGHT=2 run "2 in flight"
PAFFY_D2H_INFLIGHT=1 run "1 in flight"
done
