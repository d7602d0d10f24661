#!/bin/bash
set -o pipefail
out=gpurun_out/r6d
mkdir -p $out
export TMPDIR=/tmp
for mb in 32768 65536; do
PAFFY_COV_BITMAP_MB=$mb PAFFY_SHARD_TIMING=1 timeout -k 10 600 python bench.py --workload cfg5 --batch 10000000 --steps 2 --warmup 1 --force-dist --cpu-sample 0 > $out/b.json 2> $out/b.err; echo "cfg5 10M rc=$? [$mb]"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6d/b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('hbm',{}).get('peak_in_use_GB'))
PY
done
