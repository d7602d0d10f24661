#!/bin/bash
set -o pipefail
out=gpurun_out/r6p
mkdir -p $out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --steps 40 --cpu-sample 0 > $out/b.json 2> $out/b.err; echo -n "[$1] "
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6p/b.json').read().strip().splitlines()[-1])
k=d['kernel_ms']
print(d['value'], d['ms_per_step'], 'wave', k.get('k_size_wave'), 'lds', k.get('k_size_lds'), 'arena', k.get('k_arena_size'))
PY
}
for rep in 1 2 3; do
run base
PAFFY_LVL0_BYTES=18000 run 18000
PAFFY_LVL0_BYTES=20000 run 20000
PAFFY_LVL0_BYTES=21000 run 21000
done
