#!/bin/bash
set -o pipefail
out=gpurun_out/r6c
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_tile.py tests/test_gpu_to_bed.py tests/test_gpu_shard.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/fuzz_gpu.py 40 $((RANDOM)) tile 2>&1 | tail -1
timeout -k 10 400 python bench.py --workload cfg5 > $out/r03_g_bench_cfg5.json 2> $out/b5.err; echo "cfg5 rc=$?"; cut -c1-260 $out/r03_g_bench_cfg5.json
PAFFY_SHARD_TIMING=1 timeout -k 10 600 python bench.py --workload cfg5 --batch 10000000 --steps 2 --warmup 1 --force-dist --cpu-sample 0 > $out/r03_g_bench_cfg5_10M_records_rccl_world1.json 2> $out/b5big.err; echo "cfg5 10M rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6c/r03_g_bench_cfg5_10M_records_rccl_world1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('hbm'))
PY
