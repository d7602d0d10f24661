#!/bin/bash
set -o pipefail
out=gpurun_out/r3f
mkdir -p $out
export TMPDIR=/tmp
PAFFY_SHARD_TIMING=1 timeout -k 10 600 python bench.py --workload cfg5 --batch 10000000 --steps 2 --warmup 1 --force-dist --cpu-sample 0 > $out/bench_cfg5_10M_rccl.json 2> $out/bench_cfg5_10M_rccl.err; echo "cfg5 10M rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg5_10M_rccl.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['phase_ms_last_step'], d['hbm'])"
PAFFY_SHARD_TIMING=1 timeout -k 10 600 python bench.py --workload cfg5 --batch 10000000 --steps 2 --warmup 1 --force-dist --cpu-sample 0 --fresh-memory > $out/bench_cfg5_10M_rccl_fresh.json 2> $out/bench_cfg5_10M_rccl_fresh.err; echo "cfg5 10M fresh rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg5_10M_rccl_fresh.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['phase_ms_last_step'], d['hbm'])"
timeout -k 10 300 python bench.py --workload cfg5 --cpu-sample 0 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; echo "cfg5 rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg5.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['roofline']['frac'])"
bash tools/profile_workload.sh r03_a cfg3; echo "profile cfg3 rc=$?"
python3 tools/pmc_collect.py gpurun_out/profile_r03_a_cfg3/r03_a_pmc_cfg3.json > gpurun_out/profile_r03_a_cfg3/pmc.log 2>&1; echo "pmc rc=$?"
