#!/bin/bash
# final set of round 3: GPU tests, cfg5 traffic at the bench's batch size, cfg5 lines (2 M and 10 M records per step), cfg2 profile, cfg3 line
set -o pipefail
out=gpurun_out/r3s
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
timeout -k 10 900 python3 tools/traffic_collect.py $out/r03_d_traffic_cfg5.json --workload cfg5 --steps 2 > $out/traffic_cfg5.log 2>&1; echo "traffic cfg5 rc=$?"
cp $out/r03_d_traffic_cfg5.json profiles/ 2>/dev/null
timeout -k 10 300 python bench.py --workload cfg5 > $out/r03_d_bench_cfg5.json 2> $out/bench_cfg5.err; echo "cfg5 rc=$?"
python -c "import json; d=json.loads(open('$out/r03_d_bench_cfg5.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['algorithmic_bytes_per_launch'])"
PAFFY_SHARD_TIMING=1 timeout -k 10 600 python bench.py --workload cfg5 --batch 10000000 --steps 2 --warmup 1 --force-dist --cpu-sample 0 > $out/r03_d_bench_cfg5_10M_records_rccl_world1.json 2> $out/bench_cfg5_10M.err; echo "cfg5 10M rc=$?"
python -c "import json; d=json.loads(open('$out/r03_d_bench_cfg5_10M_records_rccl_world1.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['phase_ms_last_step'], d['hbm'])"
bash tools/profile_workload.sh r03_d cfg2; echo "profile cfg2 rc=$?"
timeout -k 10 400 python bench.py > $out/r03_d_bench_cfg3.json 2> $out/bench_cfg3.err; echo "cfg3 rc=$?"
python -c "import json; d=json.loads(open('$out/r03_d_bench_cfg3.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'], d['end_to_end'])"
