#!/bin/bash
set -o pipefail
out=gpurun_out/r3g
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_chain.py tests/test_gpu_shard.py tests/test_gpu_tile.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
PAFFY_SHARD_TIMING=1 timeout -k 10 300 python bench.py --workload cfg5 --cpu-sample 0 > $out/bench_cfg5_timing.json 2> $out/bench_cfg5_timing.err; echo "cfg5 rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg5_timing.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['roofline']['frac'], d['roofline']['step_kernels_ms'], d['phase_ms_last_step'])"
timeout -k 10 300 python bench.py --workload cfg5 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; echo "cfg5 rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg5.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['roofline']['frac'], d['roofline']['step_kernels_ms'])"
timeout -k 10 300 python bench.py --steps 20 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "cfg3 rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg3.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['end_to_end'])"
