#!/bin/bash
set -o pipefail
out=gpurun_out/r3t
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_tile.py tests/test_gpu_to_bed.py tests/test_gpu_parity.py tests/test_gpu_shard.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
timeout -k 10 300 python bench.py --workload cfg5 --cpu-sample 0 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; echo "cfg5 rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg5.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['roofline']['frac'], d['roofline']['step_kernels_ms'], {k:v for k,v in d['kernel_ms'].items() if v>0.5})"
timeout -k 10 300 python tools/bench_extra.py --cmd bed --records 200000 > $out/bed.txt 2>&1; tail -1 $out/bed.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['seconds'], {k:v for k,v in d['kernel_ms'].items() if v>1})"
timeout -k 10 300 python bench.py --steps 10 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; python -c "import json; d=json.loads(open('$out/bench_cfg3.json').read().splitlines()[-1]); print(d['value'], d['end_to_end'])"
