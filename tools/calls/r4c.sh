#!/bin/bash
set -o pipefail
out=gpurun_out/r4c
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-sample 0 "$@" > $out/$tag.json 2> $out/$tag.err; echo "$tag rc=$? $(python -c "import json,sys; d=json.loads(open('$out/$tag.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['kernel_ms'].items() if v>0.2})")"; }
run c3 --steps 40
run c4 --workload cfg4
run c2 --workload cfg2 --steps 40
timeout -k 10 200 python tools/fuzz_gpu.py 80 $((RANDOM)) pipe 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) mism 2>&1 | tail -1
python3 tools/traffic_collect.py $out/r03_e_traffic_cfg3.json --workload cfg3 --steps 3 > $out/traffic.log 2>&1; echo "traffic rc=$?"; tail -4 $out/traffic.log | cut -c1-300
