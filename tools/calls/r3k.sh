#!/bin/bash
set -o pipefail
out=gpurun_out/r3k
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
timeout -k 10 300 python bench.py --workload cfg4 --cpu-sample 0 > $out/bench_cfg4.json 2> $out/bench_cfg4.err; echo "cfg4 rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg4.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['kernel_ms'].items() if v>0.2})"
for cmd in invert filter trim; do
 timeout -k 10 300 python tools/bench_extra.py --cmd $cmd --records 131072 > $out/extra_$cmd.txt 2>&1; tail -1 $out/extra_$cmd.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cmd'], d['records_per_s'], {k:v for k,v in d['kernel_ms'].items() if v>0.2})"
 PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_prev.so timeout -k 10 300 python tools/bench_extra.py --cmd $cmd --records 131072 > $out/extra_${cmd}_prev.txt 2>&1; tail -1 $out/extra_${cmd}_prev.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['cmd'], d['records_per_s'], {k:v for k,v in d['kernel_ms'].items() if v>0.2})"
done
