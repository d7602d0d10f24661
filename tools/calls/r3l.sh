#!/bin/bash
set -o pipefail
out=gpurun_out/r3l
mkdir -p $out
export TMPDIR=/tmp
for lib in new notext; do
 if [ $lib = notext ]; then export PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_notext.so; else unset PAFFY_HIP_LIB; fi
 timeout -k 10 300 python bench.py --workload cfg4 --cpu-sample 0 > $out/bench_cfg4_$lib.json 2> $out/bench_cfg4_$lib.err; echo "cfg4 $lib rc=$?"
 python -c "import json; d=json.loads(open('$out/bench_cfg4_$lib.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['kernel_ms'].items() if v>0.2})"
 for cmd in invert add; do
  timeout -k 10 300 python tools/bench_extra.py --cmd $cmd --records 131072 > $out/extra_${cmd}_$lib.txt 2>&1; tail -1 $out/extra_${cmd}_$lib.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['cmd'], d['records_per_s'], {k:v for k,v in d['kernel_ms'].items() if v>0.2})"
 done
done
unset PAFFY_HIP_LIB
timeout -k 10 300 python bench.py --workload cfg5 --cpu-sample 0 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; echo "cfg5 rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg5.json').read().splitlines()[-1]); print(d['value'], d['ms_each_step'], d['roofline']['frac'], d['roofline']['step_kernels_ms'], {k:v for k,v in d['kernel_ms'].items() if v>0.5})"
