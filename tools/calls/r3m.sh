#!/bin/bash
set -o pipefail
out=gpurun_out/r3m
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_edges.py tests/test_gpu_properties.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
for lib in new prev2; do
 if [ $lib = prev2 ]; then export PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_notext.so; else unset PAFFY_HIP_LIB; fi
 timeout -k 10 300 python bench.py --steps 40 --cpu-sample 0 > $out/bench_cfg3_$lib.json 2> $out/bench_cfg3_$lib.err; echo "cfg3 $lib rc=$?"
 python -c "import json; d=json.loads(open('$out/bench_cfg3_$lib.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['kernel_ms'].items() if v>0.05})"
 timeout -k 10 300 python bench.py --workload cfg2 --steps 40 --cpu-sample 0 > $out/bench_cfg2_$lib.json 2> $out/bench_cfg2_$lib.err; echo "cfg2 $lib rc=$?"
 python -c "import json; d=json.loads(open('$out/bench_cfg2_$lib.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['kernel_ms'].items() if v>0.05})"
done
