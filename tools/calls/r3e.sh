#!/bin/bash
set -o pipefail
out=gpurun_out/r3e
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
run() { tag=$1; shift; timeout -k 10 200 python bench.py --steps 30 --cpu-sample 0 "$@" > $out/$tag.json 2> $out/$tag.err; echo "$tag rc=$? $(python -c "import json,sys; d=json.loads(open('$out/$tag.json').read().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'].get('k_header'), d['kernel_ms'].get('k_emit_rows'), d['kernel_ms'].get('k_size_wave'), d['kernel_ms'].get('k_size_lds'))")"; }
run p1_plain
PAFFY_DBG_LDS_PAD=9216,9216,17408 run p1_half
PAFFY_DBG_LDS_PAD=9216,9216,17408 run p2_half --pipeline 2
PAFFY_DBG_LDS_PAD=4096,4096,17408 run p2_emit12 --pipeline 2
PAFFY_DBG_LDS_PAD=9216,9216,17408 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python3 bench.py --pipeline 2 --steps 6 --cpu-sample 0 --no-kernel-events > $out/trace.json 2> $out/trace.err; echo "trace rc=$?"
cp $(ls $out/trace/*kernel_trace.csv $out/trace/*/*kernel_trace.csv 2>/dev/null | head -1) $out/trace_p2_half_kernel_trace.csv && rm -rf $out/trace
PAFFY_SHARD_TIMING=1 timeout -k 10 600 python bench.py --workload cfg5 --batch 10000000 --steps 1 --warmup 1 --force-dist --cpu-sample 0 > $out/bench_cfg5_10M_rccl.json 2> $out/bench_cfg5_10M_rccl.err; echo "cfg5 10M rc=$?"
python -c "import json; d=json.loads(open('$out/bench_cfg5_10M_rccl.json').read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phase_ms_last_step'], d['hbm'])"
timeout -k 10 300 python tools/bench_extra.py > $out/bench_extra.txt 2>&1; echo "extra rc=$?"; tail -15 $out/bench_extra.txt
