#!/bin/bash
# co-residency experiment: emit and sizing kernels at half occupancy (extra LDS) on two streams
set -o pipefail
out=gpurun_out/r3d
mkdir -p $out
export TMPDIR=/tmp
run() { tag=$1; shift; timeout -k 10 200 python bench.py --steps 30 --cpu-sample 0 "$@" > $out/$tag.json 2> $out/$tag.err; echo "$tag rc=$? $(python -c "import json,sys; d=json.loads(open('$out/$tag.json').read().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'].get('k_emit_rows'), d['kernel_ms'].get('k_size_wave'), d['kernel_ms'].get('k_size_lds'))")"; }
run p1_plain
PAFFY_DBG_LDS_PAD=9216,9216,17408 run p1_half
PAFFY_DBG_LDS_PAD=9216,9216,17408 run p2_half --pipeline 2
PAFFY_DBG_LDS_PAD=9216,0,0 run p2_emit_half --pipeline 2
PAFFY_DBG_LDS_PAD=4096,4096,17408 run p2_emit12 --pipeline 2
PAFFY_DBG_LDS_PAD=9216,9216,17408 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python3 bench.py --pipeline 2 --steps 6 --cpu-sample 0 --no-kernel-events > $out/trace.json 2> $out/trace.err; echo "trace rc=$?"
cp $(ls $out/trace/*kernel_trace.csv $out/trace/*/*kernel_trace.csv 2>/dev/null | head -1) $out/trace_p2_half_kernel_trace.csv && rm -rf $out/trace
PAFFY_SHARD_TIMING=1 timeout -k 10 600 python bench.py --workload cfg5 --batch 10000000 --steps 1 --warmup 1 --force-dist --cpu-sample 0 > $out/bench_cfg5_10M_rccl.json 2> $out/bench_cfg5_10M_rccl.err; echo "cfg5 10M rc=$?"
tail -c 1800 $out/bench_cfg5_10M_rccl.json
