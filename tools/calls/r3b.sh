#!/bin/bash
# round 3, second call: clocks and phases of the row writer, do two streams overlap, cfg5 at a GPU's real share over RCCL
set -o pipefail
out=gpurun_out/r3b
mkdir -p $out
export TMPDIR=/tmp
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_24.so timeout -k 10 200 python bench.py --steps 4 --cpu-sample 0 > $out/abl24.txt 2>&1; echo "abl24 rc=$?"
PAFFY_HIP_LIB=$PWD/paffy_amd/abl/libpaffy_hip_20.so timeout -k 10 200 python bench.py --steps 4 --cpu-sample 0 > $out/abl20.txt 2>&1; echo "abl20 rc=$?"
rocprofv3 --kernel-trace -d $out/trace_p2 --output-format csv -- python3 bench.py --pipeline 2 --steps 6 --cpu-sample 0 --no-kernel-events > $out/trace_p2.json 2> $out/trace_p2.err; echo "trace rc=$?"
cp $(ls $out/trace_p2/*/*kernel_trace.csv | head -1) $out/trace_p2_kernel_trace.csv && rm -rf $out/trace_p2
PAFFY_SHARD_TIMING=1 timeout -k 10 600 python bench.py --workload cfg5 --batch 10000000 --steps 1 --warmup 1 --force-dist --cpu-sample 0 > $out/bench_cfg5_10M_rccl.json 2> $out/bench_cfg5_10M_rccl.err; echo "cfg5 10M rc=$?"
tail -c 1500 $out/bench_cfg5_10M_rccl.json
timeout -k 10 300 python bench.py --force-dist --cpu-sample 0 > $out/bench_cfg3_rccl_ws1.json 2> $out/bench_cfg3_rccl_ws1.err; echo "cfg3 rccl rc=$?"
tail -c 300 $out/bench_cfg3_rccl_ws1.json
