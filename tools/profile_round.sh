#!/bin/bash
# Regenerates the profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01_c
# 1. rocprofv3 --kernel-trace --stats of the default bench command (trace domains only)
# 2. HBM traffic per kernel from separate --pmc passes (tools/traffic_collect.py)
# 3. SQ counters per kernel from separate --pmc passes (tools/pmc_collect.py)
set -e -o pipefail
tag=${1:-round}
export TMPDIR=/tmp
out=gpurun_out/profile_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 bench.py --steps 20 --cpu-sample 0 > $out/bench_under_rocprof.json 2> $out/rocprof.log
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_cfg3_b131072.csv
echo "kernel stats done"
python3 tools/traffic_collect.py $out/traffic.json > $out/traffic.log 2>&1
echo "traffic done"
python3 tools/pmc_collect.py $out/${tag}_pmc_record_kernels_cfg3_b131072.json > $out/pmc.log 2>&1
echo "pmc done"
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -1 $out/bench_default.json
