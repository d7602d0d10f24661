#!/bin/bash
# cfg4 (add_mismatches) profile artefacts of a round, GPU box, repo root:  tools/profile_cfg4.sh r01_g
set -e -o pipefail
tag=${1:-round}
export TMPDIR=/tmp
out=gpurun_out/profile_${tag}_cfg4
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 bench.py --workload cfg4 --steps 10 --cpu-sample 0 > $out/bench_under_rocprof.json 2> $out/rocprof.log
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_cfg4_b131072.csv
echo "kernel stats done"
python3 tools/traffic_collect.py $out/traffic_cfg4.json --workload cfg4 --steps 3 > $out/traffic.log 2>&1
echo "traffic done"
python3 bench.py --workload cfg4 --steps 20 > $out/bench_cfg4.json 2> $out/bench_cfg4.err
tail -1 $out/bench_cfg4.json
