#!/bin/bash
# Profile artefacts of one bench workload for a round, GPU box, repo root:  tools/profile_workload.sh r02_a cfg4 [bench args...]
#   1. rocprofv3 --kernel-trace --stats of the bench command (trace domains only)
#   2. HBM traffic per kernel from separate --pmc passes (tools/traffic_collect.py)
#   3. the bench line itself
set -e -o pipefail
tag=${1:-round}
wl=${2:-cfg3}
shift 2 || true
export TMPDIR=/tmp
out=gpurun_out/profile_${tag}_${wl}
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 bench.py --workload $wl --steps 10 --cpu-sample 0 "$@" > $out/bench_under_rocprof.json 2> $out/rocprof.log
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_${wl}.csv
echo "kernel stats done"
python3 tools/traffic_collect.py $out/${tag}_traffic_${wl}.json --workload $wl --steps 3 > $out/traffic.log 2>&1
echo "traffic done"
python3 bench.py --workload $wl "$@" > $out/${tag}_bench_${wl}.json 2> $out/bench.err
tail -1 $out/${tag}_bench_${wl}.json
