#!/bin/bash
# round 3, first call: probes, GPU tests, the three bench workloads, pipeline experiment
set -o pipefail
out=gpurun_out/r3a
mkdir -p $out
timeout -k 10 120 tools/probes/lds_rowstore > $out/lds_rowstore.txt 2>&1; echo "probe rc=$?"
timeout -k 10 120 tools/probes/lds_unaligned > $out/lds_unaligned.txt 2>&1; echo "probe2 rc=$?"
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
python bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err && tail -c 600 $out/bench_cfg3.json
python bench.py --pipeline 2 --cpu-sample 0 > $out/bench_cfg3_p2.json 2> $out/bench_cfg3_p2.err && tail -c 300 $out/bench_cfg3_p2.json
python bench.py --pipeline 3 --cpu-sample 0 > $out/bench_cfg3_p3.json 2> $out/bench_cfg3_p3.err && tail -c 300 $out/bench_cfg3_p3.json
python bench.py --workload cfg4 > $out/bench_cfg4.json 2> $out/bench_cfg4.err && tail -c 300 $out/bench_cfg4.json
python bench.py --workload cfg5 > $out/bench_cfg5.json 2> $out/bench_cfg5.err && tail -c 300 $out/bench_cfg5.json
