#!/bin/bash
set -o pipefail
out=gpurun_out/r6h
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
timeout -k 10 400 python bench.py --workload cfg5 > $out/r03_g_bench_cfg5.json 2> $out/b5.err; echo "cfg5 rc=$?"; cut -c1-250 $out/r03_g_bench_cfg5.json
timeout -k 10 300 python bench.py > $out/r03_g_bench_cfg3.json 2> $out/b3.err; echo "cfg3 rc=$?"; cut -c1-250 $out/r03_g_bench_cfg3.json
