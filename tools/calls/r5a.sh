#!/bin/bash
set -o pipefail
out=gpurun_out/r5a
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 120 tools/probes/rowstore_global > $out/rowstore_global.txt 2>&1; echo "probe rc=$?"; cat $out/rowstore_global.txt
timeout -k 10 300 python bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "bench rc=$?"; cut -c1-600 $out/bench_cfg3.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
