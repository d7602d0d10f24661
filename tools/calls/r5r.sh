#!/bin/bash
set -o pipefail
out=gpurun_out/r5r
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_view_stats.py tests/test_paf_api.py tests/test_gpu_parity.py tests/test_filter.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.log
[ $rc -eq 0 ] || exit 1
for c in stats filter; do timeout -k 10 200 python tools/bench_extra.py --cmd $c 2>/dev/null | tail -1 | cut -c1-400; done
