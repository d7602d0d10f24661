#!/bin/bash
set -o pipefail
out=gpurun_out/r5t
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
for m in pipe mism tile bed chain; do timeout -k 10 200 python tools/fuzz_gpu.py 45 $((RANDOM)) $m 2>&1 | tail -1; done
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-330 $out/bench_default.json
