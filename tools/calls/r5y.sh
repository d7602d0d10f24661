#!/bin/bash
set -o pipefail
out=gpurun_out/r5y
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_to_bed.py tests/test_paf_api.py tests/test_cli.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/fuzz_gpu.py 90 $((RANDOM)) bed 2>&1 | tail -1
timeout -k 10 200 python tools/bench_extra.py --cmd bed 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; top=sorted(k.items(), key=lambda x:-x[1])[:8]
print(d['cmd'], d['records'], round(d['records_per_s']/1e6,2),'M rec/s', round(d['seconds']*1e3,2),'ms', 'out GB', round(d['out_bytes']/1e9,2), top, 'sum', round(sum(k.values()),1))"
