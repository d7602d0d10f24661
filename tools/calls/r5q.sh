#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
for c in bed chain invert filter trim add remove stats dedupe tile; do timeout -k 10 200 python tools/bench_extra.py --cmd $c 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['kernel_ms']; top=sorted(k.items(), key=lambda x:-x[1])[:5]
print(d['cmd'], d['records'], round(d['records_per_s']/1e6,2),'M rec/s', round(d['seconds']*1e3,2),'ms', 'out GB', round(d['out_bytes']/1e9,2), top)"; done
