#!/bin/bash
set -o pipefail
out=gpurun_out/r5g
mkdir -p $out
export TMPDIR=/tmp
for lib in "" "$PWD/paffy_amd/abl/libpaffy_hip_flushsc.so" "" "$PWD/paffy_amd/abl/libpaffy_hip_flushsc.so"; do
  if [ -n "$lib" ]; then export PAFFY_HIP_LIB=$lib; else unset PAFFY_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 40 --cpu-sample 0 > $out/b.json 2> $out/b.err; echo "rc=$? [$lib]"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5g/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print('   ', d['value'], d['ms_per_step'], r['frac'], r['kernel_event_sum_ms'], r['dominant_avg_kernel_ms'])
PY
done
unset PAFFY_HIP_LIB
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
for m in pipe mism tile bed chain; do timeout -k 10 200 python tools/fuzz_gpu.py 60 $((RANDOM)) $m 2>&1 | tail -1; done
