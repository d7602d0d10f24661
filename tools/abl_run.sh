#!/bin/bash
# run bench.py once per ablation build under paffy_amd/abl/ (timings only; outputs are wrong by design)
for f in paffy_amd/abl/libpaffy_hip_*.so; do
  echo "== $f"
  PAFFY_HIP_LIB=$PWD/$f timeout -k 10 200 python bench.py --steps 20 --cpu-sample 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['kernel_ms'])" || exit 1
done
