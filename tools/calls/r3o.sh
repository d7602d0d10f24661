#!/bin/bash
# profile sets r03_c of the three workloads (kernel stats, traffic passes, bench line, SQ counters)
set -o pipefail
export TMPDIR=/tmp
for wl in cfg3 cfg4 cfg5; do
  bash tools/profile_workload.sh r03_c $wl; echo "profile $wl rc=$?"
  python3 tools/pmc_collect.py gpurun_out/profile_r03_c_$wl/r03_c_pmc_$wl.json --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --no-kernel-events > gpurun_out/profile_r03_c_$wl/pmc.log 2>&1; echo "pmc $wl rc=$?"
done
