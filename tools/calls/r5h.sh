#!/bin/bash
# final profile sets of the round: r03_f (cfg3, cfg2, cfg4, cfg5)
set -o pipefail
export TMPDIR=/tmp
for wl in cfg3 cfg2 cfg4 cfg5; do
  echo "== $wl"; timeout -k 10 700 bash tools/profile_workload.sh r03_f $wl 2>&1 | tail -3 | cut -c1-400 || { echo "FAILED $wl"; exit 1; }
done
