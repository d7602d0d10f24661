#!/bin/bash
set -o pipefail
out=gpurun_out/r5l
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 python3 tools/pmc_collect.py $out/r03_f_pmc_cfg3.json > $out/pmc.log 2>&1; echo "pmc rc=$?"; tail -4 $out/pmc.log | cut -c1-900
timeout -k 10 120 tools/probes/rowstore_global > $out/rowstore_global.txt 2>&1; cat $out/rowstore_global.txt
