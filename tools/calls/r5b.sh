#!/bin/bash
set -o pipefail
out=gpurun_out/r5b
mkdir -p $out
timeout -k 10 120 tools/probes/rowstore_global > $out/rowstore_global3.txt 2>&1; echo "probe rc=$?"; cat $out/rowstore_global3.txt
timeout -k 10 120 tools/probes/write_pattern > $out/write_pattern.txt 2>&1; echo "probe rc=$?"; cat $out/write_pattern.txt
