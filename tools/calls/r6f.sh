#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 900 bash tools/profile_workload.sh r03_g cfg5 2>&1 | tail -3 | cut -c1-300
