#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 600 bash tools/cli_bench.sh 2>&1 | tail -12
