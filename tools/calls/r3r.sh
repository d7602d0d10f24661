#!/bin/bash
set -o pipefail
out=gpurun_out/r3r
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_edges.py tests/test_gpu_tile.py tests/test_cli.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
run() { tag=$1; shift; timeout -k 10 300 python bench.py --cpu-sample 0 "$@" > $out/$tag.json 2> $out/$tag.err; echo "$tag rc=$? $(python -c "import json,sys; d=json.loads(open('$out/$tag.json').read().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['kernel_ms'].items() if k.startswith('k_sep') or k.startswith('k_scan_t') or k=='k_header'})")"; }
run c3_new --steps 40
run c5_new --workload cfg5
run c2_new --workload cfg2 --steps 40
