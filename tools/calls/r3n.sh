#!/bin/bash
set -o pipefail
out=gpurun_out/r3n
mkdir -p $out
export TMPDIR=/tmp
run() { tag=$1; shift; timeout -k 10 200 python bench.py --steps 30 --cpu-sample 0 "$@" > $out/$tag.json 2> $out/$tag.err; echo "$tag rc=$? $(python -c "import json,sys; d=json.loads(open('$out/$tag.json').read().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'].get('k_size_wave'), d['kernel_ms'].get('k_size_lds'), d['kernel_ms'].get('k_emit_rows'), d['kernel_ms'].get('k_emit_line'))")"; }
run c3_default
PAFFY_WAVE_OPS=1632 PAFFY_WAVE_BYTES=4800 run c3_1632
PAFFY_WAVE_OPS=1280 PAFFY_WAVE_BYTES=3700 run c3_1280
PAFFY_WAVE_OPS=1024 PAFFY_WAVE_BYTES=3000 run c3_1024
PAFFY_WAVE_OPS=3072 PAFFY_WAVE_BYTES=9000 run c3_3072
PAFFY_WAVE_OPS=4096 PAFFY_WAVE_BYTES=12000 run c3_4096
run c4_default --workload cfg4
PAFFY_WAVE_OPS=1632 PAFFY_WAVE_BYTES=4800 run c4_1632 --workload cfg4
PAFFY_WAVE_OPS=3072 PAFFY_WAVE_BYTES=9000 run c4_3072 --workload cfg4
