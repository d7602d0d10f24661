#!/usr/bin/env python3
"""Run bench.py under rocprofv3 --pmc (one pass per counter group) and print per-kernel sums as JSON.

usage (on the GPU box, from the repo root):
    python tools/pmc_collect.py OUT.json [bench.py arguments...]
Counters are collected in passes of their own (never together with a trace domain).
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

GROUPS = [
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"],
    ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"],
    ["SQ_INST_LEVEL_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_IFETCH", "SQ_ACTIVE_INST_ANY"],
]


def main():
    out = sys.argv[1]
    bench_args = sys.argv[2:] or ["--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--no-kernel-events"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ.setdefault("TMPDIR", "/tmp")
    res = {}
    for gi, grp in enumerate(GROUPS):
        d = os.path.join(root, "gpurun_out", f"pmc_pass{gi}")
        shutil.rmtree(d, ignore_errors=True)  # an earlier run's files would be summed in
        cmd = ["rocprofv3", "--pmc", *grp, "-d", d, "--output-format", "csv", "--", sys.executable, os.path.join(root, "bench.py"), *bench_args]
        print(f"pass {gi}: {' '.join(grp)}", flush=True)
        r = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        if r.returncode:
            print(f"pass {gi} failed:\n{r.stdout[-2000:]}", file=sys.stderr)
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"]
                    e = res.setdefault(k, {})
                    e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0) + float(row["Counter_Value"])
                    if gi == 0 and row["Counter_Name"] == "SQ_WAVES":
                        e["dispatches"] = e.get("dispatches", 0) + 1
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for k, e in res.items():
        w = e.get("SQ_WAVES", 0)
        if w and ("k_emit" in k or "k_size" in k or "k_cov" in k or "k_flat" in k or "k_sep" in k or "k_header" in k):
            print(k, {c: round(v / w, 1) for c, v in e.items() if c.startswith("SQ_") and c != "SQ_WAVES"})


if __name__ == "__main__":
    main()
