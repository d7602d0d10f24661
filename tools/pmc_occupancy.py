#!/usr/bin/env python3
"""Occupancy counters of the record kernels (one rocprofv3 --pmc pass): python tools/pmc_occupancy.py [bench.py arguments...]"""
import csv
import glob
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TMPDIR", "/tmp")
res = {}
for gi, grp in enumerate((["SQ_LEVEL_WAVES", "SQ_WAVES", "GRBM_GUI_ACTIVE", "SQ_CYCLES", "SQ_BUSY_CYCLES"], ["SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAIT_INST_ANY"])):
    d = os.path.join(root, "gpurun_out", f"pmc_occ{gi}")
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc", *grp, "-d", d, "--output-format", "csv", "--", sys.executable, os.path.join(root, "bench.py"), *sys.argv[1:]]
    r = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    if r.returncode:
        print(r.stdout[-2000:])
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if "k_size" not in k and "k_emit" not in k:
                    continue
                e = res.setdefault(k, {})
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0) + float(row["Counter_Value"])
                if row["Counter_Name"] in ("SQ_WAVES", "SQ_WAVE_CYCLES"):
                    e["n_" + row["Counter_Name"]] = e.get("n_" + row["Counter_Name"], 0) + 1
for k, e in res.items():
    print(k[:60], {c: round(v, 1) for c, v in e.items()})
