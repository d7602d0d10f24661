#!/usr/bin/env python3
"""Side measurements for DESIGN.md (not the driver's bench contract): `tile` and single commands
on the synthetic stream, records/s with inputs resident in HBM."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=200000)
    ap.add_argument("--mean-ops", type=int, default=2048)
    ap.add_argument("--contigs", type=int, default=24, help="contigs of the synthetic stream (chain: fewer contigs = larger groups)")
    ap.add_argument("--cmd", default="tile", choices=["tile", "invert", "trim", "trimf", "shatter", "remove", "filter", "add", "dedupe", "bed", "stats", "chain"])
    a = ap.parse_args()
    import torch

    import paffy_amd

    eng = paffy_amd.Engine()
    if a.cmd == "add":
        # cfg4: 24 + 24 contigs of 50-250 Mb generated on the device, records on homologous bases (2 % substitutions)
        t0 = time.perf_counter()
        if os.environ.get("PAFFY_X_SMALL_GENOME"):  # experiment: genomes small enough to stay in the caches
            eng.synth4_setup(0x5EED0004, a.mean_ops, n_contigs=4, tlen_min=2_000_000, tlen_span=1_000_000)
        else:
            eng.synth4_setup(0x5EED0004, a.mean_ops)
        print(f"cfg4 genomes resident in HBM ({time.perf_counter() - t0:.1f} s to generate)", file=sys.stderr)
        buf, nbytes = eng.synth4(0, a.records)
    else:
        buf, nbytes = eng.synth(0x5EED0005, a.mean_ops, 0, a.records, n_contigs=a.contigs)
    torch.cuda.synchronize()
    kinds = {"invert": paffy_amd.INVERT, "trim": paffy_amd.TRIM_IDENTITY, "shatter": paffy_amd.SHATTER, "remove": paffy_amd.REMOVE_MISMATCHES, "filter": paffy_amd.FILTER}
    eng.set_filter(min_identity=0.9)
    kinds["add"] = paffy_amd.ADD_MISMATCHES
    kinds["stats"] = paffy_amd.STATS
    res = []
    eng.profile(True)
    for rep in range(3):
        t0 = time.perf_counter()
        if a.cmd == "bed":
            opts = paffy_amd.engine.BedOpts(0, 0, 0, 1, 1)  # -n: both sides of every record
            info = paffy_amd.engine.PlanInfo()
            rc = paffy_amd.engine.lib().paffy_hip_bed_plan(eng._ctx, buf.data_ptr(), nbytes, opts, info)
            assert rc == 0, rc
        elif a.cmd == "chain":
            L = paffy_amd.engine.lib()
            assert L.paffy_hip_chain_begin(eng._ctx) == 0 and L.paffy_hip_chain_add(eng._ctx, buf.data_ptr(), nbytes) == 0
            info = paffy_amd.engine.PlanInfo()
            opts = paffy_amd.engine.ChainOpts(5000, 1, 1000000, 1.0)
            assert L.paffy_hip_chain_run(eng._ctx, opts, info) == 0
        elif a.cmd == "dedupe":
            paffy_amd.engine.lib().paffy_hip_dedupe_reset(eng._ctx)
            info = eng.dedupe_plan(buf, nbytes, True)
        else:
            info = eng.tile_plan(buf, nbytes) if a.cmd == "tile" else eng.plan([paffy_amd.stage(paffy_amd.TRIM_FIXED, 0.05, 0.1) if a.cmd == "trimf" else paffy_amd.stage(kinds[a.cmd])], buf, nbytes)
        out = eng.alloc_out(info.out_bytes)
        eng.emit(out)
        eng.sync()
        dt = time.perf_counter() - t0
        assert info.error.code == 0
        res.append(dt)
    dt = min(res)
    prof = {k: round(v[0] / max(1, v[1]), 3) for k, v in eng.profile_read().items()}
    print(json.dumps({"cmd": a.cmd, "records": a.records, "mean_ops": a.mean_ops, "in_bytes": nbytes, "out_bytes": int(info.out_bytes),
                      "seconds": round(dt, 4), "records_per_s": round(a.records / dt, 1),
                      "GBps": round((nbytes + info.out_bytes) / dt / 1e9, 1), "kernel_ms": prof}))


if __name__ == "__main__":
    main()
