#!/usr/bin/env python3
"""bench.py -- PAF records/s of the fused `invert | trim | shatter` pipe on MI355X.

A step is one pass of the hot path (plan + emit through the C-ABI) over one batch of synthetic
PAF text that is already resident in HBM. Workloads follow BASELINE.json / SURVEY 8d:
  cfg3 (default): 10M-record stream, mean 2048 cigar ops, `invert | trim | shatter`
  cfg2          :  1M-record stream, mean  512 cigar ops, `shatter`
Every step takes the next `--batch` records of the stream (rank r of N takes every N-th batch),
so K steps process K*batch distinct records per GPU (weak scaling). `--gpus N` starts the N ranks itself
when no launcher did (one process per GPU, RCCL); the ordered write across ranks -- an all-gather of
every step's output sizes, from which each rank knows its byte range -- runs inside the timed region.

Prints ONE JSON line on rank 0 (see the driver contract): value = records/s over all GPUs;
`roofline` prices one pass of the hot path over one batch: algorithmic bytes = input line bytes +
output line bytes of the batch over the summed HIP-event durations of the step's kernels;
`roofline_by_kernel` prices every record kernel with the bytes it moves itself; `cpu_baseline`
times the CPU oracle (a port of the reference algorithm; the reference cannot be built here)
single-threaded on a bounded sample of the same records.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    "cfg3": dict(seed=0x5EED0003, mean_ops=2048, total=10_000_000, pipe="invert|trim|shatter",
                 desc="10M synthetic PAF records, mean 2k cigar ops, invert | trim | shatter"),
    "cfg2": dict(seed=0x5EED0002, mean_ops=512, total=1_000_000, pipe="shatter",
                 desc="1M synthetic PAF records, mean 512 cigar ops, shatter"),
    # records on homologous bases of two device-generated genomes (24 + 24 contigs of 50-250 Mb, 2 % substitutions)
    "cfg4": dict(seed=0x5EED0004, mean_ops=2048, total=10_000_000, pipe="add_mismatches", genomes=True,
                 desc="10M synthetic PAF records, mean 2k cigar ops + 2x3.6 Gb synthetic genomes resident in HBM, add_mismatches"),
    # `paffy tile` over ONE record set per step, --batch records per GPU, partitioned by query contig across the GPUs (all-to-all of the
    # lines), tiled, and ordered by an all-gather of 32-byte keys; BASELINE's cfg5 is --gpus 8 --batch 10000000 --steps 1
    "cfg5": dict(seed=0x5EED0005, mean_ops=2048, total=80_000_000, pipe="tile", tile=True,
                 desc="synthetic PAF records, mean 2k cigar ops, partitioned by query contig across the GPUs, tile + ordered write"),
}


def stages_for(pipe, mod):
    kinds = {"invert": mod.INVERT, "trim": mod.TRIM_IDENTITY, "shatter": mod.SHATTER, "add_mismatches": mod.ADD_MISMATCHES}
    return [mod.stage(kinds[k]) for k in pipe.split("|")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed batches per GPU (default 75: 75 x 131072 = the 10M-record stream of cfg3; cfg5: 4)")
    ap.add_argument("--warmup", type=int, default=None, help="default 2")
    ap.add_argument("--batch", type=int, default=None, help="records per step per GPU (default 131072; cfg5: 2000000, a fifth of a GPU's share of the 80M records)")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a one-GPU rehearsal)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even with one rank, so that the collectives of the N-rank path (RCCL with the nccl backend) run at world size 1")
    ap.add_argument("--pipe", default=None, help="override the workload's command pipe, e.g. 'shatter' (experiments only)")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="records of the stream timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--mean-ops", type=int, default=0, help="override the workload's mean cigar ops (experiments only)")
    ap.add_argument("--kernel-events", choices=("dominant", "all", "none"), default="dominant",
                    help="HIP events inside the timed region: around the launches of the dominant kernel only (default; the other kernels are "
                         "timed in a short instrumented pass after the timed region), around every kernel (costs a step of a dozen kernels 3 %%), or none")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="stream workloads: contexts that take the batches in turn, each on its own HIP stream, so that the sizing pass of batch k + 1 runs beside "
                         "the writers of batch k (1 = one context, batches strictly one after the other; default 2, 1 for the workload with genomes)")
    ap.add_argument("--keep-cache", action="store_true", help="cfg5: do not return torch's cached blocks to the driver between steps (experiments only)")
    ap.add_argument("--tile-text-batch", type=int, default=200_000, help="cfg5: records per text batch handed to the library (each batch stays below 2 GiB)")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region, gather the ordered output of the last step(s) on rank 0 (the batches travel there in batch order: RCCL send / recv with the "
                         "nccl backend) and compare it with a one-process pass over the same records; small --batch only. cfg5: the tile of the last step")
    ap.add_argument("--rehearse", action="store_true",
                    help="plumbing rehearsal without a GPU: ranks, rendezvous, the per-step size exchange and the reductions run, the hot path does not (value is null)")
    args = ap.parse_args()
    tile_wl = bool(WORKLOADS[args.workload].get("tile"))
    if args.steps is None:
        args.steps = 4 if tile_wl else 75
    if args.warmup is None:
        args.warmup = 2
    if args.batch is None:
        args.batch = 2_000_000 if tile_wl else 131072
    if args.pipeline is None:
        args.pipeline = 1 if (tile_wl or WORKLOADS[args.workload].get("genomes")) else 2
    if args.no_kernel_events:
        args.kernel_events = "none"

    # `python bench.py --gpus N` outside a launcher: start the N ranks here, before anything touches the GPU
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.rehearse:
        return rehearse(args)

    import torch

    import paffy_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        torch.cuda.set_device(0 if args.one_device else local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    wl = dict(WORKLOADS[args.workload])
    if args.pipe:
        wl["pipe"] = args.pipe
    if args.mean_ops:
        wl["mean_ops"] = args.mean_ops
    if wl.get("tile"):
        return bench_tile(args, wl, rank, world, dist, dev)
    eng = paffy_amd.Engine()
    # --pipeline P: P contexts (workspace + stream each) take the batches in turn; the host plans batch k + 1 on one stream while the
    # GPU still writes batch k on another. Every batch is still one whole pass of the hot path (plan + emit).
    engs = [eng] + [paffy_amd.Engine() for _ in range(max(1, args.pipeline) - 1)]
    if len(engs) > 1:
        for e in engs:
            e.use_stream(torch.cuda.Stream(device=dev))
    stages = [] if wl.get("tile") else stages_for(wl["pipe"], paffy_amd)
    n_batches = args.warmup + args.steps

    # ---- synthetic input, generated on the device, resident before the timed region ----
    from paffy_amd import shard

    batches = []
    mine = shard.batches_of_rank(rank, world, n_batches * world * args.batch, args.batch)  # rank r: batches r, r+N, ...
    if wl.get("genomes"):
        if len(engs) > 1:
            raise SystemExit("--pipeline > 1 is for the stream workloads without genomes (every context would hold its own copy)")
        eng.synth4_setup(wl["seed"], wl["mean_ops"])  # both genomes replicated on every GPU
    for _, first, n in mine:
        r0 = first % max(1, wl["total"] - args.batch + 1)  # weak scaling: past the end the stream repeats
        buf, nbytes = eng.synth4(r0, n) if wl.get("genomes") else eng.synth(wl["seed"], wl["mean_ops"], r0, n)
        batches.append((buf, nbytes, r0))
    torch.cuda.synchronize()
    aligned_per_record = 0.0
    extra_per_base = 4.0 if wl.get("tile") else 2.0  # tile: 2 B read + 2 B write of the counter of every aligned base
    if wl.get("genomes") or wl.get("tile"):  # add_mismatches also reads one byte of each genome per aligned base (PAF column 10 here)
        head = bytes(batches[0][0][: batches[0][1]].cpu().numpy().tobytes())
        lines = head.split(b"\n")[:-1]
        aligned_per_record = sum(int(l.split(b"\t", 10)[9]) for l in lines) / max(1, len(lines))

    def plan(buf, nbytes):
        return eng.tile_plan(buf, nbytes) if wl.get("tile") else eng.plan(stages, buf, nbytes)

    # one untimed plan to size the output slabs (one per context, reused by every step)
    eng.sync()
    torch.cuda.synchronize()
    info0 = plan(batches[0][0], batches[0][1])
    out_cap = [int(info0.out_bytes * 1.25) + (1 << 20)] * len(engs)
    d_out = [e.alloc_out(out_cap[0]) for e in engs]

    def step(i):
        k = i % len(engs)
        e = engs[k]
        buf, nbytes, _ = batches[i]
        info = e.tile_plan(buf, nbytes) if wl.get("tile") else e.plan(stages, buf, nbytes)
        if info.error.code:
            raise RuntimeError(f"synthetic record failed: code {info.error.code} record {info.error.record}")
        if info.out_bytes > out_cap[k]:
            e.sync()
            out_cap[k] = int(info.out_bytes * 1.25)
            d_out[k] = e.alloc_out(out_cap[k])
        e.emit(d_out[k])
        return info

    for i in range(max(args.warmup, len(engs) if args.warmup else 0)):
        step(i % n_batches)
    for e in engs:
        e.sync()

    # the kernel the roofline object names: bracketed with HIP events on its launch stream inside the timed region
    dominant = "k_emit_rows" if wl["pipe"].endswith("shatter") else ("k_size_lds" if wl.get("genomes") else "k_emit_line")
    for e in engs:
        e.profile(args.kernel_events != "none", only=dominant if args.kernel_events == "dominant" else None)
    # the ordered write across ranks: every step's output sizes are exchanged (RCCL all-gather) inside the timed region, and each
    # rank's bytes stay in its own HBM at a known offset of the ordered output (per-rank ranges, no gather of the bytes)
    exch = SizeExchange(dist, world, args.steps, dev if (dist is None or args.dist_backend == "nccl") else "cpu")
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infos = []
    for i in range(args.steps):
        infos.append(step(args.warmup + i))
        exch.post(i, infos[-1].out_bytes)
    for e in engs:
        e.sync()
    my_offsets, ordered_total = exch.finish(rank)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernels = {}
    for e in engs:
        for name, (ms, cnt) in e.profile_read().items():
            have = kernels.get(name, (0.0, 0))
            kernels[name] = (have[0] + ms, have[1] + cnt)
        e.profile(False)
    timed_kernels = dict(kernels)
    prof_steps = 0
    if args.kernel_events == "dominant":
        # every kernel of a step, bracketed, in a pass of its own behind the timed region: one context, the batches one after the other
        prof_steps = min(args.steps, 8)
        eng.profile(True)
        for i in range(prof_steps):
            info = eng.plan(stages, batches[args.warmup + i][0], batches[args.warmup + i][1])
            eng.emit(d_out[0])
        eng.sync()
        kernels = eng.profile_read()
        eng.profile(False)

    verified = None
    if args.verify:
        verified = verify_stream(args, wl, eng, stages, rank, world, dist, dev if (dist is None or args.dist_backend == "nccl") else "cpu", mine, batches)
    if dist:
        red_dev = dev if args.dist_backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # whole-job byte and row counts (every rank holds only its own batches)
        c = torch.tensor([float(sum(i.in_bytes for i in infos)), float(sum(i.out_bytes for i in infos)),
                          float(sum(i.n_rows for i in infos))], dtype=torch.float64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        job_in, job_out, job_rows = (float(x) for x in c.tolist())

    records = args.batch * args.steps * world
    in_bytes = sum(i.in_bytes for i in infos)    # rank 0's own batches: what one k_emit_lds launch moves
    out_bytes = sum(i.out_bytes for i in infos)
    rows = sum(i.n_rows for i in infos)
    if not dist:
        job_in, job_out, job_rows = float(in_bytes), float(out_bytes), float(rows)

    if rank == 0:
        # One launch of the hot path = one pass over one batch = every kernel of a step. The headline `roofline` prices that pass:
        # the batch's algorithmic bytes (input + output line bytes, + the per-base bytes of add_mismatches / tile) over
        # the summed HIP-event time of the step's kernels. No single kernel moves all of those bytes -- the sizing kernel reads
        # the text, the writers write the lines -- so `roofline_by_kernel` prices each kernel with the algorithmic bytes that
        # kernel itself moves, and carries its measured HBM traffic where a PMC pass of this workload is committed.
        per_step_in, per_step_out = in_bytes / args.steps, out_bytes / args.steps
        per_step_extra = extra_per_base * aligned_per_record * args.batch
        per_launch_bytes = per_step_in + per_step_out + per_step_extra
        readers = ("k_size_lds", "k_tile_walk", "k_tile_slices", "k_bed_cover")
        writers = ("k_emit_rows", "k_emit_lds", "k_emit_lds<line>", "k_emit_line", "k_tile_emit", "k_arena_emit", "k_arena_emit<line>")
        by_kernel = {}
        if "k_size_wave" in kernels and "k_size_lds" in kernels:
            # the sizing pass is two launches of one kernel: short cigars one wave per record, the rest four waves per record;
            # together they read the batch's text once
            ms = kernels["k_size_wave"][0] / max(1, kernels["k_size_wave"][1]) + kernels["k_size_lds"][0] / max(1, kernels["k_size_lds"][1])
            own = per_step_in + per_step_extra
            tr = [measured_traffic(args, k) for k in ("k_size_wave", "k_size_lds")]
            by_kernel["k_size_wave+k_size_lds"] = {"avg_kernel_ms": round(ms, 4), "own_algorithmic_bytes": int(own), "achieved": round(own / (ms * 1e-3) / 1e9, 1),
                                                   "frac": round(own / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": int(sum(tr)) if all(tr) else None}
        for name, (ms, launches) in kernels.items():
            if launches <= 0 or ms <= 0 or name not in readers + writers or (name == "k_size_lds" and "k_size_wave" in kernels):
                continue
            own = per_step_in + per_step_extra if name in readers else per_step_out
            ach = own / (ms / launches * 1e-3) / 1e9
            by_kernel[name] = {"avg_kernel_ms": round(ms / launches, 4), "own_algorithmic_bytes": int(own), "achieved": round(ach, 1),
                               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args, name)}
        roofline = None
        if kernels:
            k_steps = prof_steps if prof_steps else args.steps  # steps the per-kernel sums were taken over
            dom = dominant if dominant in kernels else max(kernels, key=lambda k: kernels[k][0])
            kernel_sum_ms = sum(ms for ms, _ in kernels.values()) / k_steps  # all kernels of one step, HIP events on the launch stream
            step_ms = kernel_sum_ms
            priced_on = "summed HIP-event time of the step's kernels (timed region)"
            if len(engs) > 1 or prof_steps:
                # pipelined contexts: kernels of neighbouring batches run beside each other, so their event times add up to more than
                # the time a step takes; and with the dominant kernel alone bracketed in the timed region the other kernels' times come
                # from the instrumented pass behind it. The pass is then priced with the wall time of a step (barrier to barrier / steps),
                # launch gaps and the host's part included.
                step_ms = elapsed / args.steps * 1e3
                priced_on = "wall time of a step (barrier to barrier / steps)"
            achieved = per_launch_bytes / (step_ms * 1e-3) / 1e9
            traffic = [measured_traffic(args, k) for k in kernels]
            live = timed_kernels.get(dom)  # HIP events around the dominant kernel's launches inside the timed region
            dom_ms = live[0] / max(1, live[1]) if live and live[1] else kernels[dom][0] / max(1, kernels[dom][1])
            roofline = {"bound": "hbm", "kernel": f"all kernels of one step (dominant: {dom})", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": int(sum(t for t in traffic if t)) if any(traffic) else None,
                        "priced_on": priced_on,
                        "step_kernels_ms": round(step_ms, 4), "kernel_event_sum_ms": round(kernel_sum_ms, 4),
                        "kernel_events": args.kernel_events, "instrumented_pass_steps": prof_steps,
                        "pipeline_contexts": len(engs), "dominant_kernel": dom,
                        "dominant_avg_kernel_ms": round(dom_ms, 4), "dominant_launches_timed": int(live[1]) if live else 0,
                        # the prescribed form: the step's algorithmic bytes over the dominant kernel's own average duration
                        "dominant_frac": round(per_launch_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": int(per_launch_bytes)}
        cpu = cpu_all = e2e = None
        if args.cpu_sample > 0 and world == 1:  # the CPU leg runs on rank 0 of the one-GPU run only
            if wl.get("genomes"):
                cpu = cpu_baseline_genomes(eng, wl, stages, min(args.cpu_sample, 16384))
            else:
                cpu = cpu_baseline(eng, wl, stages, args.cpu_sample)
                cpu_all = cpu_baseline_all_cores(eng, wl, stages, max(1024, args.cpu_sample // 8))
                e2e = end_to_end(eng, stages, batches[:2], args.batch)
                e2e["link"] = pcie_probe(dev)
        line = {
            "metric": "PAF records/sec (shatter+invert+trim pipe); % HBM roofline at 1/2/4/8 GPU",
            "value": round(records / elapsed, 1),
            "unit": "records/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int64 (float32 identity predicate)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "pipe": wl["pipe"], "records_per_step_per_gpu": args.batch, "pipeline_contexts": len(engs),
                       "records_timed": records, "stream_records": wl["total"], "mean_cigar_ops": wl["mean_ops"],
                       "input_bytes_per_record": round(in_bytes / (args.batch * args.steps), 1),
                       "output_bytes_per_record": round(out_bytes / (args.batch * args.steps), 1),
                       "output_rows_per_record": round(rows / (args.batch * args.steps), 2),
                       "aligned_bases_per_record": round(aligned_per_record, 1),
                       "sharding": "batch b of the stream on rank b % N; per step one all-gather of the output sizes (8 B per rank) inside the timed "
                                   "region gives every rank the offset of its bytes in the ordered output; the bytes themselves stay on their GPU"},
            "ordered_write": {"mode": "per-rank offsets from an all-gather of the step's output sizes", "total_bytes": ordered_total,
                              "rank0_first_offsets": [int(x) for x in my_offsets[:4].tolist()], "verified_against_one_process": verified},
            "whole_path_GBps_per_gpu": round(((job_in + job_out) / world + extra_per_base * aligned_per_record * args.batch * args.steps) / elapsed / 1e9, 1),
            "roofline": roofline,
            "roofline_by_kernel": {k: v for k, v in by_kernel.items() if "+" in k or kernels[k][0] >= 0.05 * max(x[0] for x in kernels.values())},
            "whole_path_frac_of_hbm_peak": round(((job_in + job_out) / world + extra_per_base * aligned_per_record * args.batch * args.steps) / elapsed / 1e9 / HBM_PEAK_GBS, 4),
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "end_to_end": e2e,
            "kernel_ms": {k: round(v[0] / max(1, v[1]), 4) for k, v in kernels.items()},
            # the lean pipes are sized by the flat pass (paffy_amd/csrc/flat_kernel.h); what it left to the record kernels in the last plan
            "flat_pass": dict(zip(("records_left_to_record_kernels", "by_reason"), eng.flat_stats())),
        }
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


def bench_tile(args, wl, rank, world, dist, dev):
    """cfg5: a step is `paffy tile` over one set of world x --batch records. Every rank generates a contiguous share of the set on
    its GPU (untimed); timed: the partition by query contig (hash + regroup on the device, one all-to-all of the lines), the tile of
    what the rank owns (bitmaps, slice walk, merge), the all-gather of the 32-byte keys that places every line in the ordered output,
    and the local write of the lines."""
    import torch

    import paffy_amd
    from paffy_amd import shard

    eng = paffy_amd.Engine()
    worker = shard.GpuTileWorker(eng)
    comm = dev if (dist is None or args.dist_backend == "nccl") else "cpu"
    per_batch = args.tile_text_batch  # records per text batch (default 200 000: about 1.1 GB of text, below the 2 GiB of a batch)

    def share(step_no):
        first, n = shard.share_of_rank(rank, world, world * args.batch)
        first += step_no * world * args.batch
        out = []
        for r0 in range(first, first + n, per_batch):
            buf, nbytes = eng.synth(wl["seed"], wl["mean_ops"], r0 % wl["total"], min(per_batch, first + n - r0))
            out.append((buf, nbytes))
        return out, first

    n_steps = args.warmup + args.steps
    in_bytes = out_bytes = rows = 0
    aligned_per_record = 0.0
    eng.profile(False)
    elapsed, last, kernels, each = 0.0, None, {}, []
    for i in range(n_steps):
        last = res = out = None  # the step before: its output, keys and text are let go before this step's input is generated
        worker.release()
        if not args.keep_cache:
            # untimed: torch's cached blocks go back to the driver, so the step allocates its buffers as the first tile of a process does.
            # (Left cached, the blocks of the step before fit this step's slightly different sizes only by luck; with a share of 54 GB per
            # buffer the allocator then frees and re-allocates everything inside the timed region: 2.3 s instead of 0.73 s per step.)
            torch.cuda.empty_cache()
        batches, first = share(i)
        if i == 0:
            head = bytes(batches[0][0][: min(batches[0][1], 4 << 20)].cpu().numpy().tobytes())
            lines = head.split(b"\n")[:-1]
            aligned_per_record = sum(int(l.split(b"\t", 10)[9]) for l in lines) / max(1, len(lines))
        timed = i >= args.warmup
        if i == args.warmup:
            eng.profile(not args.no_kernel_events)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = shard.tile_sharded(worker, dist, rank, world, batches, first, comm, consume=True)  # the batches are handed over: freed as they are split
        out = worker.emit()
        eng.sync()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if timed:
            elapsed += dt
            each.append(round(dt * 1e3, 2))
            in_bytes += sum(n for _, n in worker.keep)
            out_bytes += int(out.numel())
            rows += int(res["keys"].shape[0])
        last = (res, out, None, first)
        free_b, total_b = torch.cuda.mem_get_info()
        hbm_used_after_step = total_b - free_b
    kernels = eng.profile_read()
    eng.profile(False)
    red = dev if (dist is not None and args.dist_backend == "nccl") else "cpu"
    # what every rank ended up owning in the last step: the partition by query contig (heaviest contig to the lightest rank) bounds the
    # N-GPU efficiency before any link does, so the line shows it
    mine_load = torch.tensor([sum(n for _, n in worker.keep), int(last[0]["keys"].shape[0])], dtype=torch.int64, device=red)
    loads = mine_load.reshape(1, 2)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        flat_loads = torch.zeros(world * 2, dtype=torch.int64, device=red)
        dist.all_gather_into_tensor(flat_loads, mine_load)
        loads = flat_loads.reshape(world, 2)
    loads = loads.cpu().tolist()
    verified = None
    if args.verify:
        res, out, batches, first = last
        whole = shard.gather_ordered_output(worker, dist, rank, world, out, res["keys"][:, 3].contiguous(), res["offsets"], res["total"], comm)
        if rank == 0:
            e2 = paffy_amd.Engine()
            allb = []
            base = (n_steps - 1) * world * args.batch
            for r0 in range(base, base + world * args.batch, per_batch):
                allb.append(e2.synth(wl["seed"], wl["mean_ops"], r0 % wl["total"], min(per_batch, base + world * args.batch - r0)))
            info = e2.tile_batches(allb)
            ref = e2.alloc_out(info.out_bytes)
            e2.emit(ref)
            e2.sync()
            verified = bool(info.error.code == 0 and info.out_bytes == res["total"] and torch.equal(ref[: info.out_bytes].cpu(), whole.cpu()))
            e2.close()
    if rank == 0:
        records = world * args.batch * args.steps
        per_step = (in_bytes + out_bytes) / max(1, args.steps) + 4.0 * aligned_per_record * (rows / max(1, args.steps))  # this rank's share of a step
        kernels = {k: v for k, v in kernels.items() if not k.startswith("k_synth")}  # the next step's input is generated beside the timed region
        step_ms = sum(ms for ms, _ in kernels.values()) / max(1, args.steps)
        roofline, by_kernel = None, {}
        if kernels and step_ms > 0:
            dom = max(kernels, key=lambda k: kernels[k][0])
            ach = per_step / (step_ms * 1e-3) / 1e9
            traffic = tile_step_traffic(args, kernels)
            roofline = {"bound": "hbm", "kernel": f"all kernels of one step on one rank (dominant: {dom})", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                        # the honest second figure: the bytes the step's kernels really moved (PMC pass of this workload) over the same time
                        "traffic_frac": (round(traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                        "step_kernels_ms": round(step_ms, 4), "dominant_kernel": dom,
                        "dominant_avg_kernel_ms": round(kernels[dom][0] / max(1, kernels[dom][1]), 4), "algorithmic_bytes_per_launch": int(per_step),
                        "note": "algorithmic bytes = input lines + output lines + 4 B per aligned base (SURVEY 8d counts the counter of every aligned base as 2 B read + 2 B "
                                "written in HBM; here a slice's counters live in LDS while its records are walked, so the walk itself moves one bit per base)"}
            for name, (ms, launches) in kernels.items():
                if launches > 0 and ms >= 0.05 * kernels[dom][0]:
                    by_kernel[name] = {"avg_kernel_ms": round(ms / launches, 4), "launches_per_step": round(launches / max(1, args.steps), 2)}
        cpu = cpu_baseline_tile(eng, wl, min(args.cpu_sample, 8192)) if (args.cpu_sample > 0 and world == 1) else None
        print(json.dumps({
            "metric": "PAF records/sec (tile, records partitioned by query contig + ordered write; BASELINE cfg5); % HBM roofline at 1/2/4/8 GPU",
            "value": round(records / elapsed, 1), "unit": "records/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u16 counters / int64 keys", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}", "pipe": "tile", "records_per_step_per_gpu": args.batch, "records_timed": records,
                       "mean_cigar_ops": wl["mean_ops"], "aligned_bases_per_record": round(aligned_per_record, 1),
                       "input_bytes_per_record": round(in_bytes / max(1, rows), 1), "output_bytes_per_record": round(out_bytes / max(1, rows), 1),
                       "sharding": "records partitioned by query contig (heaviest contig to the lightest rank): one all-to-all of the lines + 8 B per record, "
                                   "then an all-gather of 32 B per record places every line in the ordered output; all inside the timed region"},
            "ordered_write": {"mode": "per-line offsets from the all-gathered keys; lines written by their owner", "total_bytes": int(last[0]["total"]),
                              "verified_against_one_process": verified},
            "rank_load_last_step": {"text_bytes": [int(x[0]) for x in loads], "records": [int(x[1]) for x in loads],
                                    "max_over_mean_bytes": round(max(x[0] for x in loads) * len(loads) / max(1, sum(x[0] for x in loads)), 4)},
            "phase_ms_last_step": last[0].get("timing") or None, "ms_each_step": each,
            "hbm": {"in_use_after_last_step_GB": round(hbm_used_after_step / 1e9, 2), "peak_in_use_GB": (round(last[0]["hbm_peak"] / 1e9, 2) if last[0].get("hbm_peak") else None),
                    "peak_live_GB": (round(last[0]["hbm_live_peak"] / 1e9, 2) if last[0].get("hbm_live_peak") else None),
                    "share_text_GB": round(in_bytes / max(1, args.steps) / 1e9, 2),
                    "note": "device memory in use by this process (hipMemGetInfo: torch's cache and the library's buffers alike); the peaks are sampled at the phase "
                            "boundaries of the sharded tile with PAFFY_SHARD_TIMING=1; live = in use minus the blocks torch's caching allocator holds empty"},
            "roofline": roofline, "roofline_by_kernel": by_kernel, "cpu_baseline": cpu,
            "kernel_ms": {k: round(v[0] / max(1, v[1]), 4) for k, v in kernels.items()},
        }), flush=True)
    eng.close()
    if dist:
        dist.destroy_process_group()


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """GPUs this process could use, counted WITHOUT a HIP / torch.cuda call (torch.cuda.device_count() ends in hipGetDeviceCount when
    amdsmi discovery fails, and the launcher must not have initialised the GPU when it starts its ranks): the KFD topology nodes that
    have SIMDs (CPUs are nodes too, with simd_count 0), cut down by a HIP_/ROCR_/CUDA_VISIBLE_DEVICES list. None when there is a
    /dev/kfd whose topology cannot be read (the ranks then find out themselves); 0 without a KFD device."""
    import glob

    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None if os.path.exists("/dev/kfd") else 0
    have = 0
    for path in nodes:
        try:
            with open(path) as fh:
                for ln in fh:
                    f = ln.split()
                    if len(f) == 2 and f[0] == "simd_count" and int(f[1]) > 0:
                        have += 1
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have


def launch_ranks(args):
    """Start one process per GPU (rank r -> cuda:r, or cuda:0 with --one-device) with the torch.distributed environment set, as
    `python -m torch.distributed.run --nproc-per-node N` would; rank 0 prints the JSON line. Nothing in this process touches
    the GPU, torch is not even imported (no fork / exec from a GPU-initialised process). Returns the exit status: non-zero when
    any rank failed."""
    import subprocess

    n = args.gpus
    if not args.rehearse and not args.one_device:
        have = visible_gpus()  # from sysfs: no HIP call (and no torch) in the process that starts the ranks
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} but {have} device(s) visible (use --one-device for a one-GPU rehearsal)", file=sys.stderr)
            return 2
    env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for r, p in enumerate(procs):
            code = p.wait()
            if code != 0:
                print(f"bench.py: rank {r} exited with status {code}", file=sys.stderr)
                rc = rc or (code if code > 0 else 1)
                for q in procs:  # a rank that lost its peer would wait for the rendezvous timeout: end exactly the ones we started
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


class SizeExchange:
    """The ordered write of the stream commands across ranks: batch b of the stream belongs to rank b % N; after every
    step all ranks exchange the step's output byte counts (one all-gather of 8 bytes per rank: RCCL with the nccl backend), so that
    every rank knows the byte offset of each of its batches in the ordered output and writes its own range there. The collectives
    are issued inside the timed region, one per step, and waited for at its end."""

    def __init__(self, dist, world, steps, device):
        import torch

        self.dist, self.world, self.steps = dist, world, steps
        self.sizes = [[0] * world for _ in range(steps)]  # single rank: nothing leaves the host
        self.work = []
        self.on_gpu = dist is not None and str(device) != "cpu"
        if dist is not None:
            self.table = torch.zeros(steps, world, dtype=torch.int64, device=device)
            self.mine = torch.zeros(steps, dtype=torch.int64, device=device)
        if self.on_gpu:
            # the size is known on the host right after the plan: it travels on a side stream (pinned staging, no host wait), so
            # neither the copy nor the collective queues behind the step's emit kernel
            self.stage = torch.zeros(steps, dtype=torch.int64).pin_memory()
            self.side = torch.cuda.Stream(device=device)

    def post(self, step, out_bytes):
        import torch

        if self.dist is None:
            self.sizes[step][0] = int(out_bytes)
            return
        if self.on_gpu:
            self.stage[step] = int(out_bytes)
            with torch.cuda.stream(self.side):
                self.mine[step:step + 1].copy_(self.stage[step:step + 1], non_blocking=True)
                self.work.append(self.dist.all_gather_into_tensor(self.table[step], self.mine[step:step + 1], async_op=True))
        else:
            self.mine[step] = int(out_bytes)
            self.work.append(self.dist.all_gather_into_tensor(self.table[step], self.mine[step:step + 1], async_op=True))

    def finish(self, rank):
        """-> (offset of each of this rank's batches in the ordered output, total bytes); batch order = step-major, rank-minor"""
        import torch

        for w in self.work:
            w.wait()
        if self.dist is None:
            table = torch.tensor(self.sizes, dtype=torch.int64)
        else:
            if self.on_gpu:
                self.side.synchronize()
            table = self.table.cpu()
        flat = table.reshape(-1)
        ends = torch.cumsum(flat, 0)
        offs = (ends - flat).reshape(self.steps, self.world)[:, rank]
        return offs, int(ends[-1].item()) if flat.numel() else 0


def verify_stream(args, wl, eng, stages, rank, world, dist, comm, mine, batches, last=2):
    """The ordered write of the stream workloads, checked: the outputs of every rank's last `last` timed batches travel to rank 0 in batch
    order (shard.gather_to_writer: point-to-point sends, RCCL over xGMI with the nccl backend; per-batch sizes from an all-reduce) and
    rank 0 compares each with its own pass over the same records of the stream. Returns True / False on rank 0, None elsewhere."""
    import torch

    from paffy_amd import shard

    n_batches = (args.warmup + args.steps) * world
    take = [k for k in range(args.warmup + max(0, args.steps - last), args.warmup + args.steps)]
    local = {}
    for k in take:
        b, _, _ = mine[k]
        buf, nbytes, _ = batches[k]
        info = eng.plan(stages, buf, nbytes)
        out = eng.alloc_out(info.out_bytes)
        eng.emit(out)
        eng.sync()
        local[b] = out[: info.out_bytes]
    if dist is None:
        sizes = [0] * n_batches
        for b, t in local.items():
            sizes[b] = int(t.numel())
    else:
        sizes = shard.gather_batch_sizes(dist, {b: int(t.numel()) for b, t in local.items()}, n_batches, device=comm)
    got = {}
    if dist is None:
        got = dict(local)
    else:
        shard.gather_to_writer(dist, rank, world, {b: (t if str(t.device) == str(comm) else t.to(comm)) for b, t in local.items()}, sizes,
                               lambda b, t: got.__setitem__(b, t.clone()), device=comm)
    if rank != 0:
        return None
    ok = len(got) == sum(1 for s in sizes if s > 0)
    for b in sorted(got):
        r0 = (b * args.batch) % max(1, wl["total"] - args.batch + 1)
        buf, nbytes = eng.synth4(r0, args.batch) if wl.get("genomes") else eng.synth(wl["seed"], wl["mean_ops"], r0, args.batch)
        info = eng.plan(stages, buf, nbytes)
        ref = eng.alloc_out(info.out_bytes)
        eng.emit(ref)
        eng.sync()
        ok = ok and info.out_bytes == sizes[b] and bool(torch.equal(ref[: info.out_bytes].cpu(), got[b].cpu()))
    return bool(ok)


def rehearse(args):
    """`--rehearse`: everything of the N-rank run but the hot path -- rendezvous, barriers, the per-step size exchange, the
    max-over-ranks reduction and the JSON line -- on CPU tensors over gloo. The hot path has no CPU implementation, so nothing is
    measured: value is null. tests/test_bench_ranks.py runs this with --gpus 2."""
    import torch

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo")
    ex = SizeExchange(dist, world, args.steps, "cpu")
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ex.post(i, 1000 * (i + 1) + rank)  # stand-in sizes: batch (step i, rank r) is 1000 (i + 1) + r bytes long
    offs, total = ex.finish(rank)
    if dist:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    want = [sum(1000 * (j + 1) * world + world * (world - 1) // 2 for j in range(i)) + sum(1000 * (i + 1) + r for r in range(rank))
            for i in range(args.steps)]
    ok = [int(x) for x in offs.tolist()] == want
    verified = None
    if args.verify:
        # the --verify plumbing without a GPU: stand-in outputs (batch b = its size in bytes of (b * 31 + position) mod 251) travel to rank 0
        # in batch order over gloo and must arrive as one process would have written them
        from paffy_amd import shard

        def standin(b, n):
            return ((torch.arange(n, dtype=torch.int64) + 31 * b) % 251).to(torch.uint8)

        n_b = args.steps * world
        local = {i * world + rank: standin(i * world + rank, 1000 * (i + 1) + rank) for i in range(args.steps)}
        if dist:
            sizes = shard.gather_batch_sizes(dist, {b: int(t.numel()) for b, t in local.items()}, n_b)
            got = {}
            shard.gather_to_writer(dist, rank, world, local, sizes, lambda b, t: got.__setitem__(b, t.clone()))
        else:
            sizes = [int(local[b].numel()) for b in range(n_b)]
            got = local
        if rank == 0:
            verified = sorted(got) == list(range(n_b)) and all(torch.equal(got[b], standin(b, sizes[b])) for b in got) and sum(sizes) == total
    if rank == 0:
        print(json.dumps({"metric": "PAF records/sec (shatter+invert+trim pipe); % HBM roofline at 1/2/4/8 GPU", "value": None, "unit": "records/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(t.item()) / max(1, args.steps) * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "none", "rehearsal": True,
                          "ordered_write": {"mode": "per-rank offsets from an all-gather of the step's output sizes", "offsets_ok": ok,
                                            "total_bytes": total, "verified_against_one_process": verified}}), flush=True)
    if dist:
        dist.destroy_process_group()
    return 0 if ok and verified is not False else 1


def tile_step_traffic(args, kernels):
    """HBM bytes of one tile step: every kernel's measured bytes per launch (profiles/*traffic*.json of this workload and batch size) times
    its launches per step. None when no such pass is committed or it misses one of the step's heavy kernels."""
    label = {"k_tile_emit": "k_line_emit"}  # launch labels that differ from the kernel's name
    total, seen = 0.0, False
    for name, (ms, launches) in kernels.items():
        t = measured_traffic(args, label.get(name, name))
        if t is None:
            if ms >= 1.0 * max(1, args.steps):  # a kernel of a millisecond or more per step without a figure: no total
                return None
            continue
        seen = True
        total += t * launches / max(1, args.steps)
    return int(total) if seen else None


def measured_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this process;
    tools/traffic_collect.py writes the files): the newest profiles/*traffic*.json taken on this workload and batch size."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            with open(path) as fh:
                t = json.load(fh)
            if t.get("workload") == args.workload and t.get("batch") == args.batch and not args.pipe:
                return t["kernels"][kernel]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline(eng, wl, stages, n):
    """CPU leg: the oracle (port of the reference algorithm, single thread) on the first n records of
    the stream; its output also checks the GPU output for the same records."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    import oracle_lib as O

    buf, nbytes = eng.synth(wl["seed"], wl["mean_ops"], 0, n)
    data = bytes(buf[:nbytes].cpu().numpy().tobytes())
    kinds = {1: O.INVERT, 2: O.TRIM_IDENTITY, 4: O.SHATTER}
    ost = [O.stage(kinds[s.kind], s.p0, s.p1) for s in stages]
    L = O.lib()
    arr = (O.Stage * len(ost))(*ost)
    out, on, err = C.c_void_p(), C.c_int64(), O.Error()
    t0 = time.perf_counter()
    L.po_run(arr, len(ost), data, len(data), None, 0, C.byref(out), C.byref(on), C.byref(err))
    dt = time.perf_counter() - t0
    want = bytes((C.c_char * on.value).from_address(out.value)) if on.value else b""
    L.po_free(out)
    got, _ = eng.run(stages, data)
    return {"value": round(n / dt, 1), "unit": "records/s", "cores": 1, "kind": "port",
            "sample": f"first {n} records of the same stream ({len(data)} B in, {len(want)} B out), {dt:.1f} s, single thread",
            "gpu_output_matches": bool(got == want and err.code == 0)}


def _oracle_share(job):
    """one process of the all-cores CPU leg: the oracle over its share of the sample (module-level: it is pickled to the pool)"""
    path, lo, hi, kinds = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    import oracle_lib as O

    with open(path, "rb") as fh:
        fh.seek(lo)
        data = fh.read(hi - lo)
    L = O.lib()
    ost = [O.stage(k, p0, p1) for k, p0, p1 in kinds]
    arr = (O.Stage * len(ost))(*ost)
    out, on, err = C.c_void_p(), C.c_int64(), O.Error()
    L.po_run(arr, len(ost), data, len(data), None, 0, C.byref(out), C.byref(on), C.byref(err))
    L.po_free(out)
    return on.value if err.code == 0 else -1


def cpu_baseline_all_cores(eng, wl, stages, per_core):
    """CPU leg at N = the host cores this run may use (the reference's own parallelism is one process per split of the
    input, tests/paf_pipeline_test.sh:42-67): N processes, each the single-threaded oracle over a contiguous share of N x per_core
    records of the same stream, read from a RAM-backed file; wall time of the slowest."""
    import multiprocessing as mp
    import tempfile

    cores = min(16, os.cpu_count() or 1)
    n = cores * per_core
    buf, nbytes = eng.synth(wl["seed"], wl["mean_ops"], 0, n)
    data = bytes(buf[:nbytes].cpu().numpy().tobytes())
    kinds = {1: 1, 2: 2, 4: 4}
    st = [(kinds[s.kind], s.p0, s.p1) for s in stages]
    cuts, at = [0], 0
    for k in range(1, cores):  # contiguous shares cut at line ends
        at = data.index(b"\n", max(at, len(data) * k // cores)) + 1
        cuts.append(at)
    cuts.append(len(data))
    d = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.NamedTemporaryFile(dir=d, suffix=".paf", delete=False) as fh:
        fh.write(data)
        path = fh.name
    try:
        ctx = mp.get_context("spawn")  # never fork a process that has initialised the GPU
        with ctx.Pool(cores) as pool:
            pool.map(_oracle_share, [(path, 0, 0, st)] * cores)  # workers up, library loaded
            t0 = time.perf_counter()
            sizes = pool.map(_oracle_share, [(path, cuts[k], cuts[k + 1], st) for k in range(cores)])
            dt = time.perf_counter() - t0
    finally:
        os.unlink(path)
    return {"value": round(n / dt, 1), "unit": "records/s", "cores": cores, "kind": "port",
            "sample": f"first {n} records of the same stream in {cores} contiguous shares, one single-threaded oracle process each ({len(data)} B in, "
                      f"{sum(sizes)} B out), {dt:.1f} s wall", "ok": bool(all(x >= 0 for x in sizes))}


def end_to_end(eng, stages, batches, batch_records):
    """The PCIe-inclusive rate (reported beside `value`, never as it): the same batches as host bytes through the streaming
    runtime of the CLI (paffy_hip_stream_*: pinned staging, H2D / kernels / D2H overlapped), output pieces landing in host memory."""
    chunks = [bytes(buf[:nbytes].cpu().numpy().tobytes()) for buf, nbytes, _ in batches]
    eng.stream_host(stages, chunks)  # both slots' buffers allocated (they stay with the context), kernels loaded
    # three runs, the best one reported and all of them listed (a run whose stream had to allocate a slot's output buffer used to take
    # seconds in that hipMalloc; the buffers now stay with the context between streams: DESIGN 4.2, tools/probes/d2h_pieces.py)
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        records, out_bytes = eng.stream_host(stages, chunks)
        runs.append((dict(eng.stream_seconds), time.perf_counter() - t0))
    sec, dt_call = min(runs, key=lambda r: r[0]["run"])
    dt = sec["run"]  # the stream itself: chunks in, pieces out. Opening it (pinning the host buffers: once per process in the CLI) is listed apart
    return {"value": round(records / dt, 1), "unit": "records/s", "sample": f"{len(chunks)} batches of {batch_records} records as host bytes: "
            f"{sum(len(c) for c in chunks)} B over PCIe in, {out_bytes} B out, {dt:.2f} s (best of 3 runs)", "GBps_out": round(out_bytes / dt / 1e9, 2),
            "runs_GBps_out": [round(out_bytes / r[0]["run"] / 1e9, 2) for r in runs],
            "seconds": {"stream_open": round(sec["open"], 3), "stream_run": round(sec["run"], 3), "of_which_host_copy_into_pinned_slots": round(sec["input_copy"], 3),
                        "stream_close": round(sec["close"], 3), "whole_call": round(dt_call, 3)},
            "records_per_s_whole_call": round(records / dt_call, 1)}


def pcie_probe(dev, mib=256, reps=4):
    """What the link of THIS box gives a plain pinned copy, each direction alone (the end_to_end leg is bounded by the D2H figure: its
    output is 23 times its input): 256 MiB pinned <-> device, the best of `reps` copies, timed with events on the current stream."""
    import torch

    n = mib << 20
    host = torch.empty(n, dtype=torch.uint8).pin_memory()
    devb = torch.empty(n, dtype=torch.uint8, device=dev)
    out = {}
    for name, dst, src in (("h2d_GBps", devb, host), ("d2h_GBps", host, devb)):
        best = 0.0
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dst.copy_(src, non_blocking=True)
            e1.record()
            e1.synchronize()
            best = max(best, n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        out[name] = round(best, 2)
    # the same D2H copy while the GPU is busy: a stream of HBM-saturating element-wise kernels on the current stream, the copies on a side
    # stream. A box whose runtime moves pinned copies with its copy engines keeps (most of) the idle rate; where the copies are shader
    # kernels they queue behind the compute -- which is what the end_to_end leg, whose copies run beside the row writer, then sees.
    try:
        busy = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        for _ in range(40):
            busy.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            e0.record()
            for _ in range(2):
                host.copy_(devb, non_blocking=True)
            e1.record()
        for _ in range(40):
            busy.add_(1)
        e1.synchronize()
        out["d2h_GBps_gpu_busy"] = round(2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9, 2)
        torch.cuda.synchronize()
        del busy
    except RuntimeError:
        out["d2h_GBps_gpu_busy"] = None
    out["copy_MiB"] = mib
    try:
        out["host_cpus"] = len(os.sched_getaffinity(0))
    except AttributeError:
        out["host_cpus"] = os.cpu_count()
    return out


def cpu_baseline_tile(eng, wl, n):
    """CPU leg of the tile workload: the oracle's `paffy tile` on the first n records (one thread; it allocates the same per-base
    counters); the same sample runs on the GPU for the equality check."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    buf, nbytes = eng.synth(wl["seed"], wl["mean_ops"], 0, n)
    data = bytes(buf[:nbytes].cpu().numpy().tobytes())
    t0 = time.perf_counter()
    want, err = O.tile(data)
    dt = time.perf_counter() - t0
    got, _ = eng.tile(data)
    return {"value": round(n / dt, 1), "unit": "records/s", "cores": 1, "kind": "port",
            "sample": f"first {n} records of the same stream ({len(data)} B in, {len(want)} B out), {dt:.1f} s, single thread",
            "gpu_output_matches": bool(got == want and err.code == 0)}


def cpu_baseline_genomes(eng, wl, stages, n):
    """CPU leg of the add_mismatches workload: the oracle on the first n records of the SAME stream against the SAME genomes as the GPU
    leg (24 + 24 contigs of 50-250 Mb, built by the host twin of the device generator: same bytes; about 7 GB of host memory, generated
    on up to 16 threads outside the timed call); the engine that holds the device copy of those genomes runs the same sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import synth_lib

    t0 = time.perf_counter()
    host = synth_lib.Synth4(wl["seed"], wl["mean_ops"])
    data = host.records(0, n)
    seqs = host.genomes(threads=min(16, os.cpu_count() or 1))
    t_gen = time.perf_counter() - t0
    ost = [O.stage(O.ADD_MISMATCHES)]
    O.run(ost, data[: data.index(b"\n") + 1], seqs)  # builds the oracle's sequence table outside the timed call
    t0 = time.perf_counter()
    want, err = O.run(ost, data, seqs)
    dt = time.perf_counter() - t0
    got, _ = eng.run(stages, data)
    return {"value": round(n / dt, 1), "unit": "records/s", "cores": 1, "kind": "port",
            "sample": f"first {n} records of the same stream on the same 24 + 24 contigs of 50-250 Mb ({sum(len(v) for v in seqs.values())} bases built on the host in "
                      f"{t_gen:.0f} s, untimed; {len(data)} B in, {len(want)} B out), {dt:.1f} s, single thread, sequence table passed per call",
            "gpu_output_matches": bool(got == want and err.code == 0)}


if __name__ == "__main__":
    main()
