#!/usr/bin/env python3
"""Soak test (GPU box): random records through random pipes, HIP path vs the CPU oracle, for a time budget.
usage: python tools/fuzz_gpu.py [seconds] [seed] [pipe|tile|mism|bed|chain|dedupe]   -- prints the first mismatch (and saves it under gpurun_out/) or a summary."""
import hashlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
import paffy_amd  # noqa: E402


def rand_record(rng):
    qn = "".join(rng.choice("abcXYZ.09_") for _ in range(rng.choice([1, 3, 7, 8, 12, 15, 16, 17, 30, 33, 47, 49, 70])))
    tn = "".join(rng.choice("tgcaN|-") for _ in range(rng.choice([1, 2, 9, 14, 15, 16, 31, 32, 48, 60])))
    n_ops = rng.choice([0, 1, 2, 3, 10, 63, 64, 65, 127, 128, 129, 300, 700, 2100, 5000, 9000])
    lens = rng.choice([[1], [1, 2, 3], [1, 9, 10, 99, 100], [5, 40, 400], [1, 1000, 100000], [7, 12345678]])
    alphabet = rng.choice(["MID", "MID", "MMID", "M=XID", "MI", "MD"])
    ops, q, t, prev = [], 0, 0, ""
    for _ in range(n_ops):
        op = rng.choice(alphabet)
        if rng.random() < 0.7 and op == prev:
            op = rng.choice(alphabet)
        L = rng.choice(lens)
        ops.append(f"{L}{op}")
        q += L if op != "D" else 0
        t += L if op != "I" else 0
        prev = op
    scale = rng.choice([1, 1, 1000, 10**6, 10**9])
    qlen = q + rng.randrange(1, 50) * scale + rng.choice([0, 10**4 - q % 10**4 if q else 0])
    tlen = t + rng.randrange(1, 50) * scale
    qs = rng.randrange(0, qlen - q + 1) if rng.random() < 0.8 else max(0, min(qlen - q, rng.choice([9990, 99995, 10**8 - 3, 10**9 - 50])))
    ts = rng.randrange(0, tlen - t + 1)
    if rng.random() < 0.03:  # inconsistent coordinates: paf_check failures must agree too
        ts += rng.choice([-1, 1, 5])
    tags = []
    if rng.random() < 0.7:
        tags.append(f"tp:A:{rng.choice('PSI')}")
    if rng.random() < 0.8:
        tags.append(f"AS:i:{rng.randrange(-5, 10**rng.randrange(1, 9))}")
    if rng.random() < 0.4:
        tags.append(f"NM:i:{rng.randrange(99)}")
    if rng.random() < 0.3:
        tags.append(f"tl:i:{rng.randrange(1, 5)}")
    if rng.random() < 0.3:
        tags.append(f"cn:i:{rng.randrange(10**6)}")
    if rng.random() < 0.5:
        tags.append(f"s1:i:{rng.randrange(10**7)}")
    if n_ops or rng.random() < 0.5:
        tags.append("cg:Z:" + "".join(ops))
    if rng.random() < 0.1:  # dozens of tokens: the header kernel takes 32 separators of a line per round; repeated tags (the last one wins)
        tags += [rng.choice([f"NM:i:{rng.randrange(99)}", "de:f:0.1", "zz", f"AS:i:{rng.randrange(99)}", f"cn:i:{rng.randrange(99)}", f"tl:i:{rng.randrange(1, 9)}",
                             f"s1:i:{rng.randrange(99)}", "tp:A:" + rng.choice("PSI")]) for _ in range(rng.choice([10, 21, 33, 70]))]
    rng.shuffle(tags)
    tab = "\t\t" if rng.random() < 0.05 else "\t"  # runs of tabs collapse (strtok_r)
    return (f"{qn}\t{qlen}\t{qs}\t{qs + q}\t{rng.choice('+-')}\t{tn}\t{tlen}\t{ts}\t{ts + t}\t{q}\t{max(q, t)}\t{rng.randrange(256)}"
            + "".join(tab + x for x in tags) + "\n")


def consistent_record(rng, qn, qlen, tn, tlen, n_ops, lens, alphabet="MID", strand=None, gap_runs=0):
    ops, q, t = [], 0, 0
    burst = 0  # ops other than M still to come in a row (gap_runs: bursts of up to that many)
    for k in range(n_ops):
        if burst:
            op, burst = rng.choice("ID"), burst - 1
        else:
            op = "M" if k % 2 == 0 else rng.choice(alphabet)
            if gap_runs and rng.random() < 0.05:
                burst = rng.randrange(1, gap_runs + 1)
        L = rng.choice(lens)
        if (op != "D" and q + L >= qlen - 2) or (op != "I" and t + L >= tlen - 2):
            break
        ops.append(f"{L}{op}")
        q += L if op != "D" else 0
        t += L if op != "I" else 0
    if not ops:
        ops, q, t = ["1M"], 1, 1
    qs, ts = rng.randrange(0, qlen - q), rng.randrange(0, tlen - t)
    tags = [f"AS:i:{rng.choice([5, 5, 40, 900])}"] + ([f"s1:i:{rng.choice([3, 3, 77])}"] if rng.random() < 0.6 else [])
    return f"{qn}\t{qlen}\t{qs}\t{qs + q}\t{strand or rng.choice('+-')}\t{tn}\t{tlen}\t{ts}\t{ts + t}\t{q}\t{q}\t60\t" + "\t".join(tags + ["cg:Z:" + "".join(ops)]) + "\n"


def fuzz_tile(eng, rng, budget):
    t0, rounds = time.time(), 0
    while time.time() - t0 < budget:
        contigs = [(f"ctg{i}", rng.choice([300, 5000, 70000, 1200000, 3000000])) for i in range(rng.randrange(1, 5))]
        recs = []
        for _ in range(rng.choice([1, 10, 200, 1500])):
            qn, qlen = rng.choice(contigs)
            recs.append(consistent_record(rng, qn, qlen, "t", 4000000, rng.choice([1, 5, 60, 900]), rng.choice([[1, 3], [5, 50, 300], [1000, 20000]])))
        data = "".join(recs).encode()
        want, werr = O.tile(data)
        got, info = eng.tile(data, raise_on_error=False)
        if got != want or info.error.code != werr.code:
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_tile_fail.paf"), "wb") as fh:
                fh.write(data)
            print("TILE MISMATCH", info.error.code, werr.code, len(got), len(want))
            sys.exit(1)
        rounds += 1
    print(f"tile fuzz ok: {rounds} rounds")


def fuzz_bed(eng, rng, budget):
    """to_bed with random options; -n walks the target side too (both strands, sequences in both roles)."""
    t0, rounds = time.time(), 0
    while time.time() - t0 < budget:
        contigs = [(f"ctg{i}", rng.choice([300, 5000, 70000, 1200000, 2500000])) for i in range(rng.randrange(1, 5))]
        recs = []
        for _ in range(rng.choice([1, 10, 200, 1200])):
            qn, qlen = rng.choice(contigs)
            tn, tlen = rng.choice(contigs) if rng.random() < 0.3 else ("t", 4000000)
            recs.append(consistent_record(rng, qn, qlen, tn, tlen, rng.choice([1, 5, 60, 900]), rng.choice([[1, 3], [5, 50, 300], [1000, 20000]])))
        if rng.random() < 0.1:  # a record that breaks an assert somewhere in the middle
            f = recs[len(recs) // 2].split("\t")
            f[3] = str(int(f[3]) + 1)
            recs[len(recs) // 2] = "\t".join(f)
        data = "".join(recs).encode()
        kw = dict(binary=rng.random() < 0.3, exclude_unaligned=rng.random() < 0.3, exclude_aligned=rng.random() < 0.2, min_size=rng.choice([1, 1, 2, 40, 5000]),
                  include_inverted=rng.random() < 0.5)
        want, werr = O.to_bed(data, **kw)
        got, info = eng.to_bed(data, raise_on_error=False, **kw)
        if got != want or info.error.code != werr.code or (werr.code and info.error.record != werr.record):
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_bed_fail.paf"), "wb") as fh:
                fh.write(data)
            print("BED MISMATCH", kw, info.error.code, werr.code, info.error.record, werr.record, len(got), len(want))
            sys.exit(1)
        rounds += 1
    print(f"to_bed fuzz ok: {rounds} rounds")


def fuzz_mismatches(eng, rng, budget):
    t0, rounds = time.time(), 0
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    while time.time() - t0 < budget:
        seqs = {}
        for i in range(rng.randrange(1, 4)):
            L = rng.choice([50, 2000, 60000, 250000])
            tseq = "".join(rng.choice("ACGTacgtN" if rng.random() < 0.2 else "ACGT") for _ in range(L))
            seqs[f"t{i}"] = tseq
            q = list(tseq)
            for _ in range(L // 20):
                q[rng.randrange(L)] = rng.choice("ACGT")
            seqs[f"q{i}"] = "".join(q)
        recs = []
        for _ in range(rng.choice([1, 20, 120])):
            i = rng.randrange(len(seqs) // 2)
            L = len(seqs[f"t{i}"])
            recs.append(consistent_record(rng, f"q{i}", L, f"t{i}", L, rng.choice([1, 9, 200, 200, 3000, 9000, 20000]), rng.choice([[1, 2], [7, 30], [100, 900], [1, 17, 40]]),
                                          alphabet=rng.choice(["MID", "ID", "MMMID"]), gap_runs=rng.choice([0, 0, 3, 70, 200])))
        data = "".join(recs).encode()
        want, werr = O.run([O.stage(O.ADD_MISMATCHES)], data, seqs)
        eng.set_sequences(seqs)
        got, info = eng.run([paffy_amd.stage(paffy_amd.ADD_MISMATCHES)], data, raise_on_error=False)
        ok = got == want and info.error.code == werr.code
        if ok and not werr.code:
            want2 = O.run([O.stage(O.ADD_MISMATCHES), O.stage(O.REMOVE_MISMATCHES), O.stage(O.SHATTER)], data, seqs)[0]
            got2 = eng.run([paffy_amd.stage(paffy_amd.ADD_MISMATCHES), paffy_amd.stage(paffy_amd.REMOVE_MISMATCHES), paffy_amd.stage(paffy_amd.SHATTER)], data,
                           raise_on_error=False)[0]
            ok = got2 == want2
            for pipe in ([O.INVERT, O.ADD_MISMATCHES], [O.ADD_MISMATCHES, O.TRIM_IDENTITY], [O.TRIM_FIXED, O.ADD_MISMATCHES, O.INVERT]):
                if not ok:
                    break
                w3, e3 = O.run([O.stage(k, 0.2, 0.5) for k in pipe], data, seqs)
                g3, i3 = eng.run([paffy_amd.stage(k, 0.2, 0.5) for k in pipe], data, raise_on_error=False)
                ok = g3 == w3 and i3.error.code == e3.code
        if not ok:
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_mism_fail.paf"), "wb") as fh:
                fh.write(data)
            print("MISMATCHES MISMATCH", info.error.code, werr.code, len(got), len(want))
            sys.exit(1)
        rounds += 1
    print(f"add_mismatches fuzz ok: {rounds} rounds")


def fuzz_chain(eng, rng, budget):
    """paffy chain on collinear runs over a few (query, target, strand) groups, random options; inputs where the reference's fresh
    iterator admits a candidate (an address tie, see oracle/paf_oracle.c) are compared with the oracle run WITHOUT that walk: the GPU's
    documented behaviour there."""
    from test_gpu_chain import collinear_set

    t0, rounds, skipped = time.time(), 0, 0
    while time.time() - t0 < budget:
        n = rng.choice([1, 3, 30, 300, 2500, 12000])
        data = collinear_set(rng, n, n_q=rng.choice([1, 2, 6]), n_t=rng.choice([1, 3]), span=rng.choice([50_000, 2_000_000]), exact=rng.choice([0.0, 0.0, 0.3]),
                             score_hi=rng.choice([30, 20000]))
        kw = dict(gap_open=rng.choice([0, 50, 5000]), gap_extend=rng.choice([0, 1, 3]), max_gap=rng.choice([500, 20000, 1000000]), trim=rng.choice([0.0, 0.3, 1.0]))
        want, werr, fresh = O.chain(data, **kw)
        if fresh:
            skipped += 1
            want, werr, _ = O.chain(data, fresh_walk=False, **kw)
        got, info = eng.chain(data, raise_on_error=False, batch_bytes=rng.choice([None, 50_000]), **kw)
        if got != want or info.error.code != werr.code:
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_chain_fail.paf"), "wb") as fh:
                fh.write(data)
            print("CHAIN MISMATCH", kw, info.error.code, werr.code, len(got), len(want))
            sys.exit(1)
        rounds += 1
    print(f"chain fuzz ok: {rounds} rounds ({skipped} of them inputs with an address tie, compared with the oracle without the fresh-iterator walk)")


def fuzz_dedupe(eng, rng, budget):
    """paffy dedupe [-a]: streams drawn from a small pool of records (repeats, swapped twins, near misses, records that fail paf_check
    or do not parse), whole and cut into random batches through one context, against the oracle's loop."""
    t0, rounds = time.time(), 0
    while time.time() - t0 < budget:
        pool = []
        for _ in range(rng.choice([3, 20, 120])):
            qn, tn = "q%d" % rng.randrange(rng.choice([1, 5, 40])), "t%d" % rng.randrange(rng.choice([1, 5, 40]))
            qs, ts, ln = rng.randrange(0, 900), rng.randrange(0, 1900), rng.randrange(1, 90)
            pool.append("\t".join([qn, "1000", str(qs), str(qs + ln), rng.choice("+-"), tn, "2000", str(ts), str(ts + ln), str(ln), str(ln), "60",
                                   rng.choice(["cg:Z:%dM" % ln, "AS:i:7\tcg:Z:%dM" % ln, "tp:A:P\tcg:Z:3S"])]) + "\n")
        bad = ["qb%d\t50\t10\t4\t+\ttb\t200\t0\t3\t3\t3\t60\tcg:Z:3M\n" % k for k in range(3)] + ["qz\t50\t1\n"]
        n = rng.choice([1, 2, 50, 400, 3000, 20000])
        p_bad = rng.choice([0.0, 0.0, 0.001, 0.02])
        lines = []
        for _ in range(n):
            l = rng.choice(bad[:3] if rng.random() < 0.9 else bad) if rng.random() < p_bad else rng.choice(pool)
            f = l.rstrip("\n").split("\t")
            if len(f) > 8 and rng.random() < 0.35:
                f[0], f[5] = f[5], f[0]
                f[1], f[6] = f[6], f[1]
                f[2], f[7] = f[7], f[2]
                f[3], f[8] = f[8], f[3]
            elif len(f) > 8 and rng.random() < 0.1:
                f[3] = str(int(f[3]) + 1)
            lines.append(("\t".join(f) + "\n").encode())
        data = b"".join(lines)
        for inv in (False, True):
            want, werr = O.dedupe(data, inv)
            got, info = eng.dedupe(data, inv, raise_on_error=False)
            ok = info.error.code == werr.code and got == want and (not werr.code or info.error.record == werr.record)
            cuts = sorted(rng.sample(range(1, n), min(n - 1, rng.randrange(0, 6)))) if n > 1 else []
            outs, base, code, rec = [], 0, 0, 0
            for i, (a, b) in enumerate(zip([0] + cuts, cuts + [n])):
                o, inf = eng.dedupe(b"".join(lines[a:b]), inv, reset=(i == 0), raise_on_error=False)
                outs.append(o)
                if inf.error.code:
                    code, rec = inf.error.code, a + inf.error.record
                    break
            ok = ok and b"".join(outs) == want and code == werr.code and (not code or rec == werr.record)
            if not ok:
                with open(os.path.join(ROOT, "gpurun_out", "fuzz_dedupe_fail.paf"), "wb") as fh:
                    fh.write(data)
                print("DEDUPE MISMATCH", inv, cuts, (info.error.code, info.error.record), (werr.code, werr.record), len(got), len(want))
                sys.exit(1)
        rounds += 1
    print(f"dedupe fuzz ok: {rounds} rounds")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mode = sys.argv[3] if len(sys.argv) > 3 else "pipe"
    rng = random.Random(seed)
    eng = paffy_amd.Engine()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    if mode == "tile":
        return fuzz_tile(eng, rng, budget)
    if mode == "bed":
        return fuzz_bed(eng, rng, budget)
    if mode == "mism":
        return fuzz_mismatches(eng, rng, budget)
    if mode == "chain":
        return fuzz_chain(eng, rng, budget)
    if mode == "dedupe":
        return fuzz_dedupe(eng, rng, budget)
    kinds = [O.INVERT, O.TRIM_IDENTITY, O.TRIM_FIXED, O.REMOVE_MISMATCHES, O.PASS, O.FILTER]
    t0, rounds, nbytes = time.time(), 0, 0
    while time.time() - t0 < budget:
        data = "".join(rand_record(rng) for _ in range(rng.choice([1, 5, 40, 200]))).encode()
        pipe = [rng.choice(kinds) for _ in range(rng.randrange(0, 4))]
        if rng.random() < 0.7:
            pipe.append(O.SHATTER)
        params = (rng.choice([0.05, 0.2, 0.9]), rng.choice([1.0, 0.5, 0.1]))
        f = dict(min_identity=rng.choice([-1.0, 0.5, 0.9]), min_alignment_score=rng.choice([-1, 100, 10**5]), invert=rng.random() < 0.3)
        O.set_filter(**f)
        eng.set_filter(**f)
        gpu_pipe = [paffy_amd.stage(k, *params) for k in pipe]
        cpu_pipe = list(pipe)
        stats_at = None
        if rng.random() < 0.3:
            # a stats stage (paffy view -s) somewhere in front of the shatter: a process of its own in a shell pipe, so what it passes on is
            # what `paf_write | paf_parse` makes of a record (an emptied cigar loses its tag) -- the oracle's PASS stage -- and the six sums are checked
            stats_at = rng.randrange(len(pipe) + (0 if pipe and pipe[-1] == O.SHATTER else 1))
            gpu_pipe.insert(stats_at, paffy_amd.stage(paffy_amd.STATS))
            cpu_pipe.insert(stats_at, O.PASS)
        want, werr = O.run([O.stage(k, *params) for k in cpu_pipe], data)
        got, info = eng.run(gpu_pipe, data, raise_on_error=False)
        if stats_at is not None and werr.code == 0 and info.error.code == 0:
            upto, uerr = O.run([O.stage(k, *params) for k in pipe[:stats_at]] or [O.stage(O.PASS)], data)
            acc = [0] * 6
            for ln in upto.splitlines():
                cg = ln.split(b"cg:Z:")
                if len(cg) > 1:
                    acc = O.cigar_stats(cg[1].split(b"\t")[0].decode(), acc, zero=False)
            if uerr.code == 0 and list(eng.plan_stats()) != acc:
                print("STATS MISMATCH", dict(pipe=pipe, stats_at=stats_at, params=params, filter=f, gpu=list(eng.plan_stats()), cpu=acc))
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "fuzz_fail.paf"), "wb") as fh:
                    fh.write(data)
                sys.exit(1)
        if info.error.code != werr.code or (werr.code and info.error.record != werr.record) or got != want:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_fail.paf"), "wb") as fh:
                fh.write(data)
            print("MISMATCH", dict(pipe=pipe, params=params, filter=f, gpu=(info.error.code, info.error.record, len(got)), cpu=(werr.code, werr.record, len(want)),
                                   sha_gpu=hashlib.sha256(got).hexdigest()[:12], sha_cpu=hashlib.sha256(want).hexdigest()[:12]))
            sys.exit(1)
        rounds += 1
        nbytes += len(want)
    print(f"fuzz ok: {rounds} rounds, {nbytes / 1e6:.1f} MB of output compared, seed {seed}")


if __name__ == "__main__":
    main()
