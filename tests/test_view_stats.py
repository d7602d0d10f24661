"""`paffy view` (impl/paf_view.c:42-213; aggregate line, per-record lines, -a rows): PAFFY_STATS sums on the GPU against the oracle's
paf_stats_calc over the same records, and the CLI line with the reference's float arithmetic and format."""
import os
import struct
import subprocess

import pytest

import oracle_lib as O
import synth_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAFFY = os.path.join(ROOT, "bin", "paffy")


def oracle_sums(lines):
    acc = [0] * 6
    for ln in lines.splitlines():
        f = ln.split(b"cg:Z:")
        if len(f) > 1:
            acc = O.cigar_stats(f[1].split(b"\t")[0].decode(), acc, zero=False)
    return acc


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def test_stats_stage_and_cli(tmp_path, human_chimp):
    import paffy_amd

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    eng = paffy_amd.Engine()
    # the stage alone: the fixture's cigars as they are
    info = eng.plan([paffy_amd.stage(paffy_amd.STATS)], eng.to_device(human_chimp), len(human_chimp))
    assert info.error.code == 0 and list(eng.plan_stats()) == oracle_sums(human_chimp)
    # behind add_mismatches on records that lie on homologous bases (both strands, long and short records)
    host = synth_lib.Synth4(0x5EED0004, 2048, n_contigs=6, tlen_min=200_000, tlen_span=200_000)
    data, seqs = host.records(0, 1500), host.genomes()
    eng.set_sequences(seqs)
    info = eng.plan([paffy_amd.stage(paffy_amd.ADD_MISMATCHES), paffy_amd.stage(paffy_amd.STATS)], eng.to_device(data), len(data))
    want = oracle_sums(O.run([O.stage(O.ADD_MISMATCHES)], data, seqs)[0])
    assert info.error.code == 0 and list(eng.plan_stats()) == want
    eng.close()
    # the CLI: same sums, the reference's line (float32 quotients printed with %f)
    paf, fa = tmp_path / "in.paf", tmp_path / "g.fa"
    paf.write_bytes(data)
    with open(fa, "wb") as fh:
        for name, s in seqs.items():
            fh.write(b">" + name.encode() + b"\n" + s + b"\n")
    m, x, qi, qd, qib, qdb = want
    line = ("Total-alignments:%d\tAvg-Identity:%f\tAvg-Identity-with-gaps:%f\tAligned-bases:%d\tAligned-bases-with-gaps:%d\tQuery-inserts:%d\tQuery-deletes:%d\n"
            % (1500, f32(f32(m) / f32(m + x)), f32(f32(m) / f32(m + x + qib + qdb)), m + x, m + x + qib + qdb, qi, qd))
    for env in ({}, {"PAFFY_CHUNK_MB": "1"}):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([PAFFY, "view", "-s", "-t", "-u", "0.9", "-i", str(paf), str(fa)], capture_output=True, env=e)
        assert r.returncode == 0, r.stderr[-800:]
        assert r.stdout.decode() == line
    # identity below the requested minimum: assert -> SIGABRT; no sequence file: exit 1; without -t: outside this build
    r = subprocess.run([PAFFY, "view", "-s", "-t", "-u", "0.995", "-i", str(paf), str(fa)], capture_output=True)
    assert r.returncode == -6 and r.stdout.decode() == line
    assert subprocess.run([PAFFY, "view", "-s", "-t", "-i", str(paf)], capture_output=True).returncode == 1
    # without -t: paf_pretty_print's stats line for every record (impl/paf.c:269-281), then the aggregate
    r = subprocess.run([PAFFY, "view", "-s", "-i", str(paf), str(fa)], capture_output=True, env=dict(os.environ, PAFFY_CHUNK_MB="1"))
    assert r.returncode == 0, r.stderr[-500:]
    got = r.stdout.decode().splitlines(keepends=True)
    assert len(got) == 1501 and got[-1] == line
    enc = O.run([O.stage(O.ADD_MISMATCHES)], data, seqs)[0].splitlines()
    for k in (0, 1, 700, 1499):
        f = enc[k].split(b"\t")
        m1, x1, i1, d1, ib1, db1 = O.cigar_stats(enc[k].split(b"cg:Z:")[1].decode())
        want_line = ("Query:%s\tQ-start:%d\tQ-length:%d\tTarget:%s\tT-start:%d\tT-length:%d\tSame-strand:%d\tScore:%d\tIdentity:%f\tIdentity-with-gaps%f"
                     "\tAligned-bases:%d\tQuery-inserts:%d\tQuery-deletes:%d\n"
                     % (f[0].decode(), int(f[2]), int(f[3]) - int(f[2]), f[5].decode(), int(f[7]), int(f[8]) - int(f[7]), 1 if f[4] == b"+" else 0,
                        int([t for t in f if t.startswith(b"AS:i:")][0][5:]), f32(f32(m1) / f32(m1 + x1)), f32(f32(m1) / f32(m1 + x1 + ib1 + db1)), m1 + x1, i1, d1))
        assert got[k] == want_line, k
    # -a: the base-level rows under every stats line (impl/paf.c:283-315), written by the GPU; whole output against the oracle's
    # paf_pretty_print of the encoded records, in one batch and in many
    by_name = {k: v for k, v in seqs.items()}
    want_all = b"".join(O.pretty_print(ln, by_name[ln.split(b"\t")[0].decode()], by_name[ln.split(b"\t")[5].decode()])[1] for ln in enc)
    for env in ({}, {"PAFFY_CHUNK_MB": "1"}):
        r = subprocess.run([PAFFY, "view", "-a", "-s", "-i", str(paf), str(fa)], capture_output=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-500:]
        assert r.stdout == want_all + line.encode()
    # without -s the totals stay zero and the reference's closing assert (NaN >= 0) ends it with SIGABRT after everything is printed
    r = subprocess.run([PAFFY, "view", "-a", "-i", str(paf), str(fa)], capture_output=True)
    assert r.returncode == -6 and r.stdout == want_all
    # lower-case bases are shown as they are in the file and still compare equal
    low = {k: (v.lower() if i % 2 else v) for i, (k, v) in enumerate(seqs.items())}
    with open(fa, "wb") as fh:
        for name, s in low.items():
            fh.write(b">" + name.encode() + b"\n" + s + b"\n")
    want_low = b"".join(O.pretty_print(ln, low[ln.split(b"\t")[0].decode()], low[ln.split(b"\t")[5].decode()])[1] for ln in enc[:200])
    small = tmp_path / "small.paf"
    small.write_bytes(b"".join(data.splitlines(keepends=True)[:200]))
    r = subprocess.run([PAFFY, "view", "-a", "-s", "-i", str(small), str(fa)], capture_output=True)
    assert r.returncode == 0 and r.stdout.startswith(want_low) and r.stdout[len(want_low):].startswith(b"Total-alignments:200\t")
    # -a -t prints nothing per record (paf_pretty_print is not called, impl/paf_view.c:170-172)
    r = subprocess.run([PAFFY, "view", "-a", "-t", "-s", "-i", str(small), str(fa)], capture_output=True)
    assert r.returncode == 0 and r.stdout.startswith(b"Total-alignments:200\t") and r.stdout.count(b"\n") == 1
