"""The command lines of the reference's tests/paf_tools_test.sh, run as real shell pipes of `bin/paffy` processes on a synthetic
alignment set (records on homologous bases of two generated genomes stand in for the wget + lastz steps, which need the network;
the pipeline script too). Exit statuses as the script expects them; the aggregate lines against the oracle."""
import os
import re
import subprocess

import pytest

import oracle_lib as O
import synth_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sh(cmd, cwd, ok=True):
    env = dict(os.environ, PATH=os.path.join(ROOT, "bin") + os.pathsep + os.environ["PATH"])
    r = subprocess.run(["bash", "-o", "pipefail", "-c", cmd], cwd=cwd, env=env, capture_output=True, text=True, timeout=300)
    assert (r.returncode == 0) == ok, (cmd, r.returncode, r.stderr[-600:])
    return r.stdout


def aligned(line):
    return int(re.search(r"Aligned-bases:(\d+)", line).group(1))


def test_paf_tools_script(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    host = synth_lib.Synth4(0x5EED0004, 512, n_contigs=4, tlen_min=400_000, tlen_span=300_000)
    data, seqs = host.records(0, 1200), host.genomes()
    (tmp_path / "output.paf").write_bytes(data)
    for name, s in seqs.items():
        (tmp_path / (name + ".fa")).write_bytes(b">" + name.encode() + b"\n" + s + b"\n")
    d = str(tmp_path)
    total = sum(int(l.split(b"\t")[9]) for l in data.splitlines())  # column 10 = aligned bases of these records
    view = f"paffy view *.fa -s -t -u 0.74 -v {total - 10}"
    base = sh(f"paffy view -i output.paf *.fa -s -t -u 0.74 -v {total - 10}", d)
    assert base.startswith("Total-alignments:1200\t") and aligned(base) == total
    for cmd in ("paffy invert -i output.paf", "paffy shatter -i output.paf", "paffy tile -i output.paf", "paffy add_mismatches -i output.paf *.fa",
                "paffy add_mismatches -i output.paf *.fa | paffy add_mismatches -a", "paffy trim -r 0.95 -i output.paf"):
        out = sh(f"{cmd} | {view}", d)
        assert aligned(out) == total, cmd
    shat = sh(f"paffy shatter -i output.paf | {view}", d).split("\t")  # gapless blocks: same matches, no indels left
    assert shat[1] == base.split("\t")[1] and shat[3] == base.split("\t")[3] and shat[5:] == ["Query-inserts:0", "Query-deletes:0\n"]
    out = sh(f"paffy add_mismatches -i output.paf *.fa | paffy trim -r 0.05 | paffy view *.fa -s -t -u 0.74 -v {int(total * 0.8)}", d)
    assert int(total * 0.8) <= aligned(out) <= total
    kept = sh(f"paffy filter -i output.paf -t 5000000 | paffy view *.fa -s -t -u 0.74 -v 1", d)
    rest = sh(f"paffy filter -i output.paf -t 5000000 -x | paffy view *.fa -s -t -u 0.74 -v 1", d)
    assert aligned(kept) + aligned(rest) == total and 0 < aligned(kept) < total
    sh(f"paffy view -i output.paf *.fa -s -t -u 0.999 -v 1", d, ok=False)  # the identity assert of view fails the pipe
    # to_bed as the script checks it
    sh("paffy to_bed -i output.paf -o output.bed && [ -s output.bed ]", d)
    sh("paffy to_bed -i output.paf -b -o output_binary.bed && [ -s output_binary.bed ] && awk '{if ($4 > 1) exit 1}' output_binary.bed", d)
    sh("paffy to_bed -i output.paf -e -o no_unaligned.bed && [ -s no_unaligned.bed ] && awk '{if ($4 == 0) exit 1}' no_unaligned.bed", d)
    sh("paffy to_bed -i output.paf -f -o unaligned_only.bed && awk '{if ($4 != 0) exit 1}' unaligned_only.bed", d)
    sh('[ "$(paffy to_bed -i output.paf -e -n | wc -l)" -ge "$(paffy to_bed -i output.paf -e | wc -l)" ]', d)
    assert (tmp_path / "output.bed").read_bytes() == O.to_bed(data)[0]
    # filters behind add_mismatches / tile, fixed trim, dedupe -a
    sh("paffy add_mismatches -i output.paf *.fa | paffy filter -u 0.7 > /dev/null", d)
    sh("paffy add_mismatches -i output.paf *.fa | paffy filter -v 0.7 > /dev/null", d)
    sh("paffy tile -i output.paf | paffy filter -w 1 > /dev/null", d)
    sh(f"paffy trim -f -t 0.1 -i output.paf | paffy view *.fa -s -t -u 0.73 -v {int(total * 0.8)}", d)
    sh("paffy invert -i output.paf > output_inv.paf", d)
    out = sh(f"cat output.paf output_inv.paf | paffy dedupe -a | {view}", d)
    assert out.split("\t")[0] == "Total-alignments:1200" and aligned(out) == total


def test_paf_pipeline_script(tmp_path):
    """The reference's tests/paf_pipeline_test.sh:32-94 (the Cactus-style pipeline) as real shell pipes, on the synthetic set instead of
    the wget + lastz steps: invert, cat, split_file -q, per contig add_mismatches | chain | tile | trim, cat, filter -w 1, chain,
    filter -s, view. Every intermediate file against the oracle run through the same steps."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    host = synth_lib.Synth4(0x5EED0004, 512, n_contigs=4, tlen_min=400_000, tlen_span=300_000)
    data, seqs = host.records(0, 1500), host.genomes()
    (tmp_path / "lastz.paf").write_bytes(data)
    for name, s in seqs.items():
        (tmp_path / (name + ".fa")).write_bytes(b">" + name.encode() + b"\n" + s + b"\n")
    d = str(tmp_path)
    sh("paffy invert -i lastz.paf > inverted.paf && cat lastz.paf inverted.paf > combined.paf && mkdir -p split && paffy split_file -q -i combined.paf -p split/", d)
    sh("for f in split/*.paf; do b=$(basename $f .paf); mkdir -p parallel_$b; "
       "paffy add_mismatches -i $f *.fa | paffy chain | paffy tile | paffy trim > parallel_$b/trimmed.paf || exit 1; done; cat parallel_*/trimmed.paf > trimmed.paf", d)
    sh("paffy view -i trimmed.paf *.fa -s -t && paffy filter -i trimmed.paf -w 1 > primary.paf && paffy chain -i primary.paf > primary_chained.paf && "
       "paffy filter -i primary_chained.paf -s 20000 > primary_final.paf", d)
    # the same steps on the oracle
    inv = O.run([O.stage(O.INVERT)], data)[0]
    combined = data + inv
    assert (tmp_path / "combined.paf").read_bytes() == combined
    by_query = {}
    for ln in combined.splitlines(keepends=True):
        by_query.setdefault(ln.split(b"\t")[0], []).append(ln)
    names = sorted(p.name for p in (tmp_path / "split").iterdir())
    assert names == sorted(q.decode() + ".paf" for q in by_query)
    trimmed = b""
    for nm in sorted("parallel_" + n[:-4] for n in names):  # the order `cat parallel_*/trimmed.paf` expands to
        part = b"".join(by_query[nm[len("parallel_"):].encode()])
        enc = O.run([O.stage(O.ADD_MISMATCHES)], part, seqs)[0]
        chained, err, _ = O.chain(enc)
        assert err.code == 0
        tiled = O.tile(chained)[0]
        want = O.run([O.stage(O.TRIM_IDENTITY, 0.05, 1.0)], tiled)[0]
        assert (tmp_path / nm / "trimmed.paf").read_bytes() == want, nm
        trimmed += want
    assert (tmp_path / "trimmed.paf").read_bytes() == trimmed
    primary = O.filter(trimmed, max_tile_level=1)[0]
    assert (tmp_path / "primary.paf").read_bytes() == primary and 0 < primary.count(b"\n") < trimmed.count(b"\n")
    rechained, err, _ = O.chain(primary)
    assert err.code == 0 and (tmp_path / "primary_chained.paf").read_bytes() == rechained
    final = O.filter(rechained, min_chain_score=20000)[0]
    assert (tmp_path / "primary_final.paf").read_bytes() == final and final.count(b"\n") > 0
    out = sh("paffy view -i primary_final.paf *.fa -s -t -u 0.74 -v 1000", d)
    assert out.startswith("Total-alignments:%d\t" % final.count(b"\n"))
