"""paffy filter (impl/paf_filter.c:120-156): oracle known answers on CPU, HIP path vs oracle on the GPU.

The reference has no golden outputs for filter (its shell tests only run it), so the known answers below are worked
out by hand from the reference's arithmetic: identity = (float)matches / (matches + mismatches) with X as the only
mismatch, identity_with_gaps adds the I and D bases to the denominator, 0/0 is NaN and fails every threshold.
"""
import hashlib
import random

import pytest

import oracle_lib as O
import synth_lib

L1 = b"q1\t100\t0\t10\t+\tt1\t100\t0\t10\t8\t10\t60\tAS:i:50\ts1:i:700\ttl:i:1\tcg:Z:8=2X\n"        # identity 0.8, gaps 0.8
L2 = b"q2\t100\t0\t10\t-\tt1\t100\t5\t16\t9\t12\t60\tAS:i:5\tcg:Z:5M2D4M1I\n"                      # identity 1.0, gaps 9/12
L3 = b"q3\t100\t0\t10\t+\tt1\t100\t0\t10\t10\t10\t60\tAS:i:90\ts1:i:10\ttl:i:3\n"                   # no cigar: NaN identities
L4 = b"q4\t100\t0\t4\t+\tt1\t100\t0\t4\t0\t4\t60\ttp:A:S\tAS:i:20\ts1:i:2000\ttl:i:2\tcg:Z:4X\n"     # identity 0, gaps 0
ALL = L1 + L2 + L3 + L4


def norm(line):
    return O.run([O.stage(O.PASS)], line)[0]


def test_oracle_known_answers():
    n = {i: norm(l) for i, l in enumerate((L1, L2, L3, L4), 1)}
    out, err = O.filter(ALL)  # defaults: everything with defined identities passes, the cigar-less record does not
    assert err.code == 0 and out == n[1] + n[2] + n[4]
    assert O.filter(ALL, invert=True)[0] == n[3]
    assert O.filter(ALL, min_identity=0.8)[0] == n[1] + n[2]           # (float)8/10 widened is above the double 0.8
    assert O.filter(ALL, min_identity=0.81)[0] == n[2]
    assert O.filter(ALL, min_identity_with_gaps=0.76)[0] == n[1]       # 9/12 = 0.75
    assert O.filter(ALL, min_alignment_score=20)[0] == n[1] + n[4]
    assert O.filter(ALL, min_chain_score=700)[0] == n[1] + n[4]        # absent s1 is -1
    assert O.filter(ALL, max_tile_level=1)[0] == n[1] + n[2]           # absent tl is -1
    assert O.filter(ALL, max_tile_level=1, invert=True)[0] == n[3] + n[4]
    O.set_filter()


def test_oracle_filter_in_a_pipe(human_chimp):
    """A dropped record reaches no later stage; kept ones go through unchanged."""
    O.set_filter(min_alignment_score=6000)
    kept, _ = O.run([O.stage(O.FILTER)], human_chimp)
    both, _ = O.run([O.stage(O.FILTER), O.stage(O.SHATTER)], human_chimp)
    assert both == O.run([O.stage(O.SHATTER)], kept)[0]
    assert 0 < kept.count(b"\n") < 207
    O.set_filter()


FILTERS = [dict(), dict(invert=True), dict(min_identity=0.9), dict(min_identity_with_gaps=0.85), dict(min_alignment_score=5000),
           dict(min_chain_score=3000000, invert=True), dict(max_tile_level=1), dict(min_identity=0.95, min_alignment_score=2000, max_tile_level=5)]


@pytest.mark.gpu
def test_gpu_filter_matches_oracle(human_chimp):
    import paffy_amd

    eng = paffy_amd.Engine()
    rng = random.Random(3)
    synth = synth_lib.generate(0x5EED0008, 300, 0, 1500)
    tiled = O.tile(human_chimp)[0]  # records with tl / tp tags
    for data in (ALL, human_chimp, tiled, synth):
        for f in FILTERS:
            O.set_filter(**f)
            eng.set_filter(**f)
            for pipe in ([O.FILTER], [O.INVERT, O.FILTER], [O.FILTER, O.TRIM_IDENTITY], [O.FILTER, O.SHATTER], [O.INVERT, O.FILTER, O.SHATTER]):
                if data is tiled and O.SHATTER in pipe and rng.random() < 0.5:
                    continue
                want, werr = O.run([O.stage(k) for k in pipe], data)
                got, info = eng.run([paffy_amd.stage(k) for k in pipe], data, raise_on_error=False)
                assert info.error.code == werr.code
                assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest(), (f, pipe, len(got), len(want))
    O.set_filter()
    eng.close()
