"""Oracle vs the reference's fixture: tests/paf_test.c:11-47 and the digests of SURVEY Appendix D."""
import json
import os

import oracle_lib as O
from conftest import GOLDEN
from golden.make_golden import PIPES, REFERENCE_MD5, digests

S = O.stage


def test_fixture_roundtrip(human_chimp):
    """test_paf: 207 records; read -> write -> read -> print is stable; every record passes paf_check."""
    once, err = O.run([S(O.PASS)], human_chimp)
    assert err.code == 0 and once.count(b"\n") == 207
    twice, err = O.run([S(O.PASS)], once)
    assert err.code == 0 and twice == once
    # paf_check on every record: invert's driver runs it, and invert∘invert∘invert == invert
    inv, err = O.run([S(O.INVERT)], human_chimp)
    assert err.code == 0
    assert O.run([S(O.INVERT), S(O.INVERT), S(O.INVERT)], human_chimp)[0] == inv


def test_fixture_digests(human_chimp):
    got = digests(human_chimp)
    with open(os.path.join(GOLDEN, "human_chimp_digests.json")) as fh:
        want = json.load(fh)
    assert got == want
    for name, md5 in REFERENCE_MD5.items():  # measured from the reference's own sources by the survey
        assert got[name]["md5"] == md5, name


def test_pipe_equals_chained_commands(human_chimp):
    """A fused stage list must equal running the commands one after another over text."""
    for name, stages in PIPES.items():
        text = human_chimp
        for st in stages:
            text, err = O.run([st], text)
            assert err.code == 0
        assert text == O.run(stages, human_chimp)[0], name


def test_tile_levels(human_chimp):
    out, err = O.tile(human_chimp)
    assert err.code == 0
    levels = [int(l.split(b"\ttl:i:")[1].split(b"\t")[0]) for l in out.splitlines()]
    assert len(levels) == 207 and levels.count(1) == 159 and levels.count(2) == 9 and levels.count(3) == 8 and levels.count(4) == 5
