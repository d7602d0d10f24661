"""Regenerates tests/golden/human_chimp_digests.json from the CPU oracle.

The digests freeze the oracle's outputs on the reference's fixture (tests/human_chimp.paf,
copied here as data) so that later edits to the oracle or the HIP path are caught. The
md5 values recorded in REFERENCE_MD5 were measured by the survey from the reference's own
unmodified sources; the oracle has to reproduce them.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

S = O.stage
PIPES = {
    "shatter": [S(O.SHATTER)],
    "shatter|invert": [S(O.SHATTER), S(O.INVERT)],
    "invert": [S(O.INVERT)],
    "invert|invert": [S(O.INVERT), S(O.INVERT)],
    "trim": [S(O.TRIM_IDENTITY)],
    "trim -r 0.95": [S(O.TRIM_IDENTITY, 0.95, 1.0)],
    "trim -f -t 0.1": [S(O.TRIM_FIXED, 0.05, 0.1)],
    "add_mismatches -a": [S(O.REMOVE_MISMATCHES)],
    "invert|trim|shatter": [S(O.INVERT), S(O.TRIM_IDENTITY), S(O.SHATTER)],
}
REFERENCE_MD5 = {
    "shatter": "a40a2d093b0ece866c368e26188b1532",
    "shatter|invert": "4791ecad5ed6ca7db98dbf65dad25525",
    "invert": "66516859b85d07bce607635a989e1203",
    "invert|invert": "125147d719d7e0657f043c330c5c0b9d",
    "trim -r 0.95": "125147d719d7e0657f043c330c5c0b9d",
    "add_mismatches -a": "125147d719d7e0657f043c330c5c0b9d",
    "trim": "a06993ae97f20aea2e9235d59a3d29c8",
    "trim -f -t 0.1": "e66b60d706e85db2850e93aa86a4dcae",
    "invert|trim|shatter": "e672bae99765b1697c064ef2f47dae5c",
    "tile": "cd1f7c0ed4280d1e934026f686d67bd2",
}


def digests(data):
    res = {}
    for name, stages in PIPES.items():
        out, err = O.run(stages, data)
        assert err.code == 0, (name, err.code)
        res[name] = {"lines": out.count(b"\n"), "bytes": len(out), "md5": hashlib.md5(out).hexdigest(),
                     "sha256": hashlib.sha256(out).hexdigest()}
    out, err = O.tile(data)
    assert err.code == 0
    res["tile"] = {"lines": out.count(b"\n"), "bytes": len(out), "md5": hashlib.md5(out).hexdigest(),
                   "sha256": hashlib.sha256(out).hexdigest()}
    return res


if __name__ == "__main__":
    with open(os.path.join(HERE, "human_chimp.paf"), "rb") as fh:
        data = fh.read()
    res = digests(data)
    for k, v in REFERENCE_MD5.items():
        assert res[k]["md5"] == v, k
    with open(os.path.join(HERE, "human_chimp_digests.json"), "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print("wrote", len(res), "digests")
