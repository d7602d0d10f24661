#!/bin/bash
# `bin/paffy tile` on a file beyond 2 GiB (the one-batch limit): writes the synthetic file on the GPU box, tiles it, prints sizes.
set -e
n=${1:-520000}
python3 - <<PY
import sys
sys.path.insert(0, ".")
import paffy_amd
e = paffy_amd.Engine()
with open("/tmp/big.paf", "wb") as fh:
    for b in range(0, $n, 65000):
        buf, nb = e.synth(0x5EED0005, 2048, b, min(65000, $n - b), n_contigs=2)
        fh.write(buf[:nb].cpu().numpy().tobytes())
PY
ls -l /tmp/big.paf
/usr/bin/time -v ./bin/paffy tile -i /tmp/big.paf -o /tmp/big.tiled.paf 2> /tmp/tile.time || { tail -5 /tmp/tile.time; exit 1; }
grep -E "Elapsed|Maximum resident" /tmp/tile.time
ls -l /tmp/big.tiled.paf
wc -l /tmp/big.paf /tmp/big.tiled.paf
