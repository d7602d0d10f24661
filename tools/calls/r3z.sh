#!/bin/bash
f=tests/golden/fuzz/tile_r3_fail.paf
python tools/dbg_tile_lib.py paffy_amd/abl/libpaffy_hip_w512.so $f 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import oracle_lib as O, paffy_amd
data=open("tests/golden/fuzz/tile_r3_fail.paf","rb").read()
e=paffy_amd.Engine()
got,info=e.to_bed(data, raise_on_error=False)
want=O.to_bed(data)
w=want[0] if isinstance(want,tuple) else want
print("to_bed equal", got==w, len(got), len(w))
PY
