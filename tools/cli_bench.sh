#!/bin/bash
# End-to-end CLI timing on the GPU box: synthetic PAF file (tools/paf_synth) -> bin/paffy <cmd> -> /dev/null
set -e -o pipefail
n=${1:-200000}
f=/tmp/cli_bench.paf
python3 - "$n" "$f" <<'PY'
import ctypes, sys, os
n=int(sys.argv[1]); path=sys.argv[2]
root=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "tests"))
import synth_lib
data = synth_lib.generate(0x5EED0003, 2048, 0, n)
open(path, "wb").write(data)
print("input bytes", len(data))
PY
for cmd in invert trim shatter "filter -u 0.9"; do
  # through a pipe into wc (what a shell pipeline does; the reader's 16 KiB reads bound it) and straight into /dev/null
  s=$(date +%s.%N)
  ./bin/paffy $cmd -i $f | wc -c > /tmp/cli_out_bytes
  e=$(date +%s.%N)
  s2=$(date +%s.%N)
  ./bin/paffy $cmd -i $f -o /dev/null
  e2=$(date +%s.%N)
  echo "$cmd: $(cat /tmp/cli_out_bytes) bytes out; | wc -c: $(python3 -c "print(f'{$e-$s:.2f} s, {$n/($e-$s):.0f} records/s')"); -o /dev/null: $(python3 -c "print(f'{$e2-$s2:.2f} s, {$n/($e2-$s2):.0f} records/s')")"
done
rm -f $f
