#!/bin/bash
# SpMV of a stencil (row-code kernel) against the number of workgroups of a launch (KSG_RC_GRID):
# with fewer workgroups than 256-row blocks a workgroup walks several blocks and stages its two
# tables once.  $1 = workload (ex2 | lap3d), $2 = grid points per side.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for G in 1048576 32768 16384 8192 4096 2048; do
  KSG_RC_GRID=$G timeout -k 10 400 python3 -u $R/bench.py --workload ${1:-ex2} --grid ${2:-4096} --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null > /tmp/line.json || exit 1
  python3 - "$G" <<'PY'
import json, sys
d = json.loads(open("/tmp/line.json").read())
m = d["kernels"]["matmult"]
print("at most %7s workgroups: %.3f ms per restart; product %.4f ms, %.0f GB/s of its own bytes" % (sys.argv[1], d["ms_per_step"], m["ms"] / m["calls"], m["GBps"]))
PY
done
