#!/bin/bash
# SpMV of the 512^3 Laplacian (row-code kernel) against the number of consecutive 256-row blocks
# that share an XCD (KSG_SPMV_ROWCODE_CHUNK; 0 = the choice made at creation from the offsets).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for C in 0 1 4 16 32; do
  KSG_SPMV_ROWCODE_CHUNK=$C timeout -k 10 400 python3 -u $R/bench.py --workload lap3d --grid ${1:-512} --steps 2 --warmup 1 --no-cpu-baseline 2> /dev/null > /tmp/line.json || exit 1
  python3 - "$C" <<'PY'
import json, sys
d = json.loads(open("/tmp/line.json").read())
m = d["kernels"]["matmult"]
print("chunk %2s: %.3f ms per restart, %.1f GB/s; product %.4f ms; %s" % (sys.argv[1], d["ms_per_step"], d["value"], m["ms"] / m["calls"],
      d["config"]["spmv_variant"].split(";")[0][-60:]))
PY
done
