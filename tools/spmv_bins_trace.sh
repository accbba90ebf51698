#!/bin/bash
# Per-kernel times of the binned product on an R-MAT graph Laplacian: tools/spmv_bins.py under
# rocprofv3 --kernel-trace.  Usage: tools/spmv_bins_trace.sh <scale> <tag> [settings ...]
set -e -o pipefail
S=${1:-24}; TAG=${2:-trace}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/bins_$TAG
mkdir -p $OUT
export TMPDIR=/tmp KSG_SPMV_REPS=${KSG_SPMV_REPS:-10}
cd /tmp
timeout -k 10 ${KSG_TRACE_TIMEOUT:-900} rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $R/tools/spmv_bins.py $S "$@" > $OUT/run.txt 2> $OUT/run.err
cd $R
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
acc = collections.OrderedDict()
with open(out + "/t_kernel_trace.csv") as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc.setdefault(k, [0, 0])
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(out + "/summary.txt", "w") as o:
    o.write(open(out + "/run.txt").read())
    o.write("\nkernels of the run (all settings together; the first product of every setting included)\n")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        o.write("%-60s %6d launches %10.4f ms each\n" % (k[:60], n, t / n / 1e6))
print(open(out + "/summary.txt").read())
PY
rm -f $OUT/t_kernel_trace.csv
