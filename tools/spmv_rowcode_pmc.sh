#!/bin/bash
# Hardware counters of the row-code product on a stencil operator (tools/spmv_bench.py): bytes fetched and
# written beyond L2, fabric requests, L2 hits and misses, one rocprofv3 --pmc pass per counter group, for
# several numbers of workgroups per launch (KSG_RC_GRID).
# Usage: tools/spmv_rowcode_pmc.sh <spec, e.g. lap3d:512> <tag> [grids ...]  -> gpurun_out/pmc_rc_<tag>/summary.txt
set -e -o pipefail
SPEC=${1:-lap3d:512}; TAG=${2:-rc}; shift 2 || true
GRIDS=${@:-16384}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_rc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for G in $GRIDS; do
  export KSG_RC_GRID=$G
  timeout -k 10 300 python3 $R/tools/spmv_bench.py $SPEC > $OUT/plain_$G.txt 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    i=$((i+1))
    timeout -s KILL ${KSG_PMC_PASS_TIMEOUT:-200} rocprofv3 --pmc $set -d $OUT -o g${G}_pass$i --output-format csv -- python3 $R/tools/spmv_bench.py $SPEC > $OUT/g${G}_pass$i.txt 2> $OUT/g${G}_pass$i.err || echo "grid $G pass $i ($set) failed" >> $OUT/failed.txt
  done
done
cd $R
python3 - "$OUT" $GRIDS <<'PY'
import csv, glob, sys, collections
out, grids = sys.argv[1], sys.argv[2:]
with open(out + "/summary.txt", "w") as o:
    for g in grids:
        acc = collections.defaultdict(list)
        for f in sorted(glob.glob(out + "/g%s_pass*counter_collection.csv" % g)):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if "k_spmv_rowcode" in row["Kernel_Name"]:
                        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        o.write("== at most %s workgroups per launch\n" % g)
        o.write(open(out + "/plain_%s.txt" % g).read())
        for c, v in sorted(acc.items()):
            o.write("k_spmv_rowcode   %-28s %16.0f  (mean of %d launches)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/summary.txt").read())
PY
rm -f $OUT/*counter_collection.csv $OUT/*agent_info.csv
