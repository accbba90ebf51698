#!/bin/bash
# Hardware counters of the power-law SpMV: L1 (TCP) requests to L2, L2 hits and misses, HBM
# fetch bytes and texture-addresser busy time of k_spmv_tiles on an R-MAT graph Laplacian,
# one rocprofv3 --pmc pass per counter group (a pass holds few counters per block).
# Usage: tools/spmv_rmat_pmc.sh <scale> [hot limit]   -> gpurun_out/pmc_rmat<scale>/summary.txt
set -e -o pipefail
S=${1:-23}
HOT=${2:-1048576}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_rmat$S
mkdir -p $OUT
export TMPDIR=/tmp KSG_RMAT_CACHE=1 KSG_SPMV_REPS=5
cd /tmp
timeout -k 10 ${KSG_PMC_PLAIN_TIMEOUT:-600} python3 $R/tools/spmv_rmat.py $S $HOT > $OUT/plain.txt 2>&1
i=0
# KSG_PMC_SHORT=1: two passes only (large scales, where every pass spends minutes setting the operator up)
if [ -n "$KSG_PMC_SHORT" ]; then
  SETS=("TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE")
else
  SETS=("TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
        "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
        "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
        "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum")
fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -s KILL ${KSG_PMC_PASS_TIMEOUT:-240} rocprofv3 --pmc $set -d $OUT -o pass$i --output-format csv -- python3 $R/tools/spmv_rmat.py $S $HOT > $OUT/pass$i.txt 2> $OUT/pass$i.err || echo "pass $i ($set) failed" >> $OUT/failed.txt
done
cd $R
python3 - "$OUT" "$S" <<'PY'
import csv, glob, sys, collections
out, scale = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            k = "k_spmv_tiles" if "k_spmv_tiles" in k else ("k_gather" if "k_gather" in k else None)
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    o.write(open(out + "/plain.txt").read())
    o.write("\ncounters per launch (mean over the launches of each pass) and per product (6 products per pass: an operator in\n"
            "column groups launches the kernel once per group), R-MAT scale %s\n" % scale)
    for k in acc:
        for c, v in sorted(acc[k].items()):
            o.write("%-14s %-36s %16.0f  (%d launches)  %16.0f per product\n" % (k, c, sum(v) / len(v), len(v), sum(v) / 6.0))
print(open(out + "/summary.txt").read())
PY
rm -f $OUT/*counter_collection.csv $OUT/*agent_info.csv
