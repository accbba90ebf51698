#!/bin/bash
# Hardware counters of the binned product (ksg_bins.hip) on an R-MAT graph Laplacian: bytes fetched from and
# written to HBM and the fabric requests of its kernels, one rocprofv3 --pmc pass per counter group, next to
# the bytes the library's model charges; instruction counts and LDS bank conflicts from the SQ.   Usage: tools/spmv_bins_pmc.sh <scale>  -> gpurun_out/pmc_bins<scale>/summary.txt
set -e -o pipefail
S=${1:-24}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_bins$S
mkdir -p $OUT
export TMPDIR=/tmp KSG_RMAT_CACHE=1 KSG_SPMV_REPS=5
cd /tmp
timeout -k 10 900 python3 $R/tools/spmv_bins.py $S W=16384 > $OUT/plain.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL ${KSG_PMC_PASS_TIMEOUT:-300} rocprofv3 --pmc $set -d $OUT -o pass$i --output-format csv -- python3 $R/tools/spmv_bins.py $S W=16384 > $OUT/pass$i.txt 2> $OUT/pass$i.err || echo "pass $i ($set) failed" >> $OUT/failed.txt
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
            for name in ("k_bins_expand", "k_bins_reduce", "k_bins_long_rows", "k_spmv_tiles", "k_gather"):
                if name in k:
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    o.write(open(out + "/plain.txt").read())
    o.write("\ncounters per launch (mean over the 6 launches of each pass), R-MAT scale %s; FETCH_SIZE and WRITE_SIZE in KiB as the tool prints them\n" % scale)
    for k in acc:
        for c, v in sorted(acc[k].items()):
            o.write("%-16s %-28s %16.0f  (%d launches)\n" % (k, c, sum(v) / len(v), len(v)))
print(open(out + "/summary.txt").read())
PY
rm -f $OUT/*counter_collection.csv $OUT/*agent_info.csv
