set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq24; mkdir -p $OUT; export TMPDIR=/tmp KSG_RMAT_CACHE=1 KSG_SPMV_REPS=3; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $set -d $OUT -o p$i --output-format csv -- python3 $R/tools/spmv_bins.py 24 W=16384 > $OUT/p$i.txt 2> $OUT/p$i.err || echo "pass $i failed" >> $OUT/failed.txt
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out+"/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]
        for name in ("k_bins_expand","k_bins_reduce"):
            if name in k: acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    for c,v in sorted(acc[k].items()): print("%-16s %-26s %16.0f (%d)"%(k,c,sum(v)/len(v),len(v)))
PY
rm -f $OUT/*counter_collection.csv
