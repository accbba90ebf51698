#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel trace + stats, then FETCH_SIZE and
# WRITE_SIZE in separate counter passes.  Usage: tools/profile_bench.sh <tag>
set -e -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# BENCH_ARGS overrides the workload, e.g. "--workload lap3d --grid 512 --steps 3 --warmup 2 --no-cpu-baseline"
ARGS=${BENCH_ARGS:-"--steps 5 --warmup 2 --no-cpu-baseline"}
export ARGS
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- python3 $R/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
# EXTRA_PMC="TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum": one more counter pass, summed per kernel
if [ -n "$EXTRA_PMC" ]; then
  timeout -s KILL 400 rocprofv3 --pmc $EXTRA_PMC -d $OUT -o extra --output-format csv -- python3 $R/bench.py $ARGS > $OUT/bench_extra.json 2> $OUT/extra.err
fi
cd $R
if [ -n "$EXTRA_PMC" ]; then
  python3 tools/prof_summary.py counters $OUT/extra_counter_collection.csv > $OUT/summary_extra.txt
  rm -f $OUT/extra_counter_collection.csv
fi
python3 tools/prof_summary.py trace $OUT/trace_kernel_trace.csv > $OUT/summary_trace.txt
python3 tools/prof_summary.py pmc $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv $OUT/bench_trace.json > $OUT/summary_pmc.txt
# a record of the file bench.py reads for roofline.traffic (add it to "records" of profiles/pmc_traffic.json)
python3 - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
kern = json.loads(open(out + "/summary_pmc.txt").read().strip().splitlines()[-1])
line = json.loads(open(out + "/bench_trace.json").read().strip().splitlines()[-1])
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py " + os.environ["ARGS"] +
                     "`, dispatches between the bench markers; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes "
                     "(FETCH_SIZE reads half of a wide coalesced stream on gfx950; calibrated on k_scale_cols, 8n read + 8n written "
                     "per launch: profiles/r01c_bench_pmc_summary.txt)",
           "workload": line["config"]["workload"], "steps": line["steps"], "warmup": line["warmup"], "kernels": kern},
          open(out + "/pmc_traffic.json", "w"), indent=1)
PY
rm -f $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv
tail -3 $OUT/bench_trace.json | cut -c1-600
cat $OUT/summary_trace.txt | head -20
cat $OUT/summary_pmc.txt | head -20
