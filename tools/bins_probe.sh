#!/bin/bash
# timing experiments on the two kernels of the binned product (KSG_BINS_PROBE: 1 reduce without the LDS
# additions, 2 expand writes where it reads, 4 expand without the LDS lookup); results are wrong by design
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pr in 0 1 2 4 6; do
  export KSG_BINS_PROBE=$pr
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_bins -o probe$pr -- python3 $R/tools/spmv_bins.py ${1:-24} ${2:-W=8192,H=2048,T=4} > $R/gpurun_out/bins_probe$pr.log 2>&1 || exit 1
  echo "== probe $pr"; python3 $R/tools/rocpd_kernels.py $R/gpurun_out/prof_bins/probe${pr}_results.db k_bins
done
