#!/bin/bash
# ms per restart of bench.py for a range of row-grid bounds of the update kernel
for g in 256 512 768 1024 1280 1536 2048 3072; do
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --mv-grid $g > /tmp/b.json 2>/dev/null || exit 1
  python - "$g" <<'PY'
import json,sys
l=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print("mv_grid %5s: %8.3f ms/restart  multvec %7.1f GB/s  dotvec %7.1f GB/s" % (sys.argv[1], l["ms_per_step"], l["kernels"]["multvec"]["GBps"], l["kernels"]["dotvec"]["GBps"]), flush=True)
PY
done
