#!/bin/bash
# the command tools/ab_libs.sh runs for the split-fp16 path: three workloads, one "frames/s psnr" line each
#   tools/ab_libs.sh OUT LIB_A LIB_B -- tools/ab_bench_lines.sh [precision]
p=${1:-f16x3}
for a in "--steps 5" "--workload 128 --steps 20" "--workload lr180 --clips 3 --steps 10"; do
  python bench.py --precision $p $a --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null |
    python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["psnr_per_rank"][0])'
done
