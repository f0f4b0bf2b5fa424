#!/bin/bash
# split-fp16 tile queue A/B in ONE session (same box): static walk / queue / static / queue, frames/s and PSNR per workload
for rep in 1 2; do for q in 0 1; do
  echo "=== tile queue $q rep $rep"
  for a in "--steps 5" "--workload 128 --steps 20" "--workload lr180 --clips 3 --steps 10"; do
    python bench.py --precision f16x3 --tile-queue $q $a --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null |
      python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["psnr_per_rank"][0])'
  done
done; done
