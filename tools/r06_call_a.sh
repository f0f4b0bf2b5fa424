#!/bin/bash
# round 6, GPU call A: the full -m gpu suite on the current build, then same-session A/B of three conv_wino.hip builds, then the lr180 probe
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
AB=pnp_vcve_amd/lib/ab
for rep in 1 2; do
  bash tools/try_libs.sh $O/ab_bench.txt $AB/lib_head.so $AB/lib_jit.so $AB/lib_jitrev.so -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
done
python - <<'PY'
import json,re
for l in open('gpurun_out/r06a/ab_bench.txt'):
    if l.startswith('==='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('  value %.2f  block avg %.1f us  frac %.3f  parity %s' % (d['value'], r['avg_launch_us'], r['frac'], d.get('parity',{}).get('max_abs_diff_vs_cpu')))
PY
for i in 1 2 3; do python tools/lr180_spread.py --tag run$i >> $O/lr180_spread.txt 2>&1; done; cat $O/lr180_spread.txt
