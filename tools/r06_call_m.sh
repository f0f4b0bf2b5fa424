#!/bin/bash
# round 6, GPU call M: full suite at HEAD; configs[4] (180x320 fp16) with 3 / 8 clips in flight, with and without hipGraph replay
O=gpurun_out/r06m; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
for args in "--clips 3" "--clips 3 --graphs" "--clips 8" "--clips 8 --graphs"; do
  echo "== lr180 fp16 $args" >> $O/lr180_fp16.txt
  python bench.py --workload lr180 --precision fp16 $args --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events >> $O/lr180_fp16.txt 2>&1
done
python - <<'PY'
import json
for l in open('gpurun_out/r06m/lr180_fp16.txt'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('  value %.1f frames/s  ms/step %.3f  roofline %s frac %.3f wall %.3f' % (d['value'], d['ms_per_step'], r['bound'], r['frac'], r.get('frac_wall', 0)))
PY
