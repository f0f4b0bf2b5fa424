#!/bin/bash
# round 6, GPU call B: the full -m gpu suite on the current build (two translation units, ragged fold), A/B head vs current at 720p and
# 180x320, and the exact command of round 5's red test (bench.py --steps 1 --warmup 1 --no-cpu-baseline) twice
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
AB=pnp_vcve_amd/lib/ab
cp pnp_vcve_amd/lib/libpnpvcve_hip.so $AB/lib_cur.so
for rep in 1 2; do
  bash tools/try_libs.sh $O/ab_bench.txt $AB/lib_head.so $AB/lib_cur.so -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
  bash tools/try_libs.sh $O/ab_lr180.txt $AB/lib_head.so $AB/lib_cur.so -- python bench.py --workload lr180 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary
done
python - <<'PY'
import json
for f in ('ab_bench','ab_lr180'):
    print(f)
    for l in open(f'gpurun_out/r06b/{f}.txt'):
        if l.startswith('==='): print(l.strip())
        elif l.startswith('{'):
            d=json.loads(l); r=d['roofline']
            print('  value %.2f  block avg %.1f us  frac %.3f  dev ms %s' % (d['value'], r['avg_launch_us'], r['frac'], d.get('kernel_events')))
PY
for rep in 1 2; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/driver_form_$rep.txt 2>&1
  python - <<'PY'
import json
d=json.load(open('bench_secondary.json'))
print('headline', d['value'], [ (e['workload'], e['precision'], round(e['value'],1)) for e in d['secondary'][10:14]])
PY
done
