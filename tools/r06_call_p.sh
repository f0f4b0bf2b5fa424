#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O
python -m pytest tests/test_gpu_wino.py tests/test_gpu_generator.py -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
AB=pnp_vcve_amd/lib/ab
cp pnp_vcve_amd/lib/libpnpvcve_hip.so $AB/lib_cur.so
for rep in 1 2; do
  bash tools/try_libs.sh $O/ab_bench.txt $AB/lib_lump8.so $AB/lib_lump4.so $AB/lib_cur.so $AB/lib_lump1.so -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
done
python - <<'PY'
import json
for l in open('gpurun_out/r06q/ab_bench.txt'):
    if l.startswith('==='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('  value %.2f  ms/step %.2f  block avg %.1f us  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac']))
PY
