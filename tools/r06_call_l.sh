#!/bin/bash
# round 6, GPU call L: the seam cut again (all single-source kernels | branch kernels only), same session; lr180 on 4x4-block maps = branch kernels
O=gpurun_out/r06l; mkdir -p $O
AB=pnp_vcve_amd/lib/ab
cp pnp_vcve_amd/lib/libpnpvcve_hip.so $AB/lib_cur.so
for rep in 1 2; do
  bash tools/try_libs.sh $O/ab_bench.txt $AB/lib_cur.so $AB/lib_seam.so -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
  bash tools/try_libs.sh $O/ab_par.txt $AB/lib_cur.so $AB/lib_seampar.so -- python tools/bench_wino.py --h 720 --w 1280 --iters 30 --rounds 1
done
python - <<'PY'
import json
for l in open('gpurun_out/r06l/ab_bench.txt'):
    if l.startswith('==='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('  value %.2f  ms/step %.2f  block avg %.1f us  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac']))
PY
grep "===\|winograd" gpurun_out/r06l/ab_par.txt
