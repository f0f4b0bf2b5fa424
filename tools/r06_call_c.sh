#!/bin/bash
# round 6, GPU call C: full suite; the F(4x4) upper-bound ubench next to the F(2x2) one; lr180 with 8x8 blocks (ragged fold) head vs current
O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Xclang -target-feature -Xclang -packed-fp32-ops -mllvm -pragma-unroll-threshold=1000000 -o ub_winograd ub_winograd.hip 2>/dev/null )
timeout 300 tools/ubench/ub_winograd_f4 > $O/ub_winograd_f4.txt 2>&1; cat $O/ub_winograd_f4.txt
timeout 300 tools/ubench/ub_winograd > $O/ub_winograd_f2.txt 2>&1; tail -12 $O/ub_winograd_f2.txt
AB=pnp_vcve_amd/lib/ab
cp pnp_vcve_amd/lib/libpnpvcve_hip.so $AB/lib_cur.so
for rep in 1 2; do
  bash tools/try_libs.sh $O/ab_lr180.txt $AB/lib_head.so $AB/lib_cur.so -- python bench.py --workload lr180 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary
done
python - <<'PY'
import json
for l in open('gpurun_out/r06c/ab_lr180.txt'):
    if l.startswith('==='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('  value %.2f  block avg %.1f us  frac %.3f' % (d['value'], r['avg_launch_us'], r['frac']))
PY
