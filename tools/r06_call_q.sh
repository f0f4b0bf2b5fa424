#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O
AB=pnp_vcve_amd/lib/ab
for rep in 1 2; do
  bash tools/try_libs.sh $O/ab_128.txt $AB/lib_qcur.so $AB/lib_qsameb.so -- python bench.py --workload 128 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary
done
python - <<'PY'
import json
for l in open('gpurun_out/r06q/ab_128.txt'):
    if l.startswith('==='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); print('  value %.1f  ms/step %.3f' % (d['value'], d['ms_per_step']))
PY
