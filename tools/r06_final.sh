#!/bin/bash
# round 6, final GPU call: profile passes of the headline at HEAD (kernel trace + PMC), the bench in the driver's form (with the CPU
# baseline and every secondary workload), the 180x320 fp32 kernel trace
O=gpurun_out/r06final; mkdir -p $O
git rev-parse HEAD > $O/head.txt 2>/dev/null
bash tools/profile_gpu.sh r06 > $O/profile.log 2>&1; head -20 gpurun_out/prof_r06/summary.txt | cut -c1-140
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_stderr.txt; cp bench_secondary.json $O/bench_secondary.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06final/bench_line.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('value','ms_per_step','steps','warmup')}); print(d['roofline']); print(d.get('roofline_mv_warp')); print(d.get('cpu_baseline')); print(d.get('parity')); print(d.get('north_star_128')); print(d.get('opt_in_720p'))
s=json.load(open('bench_secondary.json'))
for e in s['secondary']: print(round(e['value'],1) if e.get('value') else e.get('skipped'), '|', e['name'][:110])
PY
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06_lr180/trace -- python3 bench.py --workload lr180 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary > $O/lr180_trace.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_r06_lr180 gpurun_out/prof_r06_lr180/pmc.json > gpurun_out/prof_r06_lr180/summary.txt 2>&1; head -14 gpurun_out/prof_r06_lr180/summary.txt | cut -c1-140
