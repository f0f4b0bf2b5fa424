#!/bin/bash
# round 6, GPU call D: full suite; F(4x4) ubench; kernel trace (rocprofv3 --kernel-trace --stats) of the headline
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 300 tools/ubench/ub_winograd_f4 > $O/ub_winograd_f4.txt 2>&1; cat $O/ub_winograd_f4.txt
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary > $O/trace.log 2>&1
python3 tools/summarize_profile.py $O $O/pmc.json > $O/summary.txt 2>&1; head -40 $O/summary.txt
