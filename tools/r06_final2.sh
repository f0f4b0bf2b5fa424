#!/bin/bash
# round 6, final GPU call: full -m gpu suite at HEAD, then tools/r06_final.sh (profile passes, driver-form bench, 180x320 trace)
mkdir -p gpurun_out/r06final
python -m pytest tests -m gpu -x -q > gpurun_out/r06final/gpu_tests.txt 2>&1; tail -3 gpurun_out/r06final/gpu_tests.txt
rm -rf gpurun_out/prof_r06 gpurun_out/prof_r06_lr180
bash tools/r06_final.sh
