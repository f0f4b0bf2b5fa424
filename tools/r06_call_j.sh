#!/bin/bash
# round 6, GPU call J: full suite at HEAD, then the profile passes (kernel trace + PMC) of the headline
O=gpurun_out/r06j; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
bash tools/profile_gpu.sh r06 > $O/profile.log 2>&1; tail -45 $O/profile.log | cut -c1-180
