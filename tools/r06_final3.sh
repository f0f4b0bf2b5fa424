#!/bin/bash
# round 6, last GPU action: the full -m gpu suite at HEAD + smoke + one driver-form bench line
mkdir -p gpurun_out/r06final3
python -m pytest tests -m gpu -x -q > gpurun_out/r06final3/gpu_tests.txt 2>&1; tail -3 gpurun_out/r06final3/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06final3/smoke.txt 2>&1; tail -2 gpurun_out/r06final3/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06final3/bench_line.json 2> gpurun_out/r06final3/bench_stderr.txt; cut -c1-300 gpurun_out/r06final3/bench_line.json
python tools/lint_store_hazard.py > gpurun_out/r06final3/lint.txt 2>&1; cat gpurun_out/r06final3/lint.txt
