#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun).  Kernel trace and PMC counters are
# collected in SEPARATE rocprofv3 runs (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit
# one pass; --pmc must not be combined with other trace domains on this pool).
#   usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_lds -- python3 bench.py $ARGS > $OUT/pmc_lds.log 2>&1
find $OUT -name '*.csv' | head -50
python3 tools/summarize_profile.py $OUT $OUT/pmc.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
