#!/bin/bash
# headline bench with two builds of the library in turn (A B A B), no side file:  tools/ab_bench_two_libs.sh libA.so libB.so [bench args]
a=$1; b=$2; shift 2
lib=pnp_vcve_amd/lib/libpnpvcve_hip.so
orig=$(mktemp /tmp/_lib_orig.XXXXXX.so); cp $lib $orig
trap 'cp $orig $lib; rm -f $orig' EXIT
for rep in 1 2; do for v in $a $b; do cp $v $lib; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"; done; done
