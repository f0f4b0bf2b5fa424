#!/bin/bash
# builds of libpnpvcve_hip.so that differ in the compile flags of conv_wino.hip AND conv_wino_ms.hip (the same source, two translation
# units) only:  tools/build_wino_variants.sh name1 "flags1" name2 "flags2" ...
#   -> pnp_vcve_amd/lib/ab/lib_<name>.so   (run them in turn with tools/try_libs.sh / tools/ab_libs.sh)
# The variant flags are ADDED to build_native.py's FLAGS + EXTRA_FLAGS of each unit (round 5 compiled the variants without them:
# packed fp32 ops on -- the known-wrong build -- and the K loop rolled with its accumulators in scratch memory).
L=pnp_vcve_amd/lib
mkdir -p $L/ab
python -m pnp_vcve_amd.build_native > /dev/null || exit 1           # the other objects, current
flags_of() { python -c "from pnp_vcve_amd import build_native as b; print(' '.join(f for f in b.FLAGS + b.EXTRA_FLAGS['$1'] if f != '-Wall'))"; }
OBJS=$(ls $L/obj/*.o | grep -v "conv_wino.o\|conv_wino_ms.o" | tr '\n' ' ')
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc $(flags_of conv_wino.hip) -w $f -c pnp_vcve_amd/csrc/conv_wino.hip -o $L/ab/conv_wino_$n.o || exit 1
  /opt/rocm/bin/hipcc $(flags_of conv_wino_ms.hip) -w $f -c pnp_vcve_amd/csrc/conv_wino_ms.hip -o $L/ab/conv_wino_ms_$n.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/ab/lib_$n.so $OBJS $L/ab/conv_wino_$n.o $L/ab/conv_wino_ms_$n.o && echo built $L/ab/lib_$n.so
done
