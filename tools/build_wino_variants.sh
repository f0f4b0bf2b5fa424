#!/bin/bash
# builds of libpnpvcve_hip.so that differ in conv_wino.hip's compile flags only:  tools/build_wino_variants.sh name1 "flags1" name2 "flags2" ...
#   -> pnp_vcve_amd/lib/ab/lib_<name>.so   (run them in turn with tools/try_libs.sh)
L=pnp_vcve_amd/lib
mkdir -p $L/ab
OBJS=$(ls $L/obj/*.o | grep -v conv_wino.o | tr '\n' ' ')
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $f -c pnp_vcve_amd/csrc/conv_wino.hip -o $L/ab/conv_wino_$n.o 2>&1 | grep -v "not a recognized feature"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/ab/lib_$n.so $OBJS $L/ab/conv_wino_$n.o && echo built $L/ab/lib_$n.so
done
