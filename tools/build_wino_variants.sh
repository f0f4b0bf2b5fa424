#!/bin/bash
# builds of libpnpvcve_hip.so that differ in the compile flags of conv_wino.hip AND conv_wino_ms.hip (the same source, two translation
# units) only:  tools/build_wino_variants.sh name1 "flags1" name2 "flags2" ...
#   -> pnp_vcve_amd/lib/ab/lib_<name>.so   (run them in turn with tools/try_libs.sh / tools/ab_libs.sh)
# The variant flags are ADDED to build_native.py's FLAGS + EXTRA_FLAGS of each unit, and the units go through the same pipeline as the
# library's (build_native.compile_unit: device listing, store-hazard padding, assembler).
while [ $# -ge 2 ]; do
  python -m pnp_vcve_amd.build_native --variant "$1" "$2" || exit 1
  shift 2
done
