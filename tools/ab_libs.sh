#!/bin/bash
# A/B of two builds of libpnpvcve_hip.so on ONE box in ONE session (boxes differ by up to 12 % in clock under load):
#   tools/ab_libs.sh OUTDIR LIB_A LIB_B -- <command that prints one result>       (runs A B A B)
out=$1; a=$2; b=$3; shift 4
lib=pnp_vcve_amd/lib/libpnpvcve_hip.so
orig=$(mktemp /tmp/_lib_orig.XXXXXX.so)
cp $lib $orig
trap 'cp $orig $lib; rm -f $orig' EXIT        # an interrupted or failing run must not leave build A or B installed
mkdir -p $out
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then cp $a $lib; else cp $b $lib; fi
    echo "=== $v rep $rep" >> $out/ab.txt
    "$@" >> $out/ab.txt 2>&1
  done
done
