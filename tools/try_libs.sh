#!/bin/bash
# run one command with each of several builds of the library installed in turn:  tools/try_libs.sh OUT lib1.so lib2.so ... -- cmd
out=$1; shift
libs=()
while [ "$1" != "--" ]; do libs+=("$1"); shift; done
shift
lib=pnp_vcve_amd/lib/libpnpvcve_hip.so
orig=$(mktemp /tmp/_lib_orig.XXXXXX.so)
cp $lib $orig
trap 'cp $orig $lib; rm -f $orig' EXIT
mkdir -p $(dirname $out)
for l in "${libs[@]}"; do
  cp $l $lib
  echo "=== $l" >> $out
  "$@" >> $out 2>&1
done
