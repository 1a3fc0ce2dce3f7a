#!/bin/bash
# Timing-experiment builds of libfdwave (results are wrong by construction): scripts/build_ablations.sh <bits> [<bits> ...]
#   <bits> or b<bits> -> -DFDW_ABL_BITS=<bits> (the bit list is in csrc/fdw_device.h), nt<n> -> -DFDW_NT=<n>; output ablate/libfdwave_a<arg>.so.
# A front end of scripts/build_variants.sh (which rebuilds every kernel translation unit with the flag).
cd "$(dirname "$0")/.."
args=()
for a in "$@"; do
  D="-DFDW_ABL_BITS=$a"; case $a in b*) D="-DFDW_ABL_BITS=${a#b}";; nt*) D="-DFDW_NT=${a#nt}";; esac
  args+=("a$a" "$D")
done
exec scripts/build_variants.sh "${args[@]}"
