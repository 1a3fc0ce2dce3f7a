#!/bin/bash
# Variant builds of libfdwave for A/B timing on the GPU box (development tool): scripts/build_variants.sh name "flags" [name "flags" ...]
# -> ablate/libfdwave_<name>.so (select with FDW_LIB=...).  Only the kernel translation units are rebuilt with the flags.
set -e
cd "$(dirname "$0")/.."
C=parallel_finite_difference_computation_amd/csrc
mkdir -p ablate
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( for k in fdw_step1 fdw_step2 fdw_stepn fdw_border; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$C $flags -c $C/$k.hip -o ablate/${k}_$name.o || exit 1
    done &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate/libfdwave_$name.so ablate/fdw_step1_$name.o ablate/fdw_step2_$name.o ablate/fdw_stepn_$name.o ablate/fdw_border_$name.o \
        $C/build/fdw_api.o $C/build/fdw_comm.o $C/build/fdw_slabs.o $C/build/fdw_host.o $C/build/fdw_config.o -lm -ldl -lpthread ) &
done
wait; rm -f ablate/*.o
ls -la ablate/*.so
