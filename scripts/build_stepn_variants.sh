#!/bin/bash
# Variant builds of libfdwave that differ in fdw_stepn.hip only (the wave-pipeline kernels), for A/B timing on the GPU box (development tool):
#   scripts/build_stepn_variants.sh name "flags" [name "flags" ...]   ->  ablate/libfdwave_<name>.so   (select with FDW_LIB=...)
# Every other object is taken from the regular build (run make first).  Timing experiments only: -DFDW_ABL_BITS=... builds give wrong results.
set -e
cd "$(dirname "$0")/.."
C=parallel_finite_difference_computation_amd/csrc
B=$C/build
mkdir -p ablate
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$C $flags -c $C/fdw_stepn.hip -o ablate/fdw_stepn_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate/libfdwave_$name.so ablate/fdw_stepn_$name.o $B/fdw_step1.o $B/fdw_step1_dd.o $B/fdw_step1_fast.o $B/fdw_step2.o $B/fdw_border.o \
        $B/fdw_api.o $B/fdw_comm.o $B/fdw_slabs.o $B/fdw_trace.o $B/fdw_host.o $B/fdw_config.o -lm -ldl -lpthread ) &
done
wait; rm -f ablate/*.o
ls -la ablate/*.so
