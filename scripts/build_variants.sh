#!/bin/bash
# Variant builds of libfdwave for A/B timing on the GPU box (development tool): scripts/build_variants.sh name "flags" [name "flags" ...]
# -> ablate/libfdwave_<name>.so (select with FDW_LIB=...).  EVERY kernel translation unit is rebuilt with the flags (fdw_step1.hip three times, as
# the Makefile does); the host objects come from the regular build (run make first).  For variants of the wave-pipeline kernels alone
# scripts/build_stepn_variants.sh is five times quicker.  -DFDW_ABL_BITS=... builds give wrong results: timing experiments only.
set -e
cd "$(dirname "$0")/.."
C=parallel_finite_difference_computation_amd/csrc
B=$C/build
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$C"
mkdir -p ablate
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( $H $flags -c $C/fdw_step1.hip -o ablate/s1_$name.o & $H $flags -DFDW_TU=1 -c $C/fdw_step1.hip -o ablate/s1d_$name.o & $H $flags -DFDW_TU=2 -c $C/fdw_step1.hip -o ablate/s1f_$name.o &
    $H $flags -c $C/fdw_step2.hip -o ablate/s2_$name.o & $H $flags -c $C/fdw_stepn.hip -o ablate/sn_$name.o & $H $flags -c $C/fdw_border.hip -o ablate/bd_$name.o & wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate/libfdwave_$name.so ablate/s1_$name.o ablate/s1d_$name.o ablate/s1f_$name.o ablate/s2_$name.o ablate/sn_$name.o ablate/bd_$name.o \
        $B/fdw_api.o $B/fdw_comm.o $B/fdw_slabs.o $B/fdw_trace.o $B/fdw_host.o $B/fdw_config.o -lm -ldl -lpthread )
done
rm -f ablate/*.o
ls -la ablate/*.so
