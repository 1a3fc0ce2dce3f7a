#!/bin/bash
# Timing-experiment builds of libfdwave (results are wrong by construction): gpurun_out is scratch.
set -e
cd "$(dirname "$0")/.."
C=parallel_finite_difference_computation_amd/csrc
mkdir -p ablate
for a in "$@"; do
  D="-DFDW_ABL_BITS=$a"; case $a in b*) D="-DFDW_ABL_BITS=${a#b}";; nt*) D="-DFDW_NT=${a#nt}";; esac
  ( for k in fdw_step1 fdw_step2 fdw_stepn; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$C $D -c $C/$k.hip -o ablate/${k}_$a.o || exit 1
    done &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate/libfdwave_a$a.so ablate/fdw_step1_$a.o ablate/fdw_step2_$a.o ablate/fdw_stepn_$a.o $C/build/fdw_api.o $C/build/fdw_host.o $C/build/fdw_config.o -lm ) &
done
wait; rm -f ablate/*.o
ls -la ablate/*.so
