#!/bin/bash
# Timing-experiment builds of libfdwave (results are wrong by construction): gpurun_out is scratch.
set -e
cd "$(dirname "$0")/.."
C=parallel_finite_difference_computation_amd/csrc
mkdir -p ablate
for a in "$@"; do
  D="-DFDW_ABL_BITS=$a"; case $a in b*) D="-DFDW_ABL_BITS=${a#b}";; nt*) D="-DFDW_NT=${a#nt}";; esac
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Iinclude -I$C $D -c $C/fdw_kernels.hip -o ablate/k$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ablate/libfdwave_a$a.so ablate/k$a.o $C/build/fdw_api.o $C/build/fdw_host.o $C/build/fdw_config.o -lm ) &
done
wait; rm -f ablate/*.o
ls -la ablate/*.so
