#!/bin/bash
# GPU box: the wave-pipeline kernel in FAST and EXACT numerics through the ablation builds of scripts/build_stepn_variants.sh (development tool)
for v in base b256 b512 b768 b1024 b128 b64 b2048; do
  for num in 1 0; do
    FDW_LIB=$PWD/ablate/libfdwave_$v.so PIPE_NUMERICS=$num PIPE_CHUNKS=173 python3 scripts/probe_pipe.py 8192 2>&1 | grep Gpt | sed "s/^/numerics=$num /"
  done
done
