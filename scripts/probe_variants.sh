#!/bin/bash
# GPU box: scripts/probe_variants.sh <variant> ...  -- the wave-pipeline kernel of ablate/libfdwave_<variant>.so in FAST and EXACT numerics (development tool)
for v in "$@"; do
  for num in 1 0; do
    FDW_LIB=$PWD/ablate/libfdwave_$v.so PIPE_NUMERICS=$num PIPE_CHUNKS=${PIPE_CHUNKS:-173} python3 scripts/probe_pipe.py ${PIPE_SIZE:-8192} 2>&1 | grep "Gpt\|Error\|error" | sed "s/^/numerics=$num /"
  done
done
