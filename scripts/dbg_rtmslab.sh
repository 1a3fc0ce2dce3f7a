#!/bin/bash
# GPU box (development tool): scripts/dbg_rtmslab.sh <repetitions> <bench.py arguments ...>  -- one line per repetition: the bitwise check of the decomposed run
n=${1:-12}; shift
for i in $(seq 1 $n); do
  python bench.py --size 1024 --warmup 4 --no-cpu-baseline --backend shm --no-exposed "$@" 2>&1 | grep "\[check\]" | cut -c1-300 | sed "s/^/run $i: /"
done
