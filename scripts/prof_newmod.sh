#!/bin/bash
# kernel-trace of the forward loop on the new_mod grid (415x295): kernel duration vs step period (development tool)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_newmod -- python3 scripts/probe_one.py 415 295 -1 0 1000 > gpurun_out/prof_newmod.log 2>&1
head -4 gpurun_out/prof_newmod/*/*kernel_stats.csv | cut -c1-200
tail -2 gpurun_out/prof_newmod.log
