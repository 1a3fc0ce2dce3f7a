#!/bin/bash
# GPU box: the profile set of a round (kernel trace + one counter group per pass each; scripts/profile_bench.py), in two or three calls:
#   scripts/profile_all.sh a | b | c | d        outputs under gpurun_out/prof_<tag>/ ; scripts/collect_profiles.py copies what is judged into profiles/
set -x
P="python3 scripts/profile_bench.py"
case "$1" in
a) $P fwd8192 -- --steps 200 --warmup 20
   $P fwd8192fast --workload-key forward-fast -- --numerics fast --steps 200 --warmup 20
   $P fwd4096 -- --size 4096 --steps 1000 --warmup 50 ;;
b) $P rtmslab8192 --workload-key rtm-slab -- --workload rtm-slab --steps 202 --warmup 10
   $P rtmslab8192fast --workload-key rtm-slab-fast -- --workload rtm-slab --numerics fast --steps 202 --warmup 10
   $P fwd16384 -- --size 16384 --steps 100 --warmup 12 ;;
c) $P model8192 --workload-key model -- --workload model --steps 200 --warmup 20
   $P model8192fast --workload-key model-fast -- --workload model --numerics fast --steps 200 --warmup 20
   $P stencil8192 --workload-key stencil -- --workload stencil --steps 200 --warmup 20
   $P fwd4096fast --workload-key forward-fast -- --numerics fast --size 4096 --steps 1000 --warmup 50 ;;
d) $P fwd16384fast --workload-key forward-fast -- --numerics fast --size 16384 --steps 100 --warmup 12 ;;
esac
