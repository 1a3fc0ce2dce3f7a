#!/bin/bash
# GPU box: the bench lines kept under profiles/ (development tool): scripts/bench_all.sh <round tag>
r=${1:-r03}; o=gpurun_out
j() { tail -n 1; }
python bench.py --steps 20 --warmup 5 2>$o/bench_err.log | j > $o/${r}_bench_8192_steps20.json
python bench.py 2>>$o/bench_err.log | j > $o/${r}_bench_8192_steps1000.json
python bench.py --size 4096 --no-cpu-baseline 2>>$o/bench_err.log | j > $o/${r}_bench_4096.json
python bench.py --size 16384 --steps 400 --no-cpu-baseline 2>>$o/bench_err.log | j > $o/${r}_bench_16384.json
python bench.py --numerics fast --no-cpu-baseline 2>>$o/bench_err.log | j > $o/${r}_bench_8192_fast.json
python bench.py --size 16384 --steps 400 --numerics fast --no-cpu-baseline 2>>$o/bench_err.log | j > $o/${r}_bench_16384_fast.json
python bench.py --workload rtm-slab --steps 202 --warmup 10 2>>$o/bench_err.log | j > $o/${r}_bench_rtm_slab_8192.json
python bench.py --workload rtm-slab --steps 202 --warmup 10 --numerics fast --no-cpu-baseline 2>>$o/bench_err.log | j > $o/${r}_bench_rtm_slab_8192_fast.json
python bench.py --workload model --steps 200 --warmup 20 2>>$o/bench_err.log | j > $o/${r}_bench_model_8192.json
python bench.py --workload model --steps 200 --warmup 20 --numerics fast --no-cpu-baseline 2>>$o/bench_err.log | j > $o/${r}_bench_model_8192_fast.json
python bench.py --workload stencil --steps 200 --warmup 20 2>>$o/bench_err.log | j > $o/${r}_bench_stencil_8192.json
python bench.py --workload rtm 2>>$o/bench_err.log | j > $o/${r}_bench_rtm_new_mod.json
python bench.py --gpus 2 --backend shm --size 2048 --steps 48 --warmup 8 --no-cpu-baseline 2>>$o/bench_err.log | j > $o/${r}_bench_2ranks_shm_rehearsal_2048.json
for f in $o/${r}_bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
extra = ""
if "fast_numerics" in d: extra += f"  fast {d['fast_numerics']['value']}"
if "extra" in d: extra += f"  4096^2: {d['extra']['baseline_config_2']['value']}"
if d.get("cpu_baseline"): extra += f"  cpu {d['cpu_baseline']['value']} x{d['cpu_baseline']['cores']}"
print(sys.argv[1].split('/')[-1], d["value"], d["unit"], "frac", d["roofline"]["frac"], extra)
PY
done
