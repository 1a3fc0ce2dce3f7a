for n in 8064 8128 8192 8256 8320; do
  for num in exact fast; do
    python3 bench.py --size $n --steps 40 --warmup 8 --no-cpu-baseline --no-extra --no-fast-line --numerics $num 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('size $n $num', d['value'], 'Gpt/s  launch_us', d['roofline']['launch_us'])
"
  done
done
