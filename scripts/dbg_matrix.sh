R="scripts/dbg_rtmslab.sh"
echo "== A rtm-slab 3 ranks k3 steps 11 (known flaky)"; $R 15 --workload rtm-slab --gpus 3 --ksteps 3 --steps 11 | grep -c True
echo "== B same, one window"; $R 15 --workload rtm-slab --gpus 3 --ksteps 3 --steps 11 --max-windows 1 | grep -c True
echo "== C same, no overlap"; $R 15 --workload rtm-slab --gpus 3 --ksteps 3 --steps 11 --no-overlap | grep -c True
echo "== D forward only 3 ranks k3 steps 11"; $R 15 --workload forward --gpus 3 --ksteps 3 --steps 11 | grep -c True
echo "== E rtm-slab 2 ranks k3 steps 11"; $R 15 --workload rtm-slab --gpus 2 --ksteps 3 --steps 11 | grep -c True
echo "== F rtm-slab 3 ranks k4 steps 12 (whole cycles)"; $R 15 --workload rtm-slab --gpus 3 --ksteps 4 --steps 12 | grep -c True
