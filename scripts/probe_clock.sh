#!/bin/bash
# GPU box: scripts/probe_clock.sh <variant> ...  -- shader clock while the wave-pipeline kernel of ablate/libfdwave_<variant>.so runs (development tool):
# GRBM_GUI_ACTIVE per dispatch (one rocprofv3 --pmc process per case, nothing else collected) / 8 XCDs / the launch time the probe prints.
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
for v in "$@"; do
  for num in 1 0; do
    out=$R/gpurun_out/clock_${v}_$num
    rm -rf $out; mkdir -p $out
    export FDW_LIB=$R/ablate/libfdwave_$v.so PIPE_NUMERICS=$num PIPE_CHUNKS=173
    ( cd /tmp && timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $out --output-format csv -- python3 $R/scripts/probe_pipe.py 8192 > $out/log.txt 2>&1 )
    python3 - "$out" "$v" "$num" <<'PY'
import csv, glob, re, sys
out, v, num = sys.argv[1:4]
vals = [float(r["Counter_Value"]) for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
        if "fdw_stepn" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
us = [float(m.group(1)) for m in re.finditer(r"([0-9.]+) us/step", open(out + "/log.txt").read())]
if vals and us:
    cyc = sum(vals) / len(vals) / 8.0
    print(f"numerics={num} {v:8s} {us[0]:7.2f} us/step  {cyc / 1e3:8.1f} k shader cycles per launch (4 steps)  ->  {cyc / (4.0 * us[0]) / 1e3:.2f} GHz while the kernel runs", flush=True)
else:
    print(f"numerics={num} {v}: no data ({len(vals)} counter rows, {len(us)} timings)", flush=True)
PY
  done
done
