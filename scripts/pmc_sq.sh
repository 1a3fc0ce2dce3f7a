#!/bin/bash
# usage: scripts/pmc_sq.sh <outdir> <python script args...>  -- SQ occupancy/issue counters only (3 passes)
out=$1; shift
mkdir -p $out
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
GRBM_GUI_ACTIVE GRBM_COUNT
LIST
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fdw_" not in row["Kernel_Name"]:
            continue
        a = agg[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg):
        line = f"{k:40s} per-dispatch avg {agg[k][0]/agg[k][1]:18.1f}   (n={agg[k][1]})"
        print(line); fo.write(line + "\n")
PY
