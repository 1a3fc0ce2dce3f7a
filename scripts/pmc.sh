#!/bin/bash
# usage: scripts/pmc.sh <outdir> <python script args...>   -- runs PMC passes (each its own rocprofv3 run)
out=$1; shift
mkdir -p $out
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_VMEM_WR_TA_DATA_FIFO_FULL
TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY
TCP_TCC_READ_REQ TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TA_TA_BUSY
TCP_GATE_EN1 TCP_TA_TCP_STATE_READ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES
FETCH_SIZE TCC_HIT
WRITE_SIZE TCC_MISS TCC_REQ
TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL
GRBM_GUI_ACTIVE GRBM_TA_BUSY
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
