#!/bin/bash
# usage: scripts/pmc_traffic.sh <outdir> <python script args...>  -- HBM-side traffic counters only (2 passes, each its own rocprofv3 run)
out=$1; shift
mkdir -p $out
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done <<'LIST'
FETCH_SIZE TCC_EA0_RDREQ TCC_EA0_RDREQ_32B
WRITE_SIZE TCC_EA0_WRREQ TCC_EA0_WRREQ_64B
LIST
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fdw_" not in row["Kernel_Name"]:
            continue
        a = agg[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open(out + "/summary.txt", "w") as fo:
    for kname in sorted(agg):
        for k in sorted(agg[kname]):
            v = agg[kname][k]
            line = f"{kname:62s} {k:24s} per-dispatch avg {v[0]/v[1]:16.1f}   (n={v[1]})"
            print(line); fo.write(line + "\n")
PY
