#!/usr/bin/env python3
"""Copies what scripts/profile_bench.py left under gpurun_out/prof_<tag>/ into profiles/ (rNN_pmc_<tag>.txt = summary.txt, rNN_kernel_stats_<tag>.csv)
and rebuilds profiles/traffic.json from the traffic_entry.json of every tag given -- entries of other workload/size keys are kept.
    python3 scripts/collect_profiles.py r03 fwd8192 fwd8192fast ..."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tags = sys.argv[1], sys.argv[2:]
tj = os.path.join(ROOT, "profiles", "traffic.json")
entries = json.load(open(tj)) if os.path.exists(tj) else []
for tag in tags:
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    if not os.path.exists(os.path.join(d, "summary.txt")):
        print(f"{tag}: no summary.txt, skipped")
        continue
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", f"{rnd}_pmc_{tag}.txt"))
    if os.path.exists(os.path.join(d, "kernel_stats.csv")):
        shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_{tag}.csv"))
    te = os.path.join(d, "traffic_entry.json")
    if os.path.exists(te):
        e = json.load(open(te))
        e["source"] = e["sq_source"] = f"profiles/{rnd}_pmc_{tag}.txt"
        entries = [x for x in entries if not (x.get("workload") == e["workload"] and x.get("size") == e["size"] and x.get("steps_per_launch") == e["steps_per_launch"])]
        entries.append(e)
        print(f"{tag}: {e['workload']} {e['size']}: {e.get('hbm_bytes_per_launch', 0) / 1e6:.1f} MB per launch, kernel-trace average {e.get('kernel_trace_avg_us')} us, VALU busy {e.get('valu_busy')}")
json.dump(entries, open(tj, "w"), indent=1)
