#!/usr/bin/env python3
"""Profiles one bench.py command on the GPU box and writes everything the bench line's `roofline` object quotes.

    python3 scripts/profile_bench.py <tag> [--workload-key forward] -- <bench.py arguments>

Runs, each as its own rocprofv3 process (counters never together with tracing; one counter group per pass):
    1. rocprofv3 --kernel-trace --stats          -> gpurun_out/prof_<tag>/kernel_stats.csv   (per-kernel calls / average / min / max)
    2. rocprofv3 --pmc FETCH_SIZE ...            -> HBM-side read bytes  (x 2 on gfx950, MI355X_MICROARCH.md "HBM")
    3. rocprofv3 --pmc WRITE_SIZE ...            -> HBM-side write bytes
    4. rocprofv3 --pmc SQ_* (two passes) + GRBM  -> VALU busy, VALU / SALU / LDS / VMEM instructions per dispatch
and then gpurun_out/prof_<tag>/summary.txt and traffic_entry.json (an entry of profiles/traffic.json, stamped with the hash of the kernel
sources so that bench.py only quotes it for the build it was taken on).  This script never touches the GPU itself.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PMC_PASSES = [
    "FETCH_SIZE TCC_EA0_RDREQ",
    "WRITE_SIZE TCC_EA0_WRREQ",
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU",
    "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA",
    "GRBM_GUI_ACTIVE GRBM_COUNT",
]


def main():
    if "--" not in sys.argv or len(sys.argv) < 3:
        sys.exit(__doc__)
    cut = sys.argv.index("--")
    head, bench_args = sys.argv[1:cut], sys.argv[cut + 1:]
    tag = head[0]
    wkey = head[head.index("--workload-key") + 1] if "--workload-key" in head else "forward"
    out = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    os.makedirs(out, exist_ok=True)
    cmd_tail = ["--", "python3", os.path.join(ROOT, "bench.py")] + bench_args + ["--no-cpu-baseline", "--no-rest-line", "--no-fast-line", "--no-extra"]
    env = dict(os.environ, TMPDIR="/tmp")

    def run(name, args):
        log = open(os.path.join(out, name + ".log"), "w")
        r = subprocess.run(["rocprofv3"] + args + ["-d", os.path.join(out, name), "--output-format", "csv"] + cmd_tail, stdout=log, stderr=subprocess.STDOUT, cwd="/tmp", env=env)
        print(f"[profile_bench] {name}: rc {r.returncode}", flush=True)
        return r.returncode

    run("trace", ["--kernel-trace", "--stats"])
    for i, ctrs in enumerate(PMC_PASSES):
        run(f"pmc{i}", ["--pmc"] + ctrs.split())

    # ---- kernel stats ----
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    kernels = {}
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(out, "kernel_stats.csv"), "w") as fo:
            fo.write(open(stats[0]).read())
        for r in rows:
            kernels[r["Name"]] = r
    # ---- counters: per-dispatch averages per kernel ----
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "fdw" not in row["Kernel_Name"]:
                continue
            a = agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    bench_line = None
    for ln in open(os.path.join(out, "trace.log")):
        if ln.startswith("{") and '"metric"' in ln:
            bench_line = json.loads(ln)
    from bench import kernel_source_hash
    lines = [f"command: python3 bench.py {' '.join(bench_args)} --no-cpu-baseline --no-rest-line   (kernel sources {kernel_source_hash(wkey)})", ""]
    entry = None
    for kname in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0, 1])[0]):
        c = {k: v[0] / v[1] for k, v in agg[kname].items()}
        n = max(v[1] for v in agg[kname].values())
        st = next((r for nme, r in kernels.items() if nme.split("(")[0] == kname), None)
        lines.append(f"kernel {kname}   (per-dispatch averages over {n} dispatches)")
        if st:
            lines.append(f"  kernel-trace: calls {st['Calls']}  average {float(st['AverageNs']) / 1e3:.2f} us  min {float(st['MinNs']) / 1e3:.2f} us  max {float(st['MaxNs']) / 1e3:.2f} us  share {st['Percentage']} %")
        for k in sorted(c):
            lines.append(f"  {k:28s} {c[k]:18.1f}")
        rd = c.get("FETCH_SIZE", 0.0) * 1024 * 2          # KiB; a 128-B request of a wide coalesced stream is tallied as 64 B on gfx950
        wr = c.get("WRITE_SIZE", 0.0) * 1024
        d = {}
        if rd or wr:
            lines.append(f"  HBM-side traffic: read FETCH_SIZE x 1024 x 2 = {rd / 1e6:.1f} MB, write WRITE_SIZE x 1024 = {wr / 1e6:.1f} MB, total {(rd + wr) / 1e6:.1f} MB per dispatch")
            d.update(hbm_bytes_per_launch=int(rd + wr), read_bytes=int(rd), write_bytes=int(wr))
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU"):
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0                                    # the counter sums the 8 XCDs
            busy = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc                # quad-cycles x 4 / (256 CUs x 4 SIMDs) / shader cycles
            lines.append(f"  VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) = {busy:.3f}; shader clock while active "
                         f"{cyc / (float(st['AverageNs']) if st else 1) :.2f} GHz" if st else f"  VALU busy {busy:.3f}")
            d.update(valu_busy=round(busy, 3))
        if c.get("SQ_INSTS_VALU") and c.get("SQ_INSTS_SALU"):
            d.update(salu_per_valu=round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3), valu_insts=int(c["SQ_INSTS_VALU"]), salu_insts=int(c["SQ_INSTS_SALU"]),
                     lds_insts=int(c.get("SQ_INSTS_LDS", 0)), vmem_insts=int(c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)))
        lines.append("")
        if entry is None and d and bench_line is not None:       # the dominant kernel of the command
            entry = dict(workload=wkey, size=bench_line["config"]["grid"][0], steps_per_launch=bench_line.get("roofline", {}).get("steps_per_launch", 1),
                         kernel=kname, source_hash=kernel_source_hash(wkey), source=f"profiles/r03_pmc_{tag}.txt", sq_source=f"profiles/r03_pmc_{tag}.txt", **d)
            if st:
                entry["kernel_trace_avg_us"] = round(float(st["AverageNs"]) / 1e3, 2)
    if bench_line is not None:
        lines.append("bench line of the traced run: " + json.dumps(bench_line))
    open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
    if entry is not None:
        json.dump(entry, open(os.path.join(out, "traffic_entry.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
