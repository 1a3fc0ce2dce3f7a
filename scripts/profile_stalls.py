#!/usr/bin/env python3
"""Stall and latency counters of one bench.py command on the GPU box (a companion of profile_bench.py: that one collects the bytes and the
issue counters the bench line quotes, this one asks WHERE the waves wait):

    python3 scripts/profile_stalls.py <tag> -- <bench.py arguments>          -> gpurun_out/stalls_<tag>/summary.txt

One rocprofv3 --pmc process per counter group (never together with tracing); per-dispatch averages of the dominant fdw kernel, and the
ratios that mean something: average L1->L2 read / write latency (TCP_TCC_*_REQ_LATENCY / requests), average VMEM and LDS instructions in
flight per wave-cycle, requests outstanding at the memory interface (TCC_EA0_*REQ_LEVEL / requests = cycles one request stays there),
share of cycles the texture path holds up the address FIFO, LDS bank conflicts.  A pass whose counters this profiler build does not
accept together is reported and skipped.  This script never touches the GPU itself."""
import collections
import csv
import glob
import os
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = [
    "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY",
    "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS",
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL",
    "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR",
    "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ",
    "TCP_PENDING_STALL_CYCLES TCP_TCR_RDRET_STALL TCP_TCP_TA_DATA_STALL_CYCLES TCP_GATE_EN1",
    "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS",
    "TA_BUSY",
    "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ",
    "TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_BUSY TCC_CYCLE",
    "TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL",
    "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_SRC_FIFO_FULL TCC_LATENCY_FIFO_FULL",
    "GRBM_GUI_ACTIVE GRBM_COUNT",
    "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL",
]


def main():
    if "--" not in sys.argv or len(sys.argv) < 3:
        sys.exit(__doc__)
    cut = sys.argv.index("--")
    tag, bench_args = sys.argv[1], sys.argv[cut + 1:]
    out = os.path.join(ROOT, "gpurun_out", f"stalls_{tag}")
    os.makedirs(out, exist_ok=True)
    tail = ["--", "python3", os.path.join(ROOT, "bench.py")] + bench_args + ["--no-cpu-baseline", "--no-rest-line", "--no-fast-line", "--no-extra"]
    env = dict(os.environ, TMPDIR="/tmp")
    skipped = []
    only = [int(x) for x in os.environ["FDW_STALL_PASSES"].split(",")] if os.environ.get("FDW_STALL_PASSES") else None      # e.g. FDW_STALL_PASSES=0,13
    for i, ctrs in enumerate(PASSES):
        if only is not None and i not in only:
            continue
        log = open(os.path.join(out, f"pmc{i}.log"), "w")
        # (a counter group the hardware cannot collect together makes rocprofv3 abort and then sit in its signal handler: bounded, and ended by PID)
        child = subprocess.Popen(["rocprofv3", "--pmc"] + ctrs.split() + ["-d", os.path.join(out, f"pmc{i}"), "--output-format", "csv"] + tail,
                                 stdout=log, stderr=subprocess.STDOUT, cwd="/tmp", env=env, start_new_session=True)
        try:
            rc = child.wait(timeout=float(os.environ.get("FDW_PROFILE_PASS_TIMEOUT", "150")))
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except OSError:
                pass
            child.wait()
            rc = -9
        print(f"[profile_stalls] pass {i} ({ctrs}): rc {rc}", flush=True)
        if rc != 0:
            skipped.append(ctrs)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "fdw" not in row["Kernel_Name"]:
                continue
            a = agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    lines = [f"command: python3 bench.py {' '.join(bench_args)} --no-cpu-baseline --no-rest-line --no-fast-line --no-extra   (scripts/profile_stalls.py)", ""]
    for kname in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0, 1])[0])[:2]:
        c = {k: v[0] / v[1] for k, v in agg[kname].items()}
        n = max(v[1] for v in agg[kname].values())
        lines.append(f"kernel {kname}   (per-dispatch averages over {n} dispatches)")
        for k in sorted(c):
            lines.append(f"  {k:36s} {c[k]:18.1f}")

        def ratio(label, num, den, scale=1.0, unit=""):
            if c.get(den):
                lines.append(f"  -> {label}: {scale * c.get(num, 0.0) / c[den]:.3f}{unit}")
        ratio("share of wave-cycles waiting on a counter (s_waitcnt)", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
        ratio("share of wave-cycles waiting to issue an LDS instruction", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")
        ratio("share of wave-cycles waiting for anything", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
        ratio("average time a VMEM instruction is in flight (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM)", "SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM", unit=" cycles")
        ratio("average time an LDS instruction is in flight", "SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", unit=" cycles")
        ratio("LDS bank-conflict cycles per active LDS cycle", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
        ratio("L1 -> L2 read latency (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ)", "TCP_TCC_READ_REQ_LATENCY", "TCP_TCC_READ_REQ", unit=" cycles")
        ratio("L1 -> L2 write latency", "TCP_TCC_WRITE_REQ_LATENCY", "TCP_TCC_WRITE_REQ", unit=" cycles")
        ratio("time a read stays at the memory interface (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ)", "TCC_EA0_RDREQ_LEVEL", "TCC_EA0_RDREQ", unit=" cycles")
        ratio("time a write stays at the memory interface", "TCC_EA0_WRREQ_LEVEL", "TCC_EA0_WRREQ", unit=" cycles")
        ratio("L2 hit rate", "TCC_HIT", "TCC_REQ")
        ratio("instruction-cache miss rate (SQC_ICACHE_MISSES / SQC_ICACHE_REQ)", "SQC_ICACHE_MISSES", "SQC_ICACHE_REQ")
        ratio("average time an instruction fetch is in flight (SQ_IFETCH_LEVEL / SQ_IFETCH)", "SQ_IFETCH_LEVEL", "SQ_IFETCH", unit=" cycles")
        ratio("L2 cycles stalled on the write interface per L2 cycle", "TCC_EA0_WRREQ_STALL", "TCC_CYCLE")
        ratio("texture-address cycles stalled by the cache per busy cycle", "TA_ADDR_STALLED_BY_TC_CYCLES", "TA_BUSY")
        ratio("texture-data cycles stalled by the cache per busy cycle", "TA_DATA_STALLED_BY_TC_CYCLES", "TA_BUSY")
        lines.append("")
    if skipped:
        lines.append("passes the profiler refused: " + "; ".join(skipped))
    for ln in (open(os.path.join(out, "pmc0.log")) if os.path.exists(os.path.join(out, "pmc0.log")) else ()):
        if ln.startswith("{") and '"metric"' in ln:
            lines.append("bench line of the first pass: " + ln.strip()[:400] + " ...")
    open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
