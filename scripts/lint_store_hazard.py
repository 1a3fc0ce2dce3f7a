#!/usr/bin/env python3
"""Static check of the BUILT libfdwave.so for the gfx950 store hazard of csrc/fdw_device.h (f4_store_arr): a buffer_store_dwordx3/x4 whose
soffset is an SGPR must not be followed, within two wait states, by an instruction that writes one of its data VGPRs (hipcc pads that pair
only for stores without a register soffset; the hardware was seen to store the NEW value).

    python3 scripts/lint_store_hazard.py [path/to/libfdwave.so]        exit code 1 and one line per finding if any

Disassembles every gfx950 code object embedded in the library (llvm-objcopy, clang-offload-bundler, llvm-objdump of the ROCm install), so
it checks what actually runs, inline asm included.  No GPU needed."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def vregs(tok):
    """'v[40:43]' -> {40..43}, 'v7' -> {7}, anything else -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def written_vgprs(mnem, ops):
    """VGPRs an instruction writes (destination operand first in this ISA; stores, compares into SGPRs and the like write none)."""
    if not ops or mnem.startswith(("buffer_store", "global_store", "flat_store", "ds_write", "s_", "v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    if mnem.startswith(("v_", "ds_read", "ds_bpermute", "ds_permute", "buffer_load", "global_load", "flat_load")):
        return vregs(ops[0])
    return set()


def disassemble(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fatbin.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        if not starts:
            sys.exit(f"{lib}: no offload bundle found")
        for i, a in enumerate(starts):
            piece = os.path.join(td, f"bundle{i}.bin")
            open(piece, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(td, f"code{i}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={piece}", f"--output={co}"],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            yield i, subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout


def check(lib):
    findings, nstores, nkernels = [], 0, 0
    for unit, text in disassemble(lib):
        func = "?"
        insts = []
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                func = m.group(1)
                nkernels += 1
                insts.append(None)                    # no fall-through bookkeeping across symbols
                continue
            m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
            if m:
                ops = [o.strip() for o in m.group(2).split(",")] if m.group(2) else []
                insts.append((func, m.group(1), ops, line.strip()))
        for i, ins in enumerate(insts):
            if ins is None or not re.match(r"buffer_store_dwordx[34]$", ins[1]):
                continue
            ops = ins[2]
            # buffer_store_dwordx4 vdata, voffset|off, srsrc, soffset [offen ...]
            soff = ops[3].split()[0] if len(ops) > 3 else "0"
            if not re.fullmatch(r"s\d+|m0|vcc_lo|vcc_hi|ttmp\d+", soff):
                continue
            nstores += 1
            data = vregs(ops[0])
            states = 0
            for nxt in insts[i + 1:i + 4]:
                if nxt is None or states >= 2:
                    break
                if nxt[1] == "s_nop":
                    states += int(nxt[2][0], 0) + 1 if nxt[2] else 1
                    continue
                hit = written_vgprs(nxt[1], nxt[2]) & data
                if hit and nxt[1].startswith("v_"):
                    findings.append(f"code object {unit}, {ins[0]}: '{ins[3].split('//')[0].strip()}' then '{nxt[3].split('//')[0].strip()}' after {states} wait state(s)")
                    break
                states += 1
    return findings, nstores, nkernels


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel_finite_difference_computation_amd", "libfdwave.so")
    f, ns, nk = check(lib)
    print(f"{os.path.basename(lib)}: {nk} symbols, {ns} x3/x4 buffer stores with a register soffset, {len(f)} within two wait states of a VALU write of their data")
    for line in f:
        print("  " + line)
    sys.exit(1 if f else 0)
