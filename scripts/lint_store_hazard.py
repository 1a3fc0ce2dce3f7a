#!/usr/bin/env python3
"""Static check of the BUILT libfdwave.so for the gfx950 store-data hazard (csrc/fdw_device.h, f4_store_arr): a 96- or 128-bit vector
memory store must not be followed, within two wait states, by a VALU instruction that writes one of its data VGPRs -- the store still
reads them (round 2 saw the NEW value stored on some launches).  hipcc's hazard recogniser pads buffer stores WITHOUT a register soffset
and flat / global stores (two wait states on gfx940+), but not buffer stores WITH one; the kernels pad those themselves (s_nop 1).  This
lint does not trust either: it checks EVERY x3/x4 store in the library -- buffer_store (any soffset), global_store, flat_store,
scratch_store -- whoever padded it.

    python3 scripts/lint_store_hazard.py [path/to/libfdwave.so]        exit code 1 and one line per finding if any

What is followed: straight-line successors AND branch targets -- an s_branch / s_cbranch_* inside the window counts as one wait state and
both the target and (for a conditional branch) the fall-through are examined, so a store at the end of a loop body followed by a VALU
write at the loop head is found.  s_setpc / s_swappc / s_endpgm end a path (nothing to follow statically).
What counts as a writer: VALU instructions (v_*), the case the hardware hazard is about.  Memory instructions that name a data VGPR as
destination (buffer_load, global_load, ds_read, ds_bpermute ...) deliver it hundreds of cycles after issue, far outside the window; they
are listed as notes, never as findings.

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


STORE = re.compile(r"(buffer|global|flat|scratch)_store_dwordx[34]$")
WINDOW = 2          # wait states the data registers stay busy (gfx940+: LLVM's VALUWaitStates for this hazard; measured in round 2)


def parse(text):
    """[(func, addr, mnemonic, operands, line)] per code object, None between symbols; addr from objdump's '// 0000000012A0:' comment."""
    insts, func = [], "?"
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func = m.group(1)
            insts.append(None)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            ops = [o.strip() for o in m.group(2).split(",")] if m.group(2) else []
            insts.append((func, int(m.group(3), 16), m.group(1), ops, line.strip()))
    return insts


def branch_target(ins):
    """Byte address an s_branch / s_cbranch_* jumps to: PC of the next instruction + 4 * simm16."""
    try:
        imm = int(ins[3][0].split()[0], 0)
    except (IndexError, ValueError):
        return None
    if imm >= 0x8000:
        imm -= 0x10000
    return ins[1] + 4 + 4 * imm


def check(lib):
    findings, notes, nstores, nkernels, nbranches = [], [], 0, 0, 0
    for unit, text in disassemble(lib):
        insts = parse(text)
        nkernels += sum(1 for x in insts if x is None)
        by_addr = {ins[1]: k for k, ins in enumerate(insts) if ins is not None}
        for i, ins in enumerate(insts):
            if ins is None or not STORE.match(ins[2]):
                continue
            nstores += 1
            data = vregs(ins[3][0]) if ins[2].startswith("buffer") else (vregs(ins[3][1]) if len(ins[3]) > 1 else set())
            if not data:
                continue
            here = ins[4].split("//")[0].strip()
            seen = set()
            todo = [(i + 1, 0)]                                   # (index of the next instruction on this path, wait states already passed)
            while todo:
                k, states = todo.pop()
                while states < WINDOW and k < len(insts) and insts[k] is not None and (k, states) not in seen:
                    seen.add((k, states))
                    nxt = insts[k]
                    mn = nxt[2]
                    if mn == "s_nop":
                        states += int(nxt[3][0], 0) + 1 if nxt[3] else 1
                        k += 1
                        continue
                    if mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                        break
                    if mn == "s_branch" or mn.startswith("s_cbranch"):
                        nbranches += 1
                        t = branch_target(nxt)
                        if t in by_addr:
                            todo.append((by_addr[t], states + 1))
                        if mn == "s_branch":
                            break
                        states += 1
                        k += 1
                        continue
                    hit = written_vgprs(mn, nxt[3]) & data
                    if hit:
                        msg = f"code object {unit}, {ins[0]}: '{here}' then '{nxt[4].split('//')[0].strip()}' after {states} wait state(s)"
                        (findings if mn.startswith("v_") else notes).append(msg)
                        break
                    states += 1
                    k += 1
    return findings, nstores, nkernels, notes, nbranches


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "parallel_finite_difference_computation_amd", "libfdwave.so")
    f, ns, nk, notes, nb = check(lib)
    print(f"{os.path.basename(lib)}: {nk} symbols, {ns} 96/128-bit vector stores (buffer / global / flat / scratch), {nb} branches followed inside a window, "
          f"{len(f)} within {WINDOW} wait states of a VALU write of their data, {len(notes)} followed by a memory instruction that names a data register (harmless, see docstring)")
    for line in f:
        print("  " + line)
    for line in notes[:20]:
        print("  note: " + line)
    sys.exit(1 if f else 0)
