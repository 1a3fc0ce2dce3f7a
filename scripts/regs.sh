#!/bin/bash
# usage: scripts/regs.sh <file.hip> [extra hipcc flags]   -- VGPR / spill counts of every kernel in a translation unit (development tool)
src=$1; shift
cs=/root/repo/parallel_finite_difference_computation_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -I/root/repo/include -I$cs "$@" -S --cuda-device-only -o /tmp/regs_out.s $cs/$src 2>&1 | grep -m3 error
python3 - <<'PY'
import re
t=open('/tmp/regs_out.s').read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', t, re.S):
    print(f"  vgpr {m.group(5):>3} spill {m.group(6):>3}  sgpr {m.group(3):>3} spill {m.group(4):>3}  scratch {m.group(2):>4}  {m.group(1)[:90]}")
PY
