#!/bin/bash
# development aid (build container, no GPU): VGPRs / spills / scratch / occupancy of the kernels of one .hip file as the compiler reports them
#   scripts/dev/kernel_regs.sh zr_stream.hip [extra hipcc flags ...]
R=$(cd "$(dirname "$0")/../.." && pwd)
F=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++20 -O3 -fPIC -ffp-contract=fast -I$R/include "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null $R/raytracer_project_amd/csrc/$F 2>&1 |
  python3 -c '
import re, sys
name = None; row = {}
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        if name: print(name, row)
        name = m.group(1); row = {}
    for k in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPR Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(k) + r": (\d+)", l)
        if m and "remark" in l: row[k.split(" [")[0]] = int(m.group(1))
if name: print(name, row)
' | sed -e 's/_ZN2zr//' | c++filt 2>/dev/null
