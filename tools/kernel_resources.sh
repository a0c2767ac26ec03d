#!/bin/bash
# Prints VGPRs / scratch / occupancy / spills of every kernel of libpt_hip.so (compile only, no GPU needed).
# Extra arguments go to hipcc (e.g. -DSOMETHING=1).
cd "$(dirname "$0")/../cuda-path-tracer_amd" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../include -Icsrc -Ihost "$@" \
    -c -o /dev/null csrc/pt_hip.hip -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur, row = None, {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur, row = m.group(1), {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w /\[\]]*): (\d+)", line)
    if m and cur:
        row[m.group(1).strip()] = int(m.group(2))
        if m.group(1).startswith("LDS Size"):
            name = re.sub(r"^_ZN3ptd(12_GLOBAL__N_1)?\d+", "", cur)[:34]
            print("%-36s VGPR %3d AGPR %3d SGPR %3d scratch %3d occ %d sgpr_spill %3d vgpr_spill %3d" % (
                name, row.get("VGPRs", -1), row.get("AGPRs", -1), row.get("TotalSGPRs", -1),
                row.get("ScratchSize [bytes/lane]", -1), row.get("Occupancy [waves/SIMD]", -1),
                row.get("SGPRs Spill", -1), row.get("VGPRs Spill", -1)))
'
