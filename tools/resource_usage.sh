#!/bin/bash
# Prints VGPR/AGPR/scratch/occupancy/LDS per kernel of one HIP source: tools/resource_usage.sh <file.hip> [extra flags]
src=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed "$@" \
  -Rpass-analysis=kernel-resource-usage --cuda-device-only -c "$src" -o /dev/null 2>&1 | python3 -c '
import sys,re
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        cur={"name":t.split(":",1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.split(":",1); cur[k.strip()]=v.strip()
for r in rows:
    print("%-70s vgpr %4s agpr %4s sgpr %4s scratch %5s occ %2s lds %6s" % (r["name"][:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
