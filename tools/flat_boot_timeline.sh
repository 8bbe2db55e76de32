#!/bin/bash
# Durations of the first three kernels of a flat search (prep, bootstrap tiles, bootstrap selection) for the library in PREFHETCH_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fb_$tag -- python3 tools/run_kernel.py flat 3 1024 > gpurun_out/fb_$tag.log 2>&1 || exit 1
python3 - $tag <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("gpurun_out/fb_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if "pf::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
rows = rows[-(len(rows) // 3):]
print(sys.argv[1], " | ".join("%s %.1f" % (n.split("(")[0][-22:], (e - s) / 1e3) for s, e, n in rows[:3]))
PY
