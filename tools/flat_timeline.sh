#!/bin/bash
# Kernel timeline (last repetition) of one flat search: tools/flat_timeline.sh <nq>
nq=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ft_$nq -- python3 tools/run_kernel.py flat 3 $nq > gpurun_out/ft_$nq.log 2>&1 || exit 1
python3 - $nq <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("gpurun_out/ft_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if "pf::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
rows = rows[-(len(rows) // 3):]
t0 = rows[0][0]
print("nq", sys.argv[1], "total us", (rows[-1][1] - t0) / 1e3)
for s, e, n, g in rows:
    print("  +%.1f  dur %.1f  %s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, g))
PY
