#!/bin/bash
# rocprofv3 kernel trace of bench.py; prints the kernel sequence of one timed step and the per-kernel totals.
# usage (on the GPU box): tools/step_trace.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/prof_${tag}_kernel_stats.txt
cat gpurun_out/prof_${tag}_kernel_stats.txt
python3 - $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_rows_prep" in r["Kernel_Name"] or "k_row_norms" in r["Kernel_Name"]]
i0, i1 = idx[6], idx[7]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    print("%9.1f us +%8.1f us  grid %8s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                  r["Grid_Size_X"], r["Kernel_Name"][:80]))
PY
