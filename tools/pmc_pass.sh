#!/bin/bash
# One rocprofv3 PMC pass (counters only, csv) for one kernel run; usage: tools/pmc_pass.sh <tag> "<counters>" <run_kernel args...>
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/tools/run_kernel.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-40:]
    if not ("pf::" in r["Kernel_Name"]): continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()}, "launches", max(len(v) for v in d.values()))
PY
