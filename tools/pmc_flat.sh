#!/bin/bash
# rocprofv3 PMC passes (counters only) over the flat-L2 pre-filter at config 3; per kernel the SUM over the launches of the run.
# usage (GPU box): tools/pmc_flat.sh <tag> [reps] [nq]
tag=$1; reps=${2:-3}; nq=${3:-1024}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in "a:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "b:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
            "m:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE" \
            "x:SQ_VALU_MFMA_COEXEC_CYCLES" \
            "t:TCC_HIT_sum TCC_MISS_sum" "c:GRBM_GUI_ACTIVE" "f:FETCH_SIZE"; do
  p=${pass%%:*}
  rocprofv3 --pmc ${pass#*:} --output-format csv -d $O/pmc_${tag}_$p -- python3 $R/tools/run_kernel.py flat $reps $nq > $O/pmc_${tag}_$p.log 2>&1
  f=$(find $O/pmc_${tag}_$p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $reps <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in rows:
    if "pf::" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0][-34:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: "%.4g" % (v / reps) for c, v in d.items()}, "launches/search", max(n[(k, c)] for c in d) / reps)
PY
done
