#!/bin/bash
# rocprofv3 PMC passes (counters only) over a search at one row length above 256: tools/pmc_wide.sh <tag> <d> <law> [nb]
tag=$1; d=$2; law=$3; nb=${4:-1000000}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in "a:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "b:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "m:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "c:GRBM_GUI_ACTIVE" "f:FETCH_SIZE"; do
  p=${pass%%:*}
  rocprofv3 --pmc ${pass#*:} --output-format csv -d $O/pmc_${tag}_$p -- python3 $R/tools/run_flat_wide.py $d $law $nb 1 > $O/pmc_${tag}_$p.log 2>&1
  f=$(find $O/pmc_${tag}_$p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "pf::k_l2_wide16" in r["Kernel_Name"] or "pf::k_wide_fixup" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-24:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, dd in agg.items():
    print(k, {c: "%.4g" % v for c, v in dd.items()}, "launches", max(n[(k, c)] for c in dd))
PY
done
