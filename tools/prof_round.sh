#!/bin/bash
# Collects the round's measurement artefacts in ONE lease: bench line, rocprofv3 kernel stats of the same command, PMC passes
# of the roofline kernel (stall breakdown, clock, HBM traffic), shape sweeps.  usage (GPU box): tools/prof_round.sh <tag>
# A gpurun call is limited to 1200 s: the set is collected in two calls, `tools/prof_round.sh <tag> 1` (bench, kernel stats, PMC of k_ctpt and of the
# pre-filter, phase stamps, timelines) and `tools/prof_round.sh <tag> 2` (sweeps, key switch, config-5 split passes, encrypted round, PIR).
tag=$1; part=${2:-all}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
if [ "$part" != 2 ]; then
python3 $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/prof_${tag}_bench.log 2>&1
cd $R
python3 tools/prof_summary.py $(find $O/prof_${tag}_bench -name "*kernel_stats.csv" | head -1) $O/${tag}_bench_kernel_stats.txt
for pass in "a:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "b:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
            "c:GRBM_GUI_ACTIVE" "f:FETCH_SIZE" "w:WRITE_SIZE"; do
  bash tools/pmc_pass.sh ${tag}_ctpt_${pass%%:*} "${pass#*:}" ctpt 30 1024 > $O/${tag}_pmc_ctpt_${pass%%:*}.txt 2>&1
done
bash tools/pmc_flat.sh ${tag}_flat 3 1024 > $O/${tag}_pmc_flat_tiles.txt 2>&1
# phase stamps need a library built with -DPF_FLAT_STAMPS: the one carried along in build_variants/stamps when it is NEWER than the product library
# (built from the same sources, after it), otherwise built here into /tmp -- never a stale one
SL=build_variants/stamps/libprefhetch_hip.so
if [ ! $SL -nt prefhetch_amd/lib/libprefhetch_hip.so ]; then
  SL=/tmp/pf_stamps/libprefhetch_hip.so
  make -C prefhetch_amd/csrc -j16 BUILD=/tmp/pf_build_stamps OUT=$SL EXTRA="-DPF_EXPERIMENT_BUILD -DPF_FLAT_STAMPS" $SL > $O/${tag}_stamps_build.log 2>&1 || SL=
fi
[ -n "$SL" ] && PREFHETCH_HIP_LIB=$SL python3 tools/flat_stamps.py > $O/${tag}_flat_stamps.txt 2>&1
bash tools/flat_timeline.sh 1024 > $O/${tag}_flat_timeline.txt 2>&1
PF_RK_LAW=gauss bash tools/flat_timeline.sh 1024 > $O/${tag}_flat_timeline_gaussian.txt 2>&1
fi
if [ "$part" != 1 ]; then
python3 tools/time_flat_d.py 30 32 100 96 128 200 192 256 > $O/${tag}_flat_row_lengths.txt 2>&1
python3 tools/time_ctpt_small.py > $O/${tag}_ctpt_small_launches.txt 2>&1
python3 tools/time_flat_wide.py 384 512 1024 > $O/${tag}_flat_wide_rows.txt 2>&1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_wide -- python3 $R/tools/run_flat_wide.py 512 gauss > $O/prof_${tag}_wide.log 2>&1
cd $R && python3 tools/prof_summary.py $(find $O/prof_${tag}_wide -name "*kernel_stats.csv" | head -1) $O/${tag}_flat_wide_512_kernel_stats.txt
python3 tools/time_flat_gauss.py > $O/${tag}_flat_gaussian.txt 2>&1
python3 tools/sweep_shapes.py > $O/${tag}_shapes_sweep.json 2> $O/${tag}_shapes_sweep.err
python3 tools/sweep_flat.py > $O/${tag}_flat_sweep.json 2> $O/${tag}_flat_sweep.err
python3 tools/time_ivfpq.py > $O/${tag}_ivfpq.json 2> $O/${tag}_ivfpq.err
cd /tmp && PF_CONFIG=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_ks -- python3 $R/tools/run_kernel.py keyswitch 2 256 > $O/${tag}_keyswitch.txt 2>&1
cd $R && python3 tools/prof_summary.py $(find $O/prof_${tag}_ks -name "*kernel_stats.csv" | head -1) $O/${tag}_keyswitch_kernel_stats.txt
tail -3 $O/${tag}_keyswitch.txt
# PMC passes of the key switch kernels (counters only, one pass each): vector-pipe busy = SQ_ACTIVE_INST_VALU x 4 / 1024 / (GRBM_GUI_ACTIVE / 8)
for pass in "a:SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "b:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
            "c:GRBM_GUI_ACTIVE" "f:FETCH_SIZE" "w:WRITE_SIZE"; do
  PF_CONFIG=5 bash tools/pmc_pass.sh ${tag}_ks_${pass%%:*} "${pass#*:}" keyswitch 1 64 >> $O/${tag}_pmc_keyswitch.txt 2>&1
done
# config 5, the transforms' two-pass split (k_nsA / k_nsB / k_nsC): per-pass kernel stats and the vector-pipe / traffic counters
python3 tools/time_ns.py > $O/${tag}_config5_split.txt 2>&1
cd /tmp && PF_NS_SPLIT=7 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_ns -- python3 $R/tools/time_ns.py 2 > $O/${tag}_ns.txt 2>&1
cd $R && python3 tools/prof_summary.py $(find $O/prof_${tag}_ns -name "*kernel_stats.csv" | head -1) $O/${tag}_config5_split_kernel_stats.txt
: > $O/${tag}_pmc_config5_split.txt
for pass in "b:SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "c:GRBM_GUI_ACTIVE" "f:FETCH_SIZE" "w:WRITE_SIZE"; do
  pp=${pass%%:*}
  cd /tmp && PF_NS_SPLIT=7 rocprofv3 --pmc ${pass#*:} --output-format csv -d $O/pmc_${tag}_ns_$pp -- python3 $R/tools/time_ns.py 1 > $O/pmc_${tag}_ns_$pp.log 2>&1
  cd $R && python3 - "$(find $O/pmc_${tag}_ns_$pp -name '*counter_collection.csv' | head -1)" >> $O/${tag}_pmc_config5_split.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "pf::k_ns" in r["Kernel_Name"] or "pf::k_ntt" in r["Kernel_Name"] or "pf::k_ctpt" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()}, "launches", max(len(v) for v in d.values()))
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_enc -- python3 $R/tools/run_kernel.py encround 5 1024 > $O/${tag}_encround.txt 2>&1
cd $R && python3 tools/prof_summary.py $(find $O/prof_${tag}_enc -name "*kernel_stats.csv" | head -1) $O/${tag}_encround_kernel_stats.txt
tail -2 $O/${tag}_encround.txt
[ -x tools/time_pir ] && timeout -k 10 400 tools/time_pir 262144 2 > $O/${tag}_pir_262144_rows.json 2> $O/${tag}_pir.err
fi
cat $O/${tag}_bench.json | head -c 600; echo
# a step that printed a traceback produced no measurement: say so loudly and fail
bad=$(grep -l "Traceback (most recent call last)" $O/${tag}_* 2>/dev/null)
if [ -n "$bad" ]; then echo "prof_round.sh: THESE FILES HOLD A TRACEBACK, NOT A MEASUREMENT:"; echo "$bad"; exit 1; fi
