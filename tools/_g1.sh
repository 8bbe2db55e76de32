set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_fuzz.py -x -q -m gpu -k "key_switch" > $O/t1.log 2>&1 || { tail -30 $O/t1.log; exit 1; }
tail -1 $O/t1.log
PF_CONFIG=5 timeout -k 10 120 python tools/run_kernel.py keyswitch 3 256 > $O/ks_split.txt 2>&1; tail -1 $O/ks_split.txt
cd /tmp && export TMPDIR=/tmp
PF_CONFIG=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ks -- python3 $R/tools/run_kernel.py keyswitch 2 256 > $O/ks_prof.txt 2>&1
cd $R && python3 tools/prof_summary.py $(find $O/prof_ks -name "*kernel_stats.csv" | head -1) $O/ks_kernel_stats.txt
head -6 $O/ks_kernel_stats.txt
