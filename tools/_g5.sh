R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_flat.py -x -q -m gpu > $O/t5.log 2>&1; tail -3 $O/t5.log
for i in 1 2; do python tools/time_flat.py 20 2>&1 | grep "exact16=1"; done
python tools/time_flat_gauss.py 2>&1 | tail -3
