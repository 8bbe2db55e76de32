R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
PREFHETCH_HIP_LIB=$R/build_variants/libpf_cap4k.so timeout -k 10 600 python -m pytest tests/test_gpu_flat.py -x -q -m gpu > $O/t13.log 2>&1; tail -3 $O/t13.log
for i in 1 2; do
python tools/time_flat.py 30 2>&1 | grep "exact16=1"
PREFHETCH_HIP_LIB=$R/build_variants/libpf_cap4k.so python tools/time_flat.py 30 2>&1 | grep "exact16=1"
for d in 2 3 4; do echo "cap4k div=$d $(PF_FLAT_GROWTH_DIV=$d PREFHETCH_HIP_LIB=$R/build_variants/libpf_cap4k.so python tools/time_flat.py 30 2>&1 | grep 'exact16=1')"; done
done
