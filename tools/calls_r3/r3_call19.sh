R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_sums_gpu.py -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_norm_gpu.py tests/test_pair_gpu.py -q -x 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-in-tolerance 2>/dev/null | tail -1 | cut -c1-250
cd tools && timeout 100 python bench_ese.py 2>&1 | tail -5
