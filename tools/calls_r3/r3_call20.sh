R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_attn_norm_gpu.py tests/test_pair_gpu.py tests/test_dist_gpu.py tests/test_capacity_gpu.py -q -x 2>&1 | tail -4
timeout 300 python bench.py --precision bf16x3 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-250
