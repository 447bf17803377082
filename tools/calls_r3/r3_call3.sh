# round-3 GPU call 3: fixed-capacity threshold mode, new ShardedFrame (pipeline / coalesced exchange / capacity), e2e data path, tightened goldens
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_capacity_gpu.py tests/test_data_path_gpu.py tests/test_engine_gpu.py tests/test_post_gpu.py tests/test_attn_norm_gpu.py tests/test_sampling_gpu.py tests/test_frontend_gpu.py tests/test_plugin_gpu.py tests/test_plugin_modules_gpu.py -q -s 2>&1 | grep -v "^$" | tail -60 | tee $O/pytest_a.txt
timeout 1200 python -m pytest tests/test_dist_gpu.py -q 2>&1 | tail -15 | tee $O/pytest_dist.txt
timeout 600 python bench.py --proposals threshold --capacity 1024 --steps 30 --no-cpu-baseline 2>$O/bench_thr.err | tail -1 > $O/bench_thr.json; cut -c1-900 $O/bench_thr.json; tail -3 $O/bench_thr.err
