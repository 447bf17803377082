# round-3 GPU call 1: pair-storage kernels -- correctness, regression of the refactored bf16 kernels, tile sweep, first bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_pair_gpu.py -q -x 2>&1 | tail -25 | tee $O/pytest_pair.txt
timeout 900 python -m pytest tests/test_igemm_gpu.py tests/test_attn_norm_gpu.py tests/test_frontend_gpu.py -q 2>&1 | tail -8 | tee $O/pytest_regress.txt
MODE=pair timeout 900 python tools/tune_conv.py > $O/tune_pair.log 2>&1; tail -50 $O/tune_pair.log
cp gpurun_out/tuning_mi355x_pair.json far3d_amd/data/ 2>/dev/null
timeout 600 python bench.py --precision bf16x3 --steps 30 --no-cpu-baseline 2>$O/bench_bf16x3.err | tail -1 > $O/bench_bf16x3.json; cut -c1-400 $O/bench_bf16x3.json; tail -5 $O/bench_bf16x3.err
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x 2>&1 | tail -8 | tee $O/pytest_engine.txt
