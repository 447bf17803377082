# round-3 GPU call 6: variant 11 after the permlane fix, stage times for the multi-GPU model
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c6; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sampling_gpu.py -q 2>&1 | tail -6 | tee $O/pytest_sampling.txt
timeout 300 python tools/bench_kernels.py --iters 50 --out $O/kernels.jsonl > /dev/null 2>&1; grep -v '"v3' $O/kernels.jsonl | cut -c1-120
timeout 300 python tools/stage_times.py bf16 2>&1 | tail -22 | tee $O/stage_times_bf16.txt
timeout 300 python tools/stage_times.py bf16x3 2>&1 | tail -22 | tee $O/stage_times_bf16x3.txt
