# round-3 GPU call 8: aggregation variant 13 (128-token patches)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c8; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sampling_gpu.py -q 2>&1 | tail -6 | tee $O/pytest_sampling.txt
timeout 300 python tools/bench_kernels.py --iters 50 --out $O/kernels.jsonl > /dev/null 2>&1; grep -v '"v3' $O/kernels.jsonl | cut -c1-120
timeout 300 python tools/agg_phase_times.py 1544 13 > $O/agg_phase_v13.txt 2>&1; tail -13 $O/agg_phase_v13.txt
