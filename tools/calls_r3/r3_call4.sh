# round-3 GPU call 4: capacity-mode fixes, aggregation kernel variants 8/9/10 (correctness + timing), extra conv tiles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sampling_gpu.py -q -x 2>&1 | tail -8 | tee $O/pytest_sampling.txt
timeout 300 python tools/bench_kernels.py --iters 50 --out $O/kernels.jsonl > /dev/null 2>&1; cut -c1-170 $O/kernels.jsonl
timeout 900 python -m pytest tests/test_capacity_gpu.py -q 2>&1 | tail -12 | tee $O/pytest_capacity.txt
timeout 900 python -m pytest tests/test_dist_gpu.py -q -k "48" 2>&1 | tail -30 | tee $O/pytest_dist.txt
timeout 300 python bench.py --proposals threshold --capacity 4096 --steps 30 --no-cpu-baseline 2>$O/bench_thr.err | tail -1 > $O/bench_thr4096.json; cut -c1-300 $O/bench_thr4096.json; python -c "import json;j=json.load(open('$O/bench_thr4096.json'));print(j['config']['workload']);print(j['protocol']['sync_per_frame']['mean_ms'])"
ONLY_K=3 EXTRA_TILES=51,53,55,57,58,59,62,66,67,91,94,95,97 timeout 600 python tools/tune_conv.py > $O/tune_bf16_extra.log 2>&1; tail -30 $O/tune_bf16_extra.log | cut -c1-400
