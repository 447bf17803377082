# round-3 GPU call 5: aggregation variant 11 (A/B + phase stamps), query-sharded decoder, threshold-mode bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sampling_gpu.py -q 2>&1 | tail -6 | tee $O/pytest_sampling.txt
timeout 300 python tools/bench_kernels.py --iters 50 --out $O/kernels.jsonl > /dev/null 2>&1; grep -v '"v3' $O/kernels.jsonl | cut -c1-120
timeout 300 python tools/agg_phase_times.py 1544 7 > $O/agg_phase_v7.txt 2>&1; tail -13 $O/agg_phase_v7.txt
timeout 300 python tools/agg_phase_times.py 1544 11 > $O/agg_phase_v11.txt 2>&1; tail -13 $O/agg_phase_v11.txt
timeout 1500 python -m pytest tests/test_dist_gpu.py -q 2>&1 | tail -30 | tee $O/pytest_dist.txt
timeout 300 python bench.py --proposals threshold --capacity 4096 --steps 30 --no-cpu-baseline 2>$O/bench_thr.err | tail -1 > $O/bench_thr4096.json; python -c "import json;j=json.load(open('$O/bench_thr4096.json'));print(j['value'], j['config']['workload']);print(j['protocol']['sync_per_frame']['mean_ms'])"; tail -3 $O/bench_thr.err
