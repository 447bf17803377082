# final check of the round at the final tree: the GPU suite without the multi-process sharding file (tests/test_dist_gpu.py ran on this
# kernel code in call 20: profiles/r3/pytest_dist_call20.txt), smoke, the default bench line and the bf16x3 line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev_r3d; mkdir -p $O; cd $R
timeout 420 python -m pytest tests -m gpu -q -x --ignore=tests/test_dist_gpu.py 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 30 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 110 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-200 $O/bench.json
timeout 50 python bench.py --precision bf16x3 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bf16x3_nocpu.json; cut -c1-200 $O/bench_bf16x3_nocpu.json
