# round 4, call 6: v8 (MFMA patch weights, deeper last gather): parity, live timings, natural-order A/B, engine tests, a quick bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c6; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sampling_gpu.py -q -x 2>&1 | tail -5 | tee $O/pytest_sampling.txt
NOPERM=1 timeout 300 python tools/bench_agg_live.py tools/_scratch/agg_operands.pt 7 8 2>&1 | tee $O/agg_live.jsonl
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_plugin_gpu.py tests/test_plugin_modules_gpu.py tests/test_capacity_gpu.py -q -x 2>&1 | tail -8 | tee $O/pytest_engine.txt
timeout 300 python bench.py --no-cpu-baseline --no-in-tolerance --steps 40 2>$O/bench.err | tail -1 > $O/bench_quick.json; cut -c1-300 $O/bench_quick.json; python -c "
import json; j=json.load(open('$O/bench_quick.json')); print(j['roofline'])"
