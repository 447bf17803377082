# One GPU call that regenerates everything profiles/ cites: full GPU test suite, smoke, bench (+CPU baseline), rocprofv3
# kernel stats of the bench, per-kernel micro-benchmarks, and the PMC traffic pass for the aggregation kernel.
R=$GRAFT_REPO_ROOT; TAG=${1:-final}; O=$R/gpurun_out/ev_$TAG; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --steps 30 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-250 $O/bench.json
python tools/bench_kernels.py --iters 50 --out $O/kernels.jsonl > /dev/null 2>&1; cut -c1-160 $O/kernels.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_bench.log 2>&1
python $R/tools/layer_report.py $O/prof_bench/run_kernel_trace.csv v > $O/conv_layers.txt 2>&1; tail -9 $O/conv_layers.txt
N_LAUNCH=20 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_agg -o run -- python $R/tools/run_agg_once.py --camsort > $O/prof_agg.log 2>&1
# PMC: one derived counter per pass (FETCH_SIZE + WRITE_SIZE together exceed the hardware), every pass under `timeout`:
# a failed rocprofv3 does not exit on its own
if [ -n "$PMC" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    N_LAUNCH=20 timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_agg_$c -o run -- python $R/tools/run_agg_once.py --camsort > $O/pmc_agg_$c.log 2>&1 || echo "pmc pass $c failed"
  done
  N_LAUNCH=20 timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_agg_tcc -o run -- python $R/tools/run_agg_once.py --camsort > $O/pmc_agg_tcc.log 2>&1 || echo "pmc pass tcc failed"
  python $R/tools/pmc_report.py $O/pmc_agg_FETCH_SIZE $O/pmc_agg_WRITE_SIZE $O/pmc_agg_tcc | tee $O/pmc_agg.txt
fi
grep aggregate $O/prof_agg/run_kernel_stats.csv | cut -c1-200
