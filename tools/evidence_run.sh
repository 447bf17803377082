# One GPU call that regenerates everything profiles/<round>/ cites: full GPU test suite, smoke, bench (+CPU baseline + parity),
# the fp32 and bf16x3 bench lines, the aggregation kernel's per-phase s_memtime stamps, rocprofv3 kernel stats of the bench, per-layer / per-frame reports, per-kernel micro-benchmarks and the
# IN-FRAME PMC traffic passes for the aggregation kernel.   usage: [PMC=1] [TESTS=0] bash tools/evidence_run.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-final}; O=$R/gpurun_out/ev_$TAG; mkdir -p $O
cd $R
if [ "${TESTS:-1}" = "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
fi
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
timeout 600 python bench.py --precision fp32 --steps 20 --no-cpu-baseline 2>$O/bench_fp32.err | tail -1 > $O/bench_fp32.json; cut -c1-200 $O/bench_fp32.json
timeout 600 python bench.py --precision bf16x3 --steps 50 2>$O/bench_bf16x3.err | tail -1 > $O/bench_bf16x3.json; cut -c1-200 $O/bench_bf16x3.json
timeout 300 python tools/agg_phase_times.py > $O/agg_phase_times.txt 2>&1; tail -14 $O/agg_phase_times.txt
python tools/bench_kernels.py --iters 50 --out $O/kernels.jsonl > /dev/null 2>&1; cut -c1-160 $O/kernels.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_bench.log 2>&1
python $R/tools/layer_report.py $O/prof_bench/run_kernel_trace.csv v > $O/conv_layers.txt 2>&1; tail -9 $O/conv_layers.txt
python $R/tools/frame_report.py $O/prof_bench/run_kernel_trace.csv 40 > $O/frame_report.txt 2>&1; head -12 $O/frame_report.txt
# PMC: one derived counter per pass (FETCH_SIZE + WRITE_SIZE together exceed the hardware), every pass under `timeout` (a failed
# rocprofv3 does not exit on its own); the eager bench so that every kernel is its own dispatch
if [ -n "$PMC" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o run -- python $R/bench.py --eager --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_$c.log 2>&1 || echo "pmc pass $c failed"
  done
  timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o run -- python $R/bench.py --eager --steps 6 --warmup 2 --no-cpu-baseline > $O/pmc_tcc.log 2>&1 || echo "pmc pass tcc failed"
  python $R/tools/pmc_to_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_tcc aggregate_v7_kernel $O/aggregate_pmc.json
  # keep only the aggregation rows of the (large) counter files
  for d in $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_tcc; do
    for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep aggregate $f) > $f.agg; rm -f $f; done
    find $d -name '*kernel_trace.csv' -delete
  done
fi
grep -i aggregate $O/prof_bench/run_kernel_stats.csv | cut -c1-200
