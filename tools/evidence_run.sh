# One GPU call that regenerates everything profiles/<round>/ cites (round 6 form: the bench headline is the in-tolerance bf16x3 engine under
# the reference's sync-per-frame protocol, the bf16 engine rides in the same line as `fast_mode`): full GPU test suite, smoke, the default
# bench (headline + fast_mode + CPU baseline + parity + in-run PMC traffic), the fp32 / bf16 / threshold-proposal bench lines, stage times
# for the multi-GPU model, the aggregation kernel's per-phase stamps, rocprofv3 kernel stats of the bench (bf16x3 and bf16) with per-layer /
# per-frame reports, per-kernel micro-benchmarks, the persistent conv kernel's A/B and per-step stamps, and the IN-FRAME PMC traffic passes.
#   usage: [PMC=1] [TESTS=0] bash tools/evidence_run.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-final}; O=$R/gpurun_out/ev_$TAG; mkdir -p $O
cd $R
if [ "${TESTS:-1}" = "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -30 | tee $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
fi
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
timeout 600 python bench.py --precision fp32 --steps 20 --no-cpu-baseline --no-pmc --no-two-streams 2>$O/bench_fp32.err | tail -1 > $O/bench_fp32.json; cut -c1-200 $O/bench_fp32.json
timeout 600 python bench.py --precision bf16 --steps 50 --no-cpu-baseline --no-pmc --no-two-streams 2>$O/bench_bf16.err | tail -1 > $O/bench_bf16.json; cut -c1-200 $O/bench_bf16.json
timeout 600 python bench.py --proposals threshold --capacity 4096 --steps 30 --no-cpu-baseline --no-pmc 2>$O/bench_thr.err | tail -1 > $O/bench_threshold.json; cut -c1-200 $O/bench_threshold.json
timeout 300 python tools/stage_times.py bf16 > $O/stage_times_bf16.txt 2>&1; tail -16 $O/stage_times_bf16.txt
timeout 300 python tools/stage_times.py bf16x3 > $O/stage_times_bf16x3.txt 2>&1
# the aggregation kernel as the engine runs it (sorted mode: "8s") and as a stand-alone caller gets it (unsorted), the A/B of the two with
# round 4's greedy dealing beside them, the launch time against the number of queries, the dispatch ramp of a one-round launch
timeout 300 python tools/agg_phase_times.py 1544 8s 2>&1 | grep -v amdgpu.ids > $O/agg_phase_times.txt; tail -14 $O/agg_phase_times.txt
timeout 300 python tools/agg_phase_times.py 1544 8 2>&1 | grep -v amdgpu.ids > $O/agg_phase_times_unsorted.txt
timeout 200 python tools/probe/agg_sorted_ab.py 2>&1 | grep -v amdgpu.ids > $O/agg_sorted_ab.txt; cat $O/agg_sorted_ab.txt
for a in 193 772 1544 3088; do timeout 100 python tools/probe/agg_sorted_ab.py $a 2>&1 | grep -v amdgpu.ids | sed "s/^/A=$a /"; done > $O/agg_launch_vs_queries.txt
[ -x tools/ubench/_bin/dispatch_ramp ] && timeout 60 tools/ubench/_bin/dispatch_ramp 1544 > $O/dispatch_ramp.txt 2>&1
python tools/bench_kernels.py --iters 50 --out $O/kernels.jsonl > /dev/null 2>&1; cut -c1-160 $O/kernels.jsonl
[ -x tools/ubench/_bin/clock_probe ] && timeout 60 tools/ubench/_bin/clock_probe > $O/clock_probe.txt 2>&1
# the persistent wave-specialised 3x3 kernel: A/B against the shipped tiles (bitwise check included), per-step stamps of consumer wave 0
timeout 400 python tools/probe/ws_conv_ab.py pair 5 > $O/ws_ab_pair.txt 2>&1
[ -f far3d_amd/libfar3d_hip_prof.so ] && timeout 300 python tools/probe/ws_conv_prof.py > $O/ws_conv_phase_times.txt 2>&1
[ -f far3d_amd/libfar3d_hip_prof.so ] && timeout 200 python tools/conv_phase_times.py 2>/dev/null | grep tile > $O/conv_phase_times.txt
# the exact-fp32 decoder of the in-tolerance engine: GEMM tiles (482-494: K groups inside the workgroup) and the fp32 attention kernels
timeout 250 python tools/probe/f32x_gemm_ab.py 2>/dev/null | grep -v amdgpu.ids > $O/f32x_gemm_ab.txt
timeout 200 python tools/probe/attn_f32_ab.py 2>/dev/null | grep -v amdgpu.ids > $O/attn_f32_ab.txt
timeout 200 python tools/topk_phase_times.py --shipped 2>/dev/null | grep -v amdgpu.ids > $O/topk_phase_times_final.txt
cd /tmp && export TMPDIR=/tmp
# the per-kernel statistics come from frames that run ONE AT A TIME (sync-per-frame region + --no-pipeline for the back-to-back region): with
# the frame pipeline three camera stages and a head share the chip and a kernel's in-trace duration measures its co-runners as much as itself
PROF_ARGS="--no-pipeline --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-pmc --latency-groups 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o run -- python $R/bench.py $PROF_ARGS > $O/prof_bench.log 2>&1
python $R/tools/layer_report.py $O/prof_bench/run_kernel_trace.csv v > $O/conv_layers.txt 2>&1; tail -9 $O/conv_layers.txt
python $R/tools/frame_report.py $O/prof_bench/run_kernel_trace.csv 45 > $O/frame_report.txt 2>&1; head -12 $O/frame_report.txt
cp $O/prof_bench/run_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; rm -f $O/prof_bench/run_kernel_trace.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o run -- python $R/bench.py --precision bf16 $PROF_ARGS > $O/prof_bf16.log 2>&1
python $R/tools/layer_report.py $O/prof_bf16/run_kernel_trace.csv v > $O/conv_layers_bf16.txt 2>&1; tail -9 $O/conv_layers_bf16.txt
python $R/tools/frame_report.py $O/prof_bf16/run_kernel_trace.csv 40 > $O/frame_report_bf16.txt 2>&1
cp $O/prof_bf16/run_kernel_stats.csv $O/bench_kernel_stats_bf16.csv 2>/dev/null; rm -rf $O/prof_bf16
# PMC: one derived counter per pass (FETCH_SIZE + WRITE_SIZE together exceed the hardware), every pass under `timeout` (a failed
# rocprofv3 does not exit on its own); the eager bench so that every kernel is its own dispatch
if [ -n "$PMC" ]; then
  EAGER="--eager --steps 5 --warmup 2 --no-cpu-baseline --no-fast-mode --no-pmc --latency-groups 0"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcx3_$c -o run -- python $R/bench.py $EAGER > $O/pmcx3_$c.log 2>&1 || echo "pmc pass x3 $c failed"
  done
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmcx3_tcc -o run -- python $R/bench.py $EAGER > $O/pmcx3_tcc.log 2>&1 || echo "pmc pass x3 tcc failed"
  python $R/tools/pmc_to_json.py $O/pmcx3_FETCH_SIZE $O/pmcx3_WRITE_SIZE $O/pmcx3_tcc aggregate_v8_kernel $O/aggregate_pmc_fp32rows.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o run -- python $R/bench.py --precision bf16 $EAGER > $O/pmc_$c.log 2>&1 || echo "pmc pass $c failed"
  done
  timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o run -- python $R/bench.py --precision bf16 $EAGER > $O/pmc_tcc.log 2>&1 || echo "pmc pass tcc failed"
  python $R/tools/pmc_to_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_tcc aggregate_v8_kernel $O/aggregate_pmc.json
  # SQ counters of the aggregation kernel: where the wave cycles go (VALU issue vs waiting)
  timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_sq -o run -- python $R/tools/probe/run_agg_once.py > $O/pmc_sq.log 2>&1 || echo "pmc pass sq failed"
  for f in $(find $O/pmc_sq -name '*counter_collection.csv'); do (head -1 $f; grep aggregate $f) > $O/aggregate_sq_counters.csv; done
  # keep only the aggregation rows of the (large) counter files
  for d in $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_tcc $O/pmcx3_FETCH_SIZE $O/pmcx3_WRITE_SIZE $O/pmcx3_tcc; do
    for f in $(find $d -name '*counter_collection.csv'); do (head -1 $f; grep aggregate $f) > $f.agg; rm -f $f; done
    find $d -name '*kernel_trace.csv' -delete
  done
  rm -rf $O/pmc_sq
fi
grep -i aggregate $O/bench_kernel_stats.csv | cut -c1-200
