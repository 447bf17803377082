# copy the judged summaries of an evidence run into profiles/r4
E=$1; P=/root/repo/profiles/r4; mkdir -p $P
for f in bench.json bench_fp32.json bench_bf16x3.json bench_threshold.json bench_pipeline_r3.json bench_kernel_stats.csv bench_kernel_stats_pipelined.csv bench_kernel_stats_bf16x3.csv agg_live.jsonl conv_layers.txt conv_layers_bf16x3.txt frame_report.txt frame_report_bf16x3.txt kernels.jsonl agg_phase_times.txt aggregate_pmc.json aggregate_sq_counters.csv pytest_gpu.txt smoke.txt stage_times_bf16.txt stage_times_bf16x3.txt; do cp $E/$f $P/ 2>/dev/null || echo "missing $f"; done
for c in FETCH_SIZE WRITE_SIZE tcc; do f=$(find $E/pmc_$c -name '*.agg' | head -1); [ -n "$f" ] && cp $f $P/pmc_${c}_aggregate_rows.csv; done
for m in fp32 bf16 bf16x3 bf16_fp32dec bf16_fp32val; do cp /root/repo/gpurun_out/parity_full_$m.json $P/ 2>/dev/null; done
ls $P | wc -l
