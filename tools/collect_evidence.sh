# copy the judged summaries of an evidence run (gpurun_out/ev_<tag>) into profiles/<round>:  bash tools/collect_evidence.sh gpurun_out/ev_r6 r6
E=$1; P=/root/repo/profiles/${2:-r6}; mkdir -p $P
for f in bench.json bench_fp32.json bench_bf16.json bench_threshold.json \
         bench_kernel_stats.csv bench_kernel_stats_bf16.csv conv_layers.txt conv_layers_bf16.txt frame_report.txt \
         frame_report_bf16.txt kernels.jsonl agg_phase_times.txt aggregate_pmc.json aggregate_pmc_fp32rows.json aggregate_sq_counters.csv pytest_gpu.txt \
         smoke.txt stage_times_bf16.txt stage_times_bf16x3.txt clock_probe.txt conv_phase_times.txt topk_phase_times_final.txt ws_ab_pair.txt \
         ws_conv_phase_times.txt f32x_gemm_ab.txt attn_f32_ab.txt agg_phase_times_unsorted.txt agg_sorted_ab.txt agg_launch_vs_queries.txt dispatch_ramp.txt; do cp $E/$f $P/ 2>/dev/null || echo "missing $f"; done
for c in FETCH_SIZE WRITE_SIZE tcc; do
  f=$(find $E/pmc_$c -name '*.agg' | head -1); [ -n "$f" ] && cp $f $P/pmc_${c}_aggregate_rows.csv
  f=$(find $E/pmcx3_$c -name '*.agg' | head -1); [ -n "$f" ] && cp $f $P/pmcx3_${c}_aggregate_rows.csv
done
for m in fp32 bf16 bf16x3 bf16_fp32dec bf16_fp32val unadopted; do cp /root/repo/gpurun_out/parity_full_$m.json $P/ 2>/dev/null; done
sed -i '/amdgpu.ids: No such file/d' $P/*.txt 2>/dev/null
ls $P | wc -l
