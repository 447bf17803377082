# round 4, call 5: v8 with the patch weights on the matrix pipe (v_mfma_f32_16x16x4_f32) -- parity, live timings, stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c5; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_sampling_gpu.py -q -x 2>&1 | tail -15 | tee $O/pytest_sampling.txt
timeout 300 python tools/bench_agg_live.py tools/_scratch/agg_operands.pt 7 8 9 2>&1 | tee $O/agg_live.jsonl
for l in 5; do
  timeout 120 python tools/agg_phase_times.py 1544 8 tools/_scratch/agg_operands.pt $l $O/stamps8_l$l.npz > $O/agg_phase_v8_l$l.txt 2>&1; tail -20 $O/agg_phase_v8_l$l.txt
done
