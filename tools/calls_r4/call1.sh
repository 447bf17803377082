# round 4, call 1: live aggregation operands of a benchmark frame + per-wave phase stamps of the shipped kernel on them
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c1; mkdir -p $O; cd $R
timeout 300 python tools/dump_agg_operands.py $O/agg_operands.pt 2>&1 | tail -2
for l in 0 5; do
  timeout 120 python tools/agg_phase_times.py 1544 7 $O/agg_operands.pt $l $O/stamps_l$l.npz > $O/agg_phase_live_l$l.txt 2>&1; tail -12 $O/agg_phase_live_l$l.txt
done
timeout 300 python bench.py --no-cpu-baseline --no-in-tolerance --steps 40 2>/dev/null | tail -1 > $O/bench_quick.json; cut -c1-400 $O/bench_quick.json
