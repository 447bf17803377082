# round 4, call 4: finer stamps inside v8's build phase (live operands, layers 0 and 5)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c4; mkdir -p $O; cd $R
for l in 0 5; do
  timeout 120 python tools/agg_phase_times.py 1544 8 tools/_scratch/agg_operands.pt $l $O/stamps8_l$l.npz > $O/agg_phase_v8_l$l.txt 2>&1; tail -20 $O/agg_phase_v8_l$l.txt
done
