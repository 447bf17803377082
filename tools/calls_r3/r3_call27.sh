R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call27; mkdir -p $O; cd $R
SUMS=1 ONLY_K=1 ONLY_NAME=s timeout 100 python tools/tune_conv.py > $O/tune_k1_sums.log 2>&1; cut -c1-40 $O/tune_k1_sums.log | tail -12
