R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call13; mkdir -p $O; cd $R
timeout 900 python tools/conv_ablation.py all 2>&1 | tee $O/conv_ablation.txt
