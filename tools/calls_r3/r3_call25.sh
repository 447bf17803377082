R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call25; mkdir -p $O; cd $R
timeout 60 tools/ubench/fill_bw 2>&1 | tee $O/fill_bw.txt
