R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call22; mkdir -p $O; cd $R
timeout 120 tools/ubench/_bin/mfma_peak 2>&1 | tee $O/mfma_peak.jsonl
