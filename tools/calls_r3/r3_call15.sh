R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call15; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_igemm_gpu.py -q -x -k "partial_last_step or coalesced_rows" 2>&1 | tail -5
ONLY_K=1 timeout 600 python tools/tune_conv.py > $O/tune_k1.log 2>&1; tail -3 $O/tune_k1.log | cut -c1-200
