R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call14; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_igemm_gpu.py -q -x -k "partial_last_step or coalesced_rows" 2>&1 | tail -5
ONLY_K=1 ONLY_NAME=s timeout 600 python tools/tune_conv.py > $O/tune_k1.log 2>&1; cut -c1-60 $O/tune_k1.log; grep -o "t7[0-9] *[0-9]*\|t11[0-7] *[0-9]*\|best.*" $O/tune_k1.log | tr '\n' ' ' | sed 's/best/\nbest/g'
ONLY_K=1 ONLY_NAME=fpn timeout 600 python tools/tune_conv.py 2>&1 | cut -c1-30,200-700
