# deep-ring GEMM tiles (82-89): correctness, then the 1x1 tile sweep at the benchmarked size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call12; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_igemm_gpu.py -q -x -k "partial_last_step or coalesced_rows" 2>&1 | tail -5
ONLY_K=1 timeout 600 python tools/tune_conv.py > $O/tune_k1.log 2>&1; cut -c1-420 $O/tune_k1.log
cp gpurun_out/tuning_mi355x.json $O/tuning_k1.json
