R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_sums_gpu.py -q -x 2>&1 | tail -15
