R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-in-tolerance 2>/dev/null | tail -1 | cut -c1-250
