R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call28; mkdir -p $O; cd $R
timeout 70 python bench.py --no-cpu-baseline --no-in-tolerance --steps 40 2>$O/err.txt | tail -1 > $O/bench_head.json; cut -c1-300 $O/bench_head.json; tail -3 $O/err.txt
