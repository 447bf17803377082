R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pf; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python $R/bench.py --no-pipeline --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-pmc --latency-groups 0 > $O/prof.log 2>&1
python $R/tools/frame_report.py $O/prof/run_kernel_trace.csv 70 > $O/frame_report.txt 2>&1
rm -rf $O/prof
