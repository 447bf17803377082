# rocprofv3 kernel trace + stats of the default bench (writes under gpurun_out/prof_bench_<tag>)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r1}
rm -rf $R/gpurun_out/prof_bench_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_$TAG -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_bench_$TAG.log 2>&1
tail -1 $R/gpurun_out/prof_bench_$TAG.log | cut -c1-200
python $R/tools/layer_report.py $R/gpurun_out/prof_bench_$TAG/run_kernel_trace.csv | tail -12
head -40 $R/gpurun_out/prof_bench_$TAG/run_kernel_stats.csv | cut -c1-150
