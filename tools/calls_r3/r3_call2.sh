# round-3 GPU call 2: full GPU suite on the pair-storage engine, precision sweep, default bench (with the in_tolerance block), per-layer profile of bf16x3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 600 python tools/precision_sweep.py > $O/precision_sweep.log 2>&1; tail -40 $O/precision_sweep.log
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3 -o run -- python $R/bench.py --precision bf16x3 --no-pipeline --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_x3.log 2>&1
python $R/tools/layer_report.py $O/prof_x3/run_kernel_trace.csv v > $O/conv_layers_x3.txt 2>&1; tail -12 $O/conv_layers_x3.txt
python $R/tools/frame_report.py $O/prof_x3/run_kernel_trace.csv 45 > $O/frame_report_x3.txt 2>&1; head -48 $O/frame_report_x3.txt
rm -f $O/prof_x3/run_kernel_trace.csv
