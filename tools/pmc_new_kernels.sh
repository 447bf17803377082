# SQ counters, IN the benchmark frame, of the kernels round 5 added or rebuilt ("rocprofv3 SQ counters of the new
# kernels under profiles/r5/"): three `rocprofv3 --pmc` passes (counters only: --kernel-trace, no other trace domain) over the eager
# bench, summarised per kernel by tools/pmc_kernels.py.   usage (GPU box): bash tools/pmc_new_kernels.sh <out dir under gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc_new}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { d=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$d -o run -- python $R/bench.py --eager --steps 4 --warmup 2 --no-cpu-baseline --no-in-tolerance > $O/$d.log 2>&1 || echo "pass $d failed"; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run p2 SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
cd $R
python tools/pmc_kernels.py $O/new_kernels_sq_counters.txt $O/p1 $O/p2 -- gemm1x1_split_kernel "1, false, 2>" stem_conv_kernel ese_fused_kernel decode_topk_kernel topk_kernel prop_select_kernel gn_stats_kernel
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete
