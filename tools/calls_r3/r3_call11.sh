# Probes for the round-4 backbone work: (1) what the vendor GEMM library does on the plain-GEMM layers of the path, (2) SQ counters
# of the convolution kernels inside benchmark frames (MFMA busy, LDS activity / conflicts, wait buckets).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/probe11; mkdir -p $O; cd $R
timeout 300 python tools/gemm_probe.py $O/gemm_probe.jsonl 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline --no-in-tolerance"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_a -o run -- $B > $O/pmc_a.log 2>&1 || echo "pass a failed"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_b -o run -- $B > $O/pmc_b.log 2>&1 || echo "pass b failed"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_c -o run -- $B > $O/pmc_c.log 2>&1 || echo "pass c failed"
python $R/tools/pmc_kernels.py $O/conv_sq_counters.txt $O/pmc_a $O/pmc_b $O/pmc_c -- conv3x3_pipe gemm1x1_pipe igemm_dma
tail -3 $O/pmc_a.log $O/pmc_b.log $O/pmc_c.log | cut -c1-200
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c
