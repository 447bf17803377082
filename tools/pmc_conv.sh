cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { d=$1; shift; timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$d -o run -- python $R/tools/probe/run_conv_once.py > $R/gpurun_out/$d.log 2>&1 || echo "pass $d failed"; }
mkdir -p $R/gpurun_out
run cpmc1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU
run cpmc2 SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS
run cpmc3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run cpmc4 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
python $R/tools/pmc_report.py $R/gpurun_out/cpmc1 $R/gpurun_out/cpmc2 $R/gpurun_out/cpmc3 $R/gpurun_out/cpmc4
