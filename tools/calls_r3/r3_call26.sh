R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/call26; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o run -- python $R/bench.py --eager --steps 3 --warmup 1 --no-cpu-baseline --no-in-tolerance > $O/pmc.log 2>&1 || echo "pass failed"
python $R/tools/pmc_kernels.py $O/conv_l2_hits.txt $O/pmc_tcc -- conv3x3_pipe gemm1x1 igemm_dma ese_apply chan_sums
rm -rf $O/pmc_tcc
