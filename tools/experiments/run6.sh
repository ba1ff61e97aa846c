cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MVDETR_MSDA_QUAD=1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $O/pmc_f -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $O/pmc_w -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $O/pmc_t -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum -d $O/pmc_e -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/pmc_f/p_results.db $O/pmc_w/p_results.db $O/pmc_t/p_results.db $O/pmc_e/p_results.db --filter msda_fwd_quad | grep -v "^==" > $O/pmc_quad_mem.txt
rm -rf $O/pmc_f $O/pmc_w $O/pmc_t $O/pmc_e
cat $O/pmc_quad_mem.txt
