cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for q in 0 1; do
export MVDETR_MSDA_QUAD=$q
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $O/pmc_t -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $O/pmc_e -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
rocprofv3 --pmc TA_BUSY_sum TA_TA_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum -d $O/pmc_a -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/pmc_t/p_results.db $O/pmc_e/p_results.db $O/pmc_a/p_results.db --filter msda_fwd_ | grep -v "^==\|gather" > $O/pmc_mem_q$q.txt
rm -rf $O/pmc_t $O/pmc_e $O/pmc_a
done
cat $O/pmc_mem_q0.txt $O/pmc_mem_q1.txt
