cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
for p in 0 1; do echo "== PAIR=$p"; MVDETR_MSDA_PAIR=$p timeout 300 python tools/experiments/fwd_ab.py --noise 0 1.0 2>&1 | grep -v amdgpu.ids; done | tee $O/fwd_ab_pair.txt
cd /tmp && export TMPDIR=/tmp
for p in 0 1; do
MVDETR_MSDA_PAIR=$p timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $O/pmc_t$p -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/pmc_t$p/p_results.db --filter msda_fwd_group | grep "avg=" | cat
rm -rf $O/pmc_t$p
done | tee $O/pmc_pair.txt
