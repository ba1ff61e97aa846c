cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
(timeout 300 python tools/experiments/fwd_ab.py --noise 0 1.0 2.0; timeout 300 python tools/experiments/fwd_ab.py --config multiviewx --noise 1.0; timeout 300 python tools/experiments/fwd_ab.py --config stress16 --noise 1.0 --iters 5) 2>&1 | grep -v amdgpu.ids | tee $O/fwd_ab_slice.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d $O/pmc_t -o p -- python $GRAFT_REPO_ROOT/tools/experiments/fwd_ab.py --noise 1.0 --iters 5 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/pmc_t/p_results.db --filter msda_fwd_group | grep "avg=\|PMC" | sed 's/(float const.*//' | tee $O/pmc_slice.txt
rm -rf $O/pmc_t
