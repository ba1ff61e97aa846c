#!/bin/bash
# round 5 A/B on one box: backward variants + forward after the LDS-only barrier
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
python -m pytest tests/test_msda_gpu.py tests/test_fused_train_gpu.py -m gpu -x -q 2>&1 | tail -4
for impl in split twopass onepass; do
  echo "## bwd impl=$impl"
  MVDETR_MSDA_BWD_IMPL=$impl python tools/microbench.py --iters 20 2>&1 | grep -v amdgpu.ids | grep "msda_bwd\|msda_fwd_fused\[all\|msda_fwd\[realistic\] auto\|fused_train"
  MVDETR_MSDA_BWD_IMPL=$impl python tools/microbench.py --iters 10 --config multiviewx 2>&1 | grep "msda_bwd\|msda_fwd_fused\[all\|msda_fwd\[realistic\] auto"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/r5_trace -o t -- python $R/tools/microbench.py --iters 5 > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/r5_trace/t_results.db --filter "bwd\|fwd_group2" | cut -c1-150
rm -rf $O/r5_trace
