#!/bin/bash
# split backward (scatter-only one-pass kernel + sampling kernels, no probe): parity + A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
if [ "$1" != "notest" ]; then
python -m pytest tests/test_msda_gpu.py tests/test_fused_train_gpu.py -m gpu -x -q -k "backward or training or autograd or gradcheck or fused_train or pair" 2>&1 | tail -8
fi
for impl in ${IMPLS:-split twopass}; do
  echo "## impl=$impl"
  MVDETR_MSDA_BWD_IMPL=$impl python tools/microbench.py --iters 20 2>&1 | grep -v amdgpu.ids | grep "msda_bwd"
  MVDETR_MSDA_BWD_IMPL=$impl python tools/microbench.py --iters 10 --config multiviewx 2>&1 | grep "msda_bwd"
  MVDETR_MSDA_BWD_IMPL=$impl python tools/microbench.py --iters 5 --config stress16 2>&1 | grep "msda_bwd"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/split_trace -o t -- python $R/tools/microbench.py --iters 5 > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/split_trace/t_results.db --filter bwd | cut -c1-150
rm -rf $O/split_trace
