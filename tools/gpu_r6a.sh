#!/bin/bash
# round 6: parity of the backward changes + A/B against round 5's library and variants
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_msda_gpu.py tests/test_fused_train_gpu.py tests/test_msda_deterministic_gpu.py -m gpu -x -q -k "backward or bwd or fixed_point or fused_train or training or deterministic or adjoint or nonfinite or checksums" 2>&1 | tail -5
for lib in "" libmvdetr_ops_r05.so libmvdetr_ops_d3.so libmvdetr_ops_d3l4.so libmvdetr_ops_samelevel.so; do
  echo "== lib=${lib:-new}"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} python tools/microbench.py --iters 40 --only msda 2>&1 | grep "msda_bwd"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} python tools/microbench.py --iters 20 --only msda --config multiviewx 2>&1 | grep "msda_bwd"
done
echo "== lib=new order=spread"
MVDETR_MSDA_BWD_ORDER=spread python tools/microbench.py --iters 40 --only msda 2>&1 | grep "msda_bwd"
cd /tmp && export TMPDIR=/tmp
for lib in "" libmvdetr_ops_d3.so libmvdetr_ops_samelevel.so; do
  echo "== kernels, lib=${lib:-new}"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} rocprofv3 --kernel-trace -d $R/gpurun_out/ab_trace -o t -- python $R/tools/microbench.py --iters 10 --only msda > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/ab_trace/t_results.db --filter msda_bwd | cut -c1-60,112-160
  rm -rf $R/gpurun_out/ab_trace
done
