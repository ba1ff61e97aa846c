#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_msda_gpu.py tests/test_fused_train_gpu.py -m gpu -x -q -k "backward or bwd or fused_train or training or d32 or many or groups or sweep" 2>&1 | tail -4
for lib in "" libmvdetr_ops_r05.so; do
  echo "== lib=${lib:-new} stress16"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} python tools/microbench.py --iters 10 --only msda --config stress16 2>&1 | grep "msda_bwd\|msda_fwd\[realistic\]\|fused"
done
cd /tmp && export TMPDIR=/tmp
for lib in "" libmvdetr_ops_r05.so; do
echo "== kernels stress16, lib=${lib:-new}"
MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} rocprofv3 --kernel-trace -d $R/gpurun_out/ab_trace -o t -- python $R/tools/microbench.py --iters 5 --only msda --config stress16 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/ab_trace/t_results.db --filter msda | cut -c1-70,112-160
rm -rf $R/gpurun_out/ab_trace
done
