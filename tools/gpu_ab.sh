#!/bin/bash
# A/B of library variants on the backward microbenchmark: bash tools/gpu_ab.sh lib1.so lib2.so ...  ("" = the default build)
R=$GRAFT_REPO_ROOT; cd $R
for lib in "" "$@"; do
  echo "== lib=${lib:-default}"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} python tools/microbench.py --iters 30 2>&1 | grep "msda_bwd\|fused_train"
done
cd /tmp && export TMPDIR=/tmp
for lib in "" "$@"; do
  echo "== kernels, lib=${lib:-default}"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} rocprofv3 --kernel-trace -d $R/gpurun_out/ab_trace -o t -- python $R/tools/microbench.py --iters 10 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/ab_trace/t_results.db --filter msda_bwd | cut -c1-60,112-160
  rm -rf $R/gpurun_out/ab_trace
done
