#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_knob_routes_gpu.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -15
python -m pytest tests/test_msda_gpu.py tests/test_fused_train_gpu.py tests/test_msda_deterministic_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -4
for lib in "" libmvdetr_ops_r05.so; do
  echo "== lib=${lib:-new}"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} python tools/microbench.py --iters 40 --only msda 2>&1 | grep "msda_bwd"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} python tools/microbench.py --iters 20 --only msda --config multiviewx 2>&1 | grep "msda_bwd"
done
cd /tmp && export TMPDIR=/tmp
echo "== kernels, lib=new"
rocprofv3 --kernel-trace -d $R/gpurun_out/ab_trace -o t -- python $R/tools/microbench.py --iters 10 --only msda > /dev/null 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/ab_trace/t_results.db --filter msda_bwd | cut -c1-60,112-160
rm -rf $R/gpurun_out/ab_trace
