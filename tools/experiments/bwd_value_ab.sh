#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
for v in tokens planes tokens; do
echo "== $v"; MVDETR_MSDA_BWD_VALUE=$v python tools/microbench.py --iters 20 --only msda 2>&1 | grep -v amdgpu.ids | grep "bwd" 
done | tee $O/microbench_r4l.txt
MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/libmvdetr_ops_bwdtrace.so python tools/experiments/bwd_trace.py 2>&1 | grep -v amdgpu.ids | head -16 | tee $O/bwd_trace.txt
