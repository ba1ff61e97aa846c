#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
for rep in 1 2; do
for v in A B C D; do echo "== variant $v"; MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/libmvdetr_ops_v$v.so python tools/microbench.py --iters 20 --only msda 2>&1 | grep -v amdgpu.ids | grep "bwd\[real\|bwd_fused"; done
echo "== planes"; MVDETR_MSDA_BWD_VALUE=planes MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/libmvdetr_ops_vA.so python tools/microbench.py --iters 20 --only msda 2>&1 | grep -v amdgpu.ids | grep "bwd\[real\|bwd_fused"
done | tee $O/microbench_r4o.txt
