#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
tools/experiments/lar2.bin 2>&1 | tee $O/lar2.txt
python tools/microbench.py --iters 20 --only msda 2>&1 | grep -v amdgpu.ids | grep "fused\|bwd" | tee $O/microbench_r4i.txt
