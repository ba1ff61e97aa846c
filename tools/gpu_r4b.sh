#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
for v in 22 32 22 32; do
  MVDETR_MSDA_GROUP_VAR=$v python tools/experiments/fwd_variants.py --noise 0 1 --iters 40 2>&1 | grep "level-outer" | sed "s/^/VAR=$v /" | cut -c1-30,118-200
done
