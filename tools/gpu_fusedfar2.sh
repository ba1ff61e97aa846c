#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for lib in "" libmvdetr_ops_oldfused.so; do
  echo "== ${lib:-default}"
  MVDETR_OPS_LIB=${lib:+$R/mvdetr_amd/csrc/$lib} python tools/experiments/fused_noise_sweep.py 2>&1 | grep noise
done
