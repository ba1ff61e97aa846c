#!/bin/bash
# warp_fwd_nchw_patch: channels per chunk / LDS budget variants (libmvdetr_ops_pp<CH>_<FLOATS>.so)
R=$GRAFT_REPO_ROOT; cd $R
for lib in "" $(ls mvdetr_amd/csrc/libmvdetr_ops_pp*.so 2>/dev/null); do
  echo "== ${lib:-default}"
  MVDETR_OPS_LIB=${lib:+$R/$lib} python tools/microbench.py --iters 30 --only warp 2>&1 | grep "warp_fwd NCHW"
  MVDETR_OPS_LIB=${lib:+$R/$lib} python tools/microbench.py --iters 10 --only warp --config stress16 2>&1 | grep "warp_fwd NCHW"
done
