#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for impl in split onepass; do
  echo "== MVDETR_MSDA_BWD_IMPL=$impl"
  MVDETR_MSDA_BWD_IMPL=$impl python tools/microbench.py --iters 30 2>&1 | grep "msda_bwd\[realistic\]\|msda_bwd_fused"
  MVDETR_MSDA_BWD_IMPL=$impl python tools/microbench.py --iters 10 --config multiviewx 2>&1 | grep "msda_bwd\[realistic\]\|msda_bwd_fused"
done
