#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for s in 1 2 3; do
  echo "== sigma $s px"
  rocprofv3 --kernel-trace -d $R/gpurun_out/bn_trace -o t -- python $R/tools/experiments/bwd_noise_split.py $s > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/bn_trace/t_results.db --filter msda_bwd | cut -c1-60,112-150
  rm -rf $R/gpurun_out/bn_trace
done
