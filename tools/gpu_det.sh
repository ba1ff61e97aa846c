#!/bin/bash
# deterministic backward: parity tests + timing at Wildtrack size (default vs MVDETR_MSDA_BWD_DETERMINISTIC=1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_msda_deterministic_gpu.py -m gpu -x -q 2>&1 | tail -5
MVDETR_MSDA_BWD_DETERMINISTIC=1 python tools/microbench.py --iters 30 2>&1 | grep "msda_bwd"
cd /tmp && export TMPDIR=/tmp
MVDETR_MSDA_BWD_DETERMINISTIC=1 rocprofv3 --kernel-trace -d $O/det_trace -o t -- python $R/tools/microbench.py --iters 10 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $O/det_trace/t_results.db --filter "msda_" | cut -c1-70,112-150
rm -rf $O/det_trace
