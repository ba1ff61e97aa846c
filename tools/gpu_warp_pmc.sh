#!/bin/bash
# PMC passes over the warp microbenchmark (gather backward)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $O/wpmc -o p -- python $R/tools/microbench.py --iters 3 --only warp > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/wpmc/p_results.db --filter warp_bwd | cut -c1-160
rm -rf $O/wpmc
cd /tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/wpmc -o p -- python $R/tools/microbench.py --iters 3 --only warp > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/wpmc/p_results.db --filter warp_bwd | cut -c1-160
rm -rf $O/wpmc
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
