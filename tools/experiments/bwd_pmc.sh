#!/bin/bash
# kernel durations + LDS counters + memory-side bytes of the backward kernels at Wildtrack size
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/bwd_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/tools/experiments/bwd_trace.py > $O/run.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS -d $O/pmc -o p -- python $R/tools/experiments/bwd_trace.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmcf -o p -- python $R/tools/experiments/bwd_trace.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmcw -o p -- python $R/tools/experiments/bwd_trace.py > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $O/trace/t_results.db --filter msda
python tools/rocpd_summary.py $O/pmc/p_results.db --filter msda_bwd
python tools/rocpd_summary.py $O/pmcf/p_results.db --filter msda_bwd | grep -A1 "PMC"
python tools/rocpd_summary.py $O/pmcw/p_results.db --filter msda_bwd | grep -A1 "PMC"
rm -rf $O/trace $O/pmc $O/pmcf $O/pmcw
