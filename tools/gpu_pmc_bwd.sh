#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_b -o pmc -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/pmc_b/pmc_results.db --filter msda_bwd --per-dispatch | cut -c1-160 | head -60
rm -rf $O/pmc_b
