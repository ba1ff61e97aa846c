#!/bin/bash
# SQ counters of the fused forward (one variant), per dispatch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
V=${1:-22}
cd /tmp && export TMPDIR=/tmp
MVDETR_MSDA_GROUP_VAR=$V rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_q -o pmc -- python $R/tools/experiments/fwd_variants.py --noise 1 --iters 3 > /dev/null 2>&1
MVDETR_MSDA_GROUP_VAR=$V rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY -d $O/pmc_q2 -o pmc -- python $R/tools/experiments/fwd_variants.py --noise 1 --iters 3 > /dev/null 2>&1
cd $R
for d in pmc_q pmc_q2; do
python tools/rocpd_summary.py $O/$d/pmc_results.db --filter msda_fwd_group 2>&1 | grep "avg=\|PMC" | cut -c1-180
done
rm -rf $O/pmc_q $O/pmc_q2
