#!/bin/bash
# one-pass backward: phase stamps of one workgroup + SQ counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/libmvdetr_ops_optrace.so python tools/experiments/op_trace.py 2>&1 | grep -v amdgpu.ids | head -60
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_op -o pmc -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/pmc_op/pmc_results.db --filter onepass | cut -c1-150
rm -rf $O/pmc_op
