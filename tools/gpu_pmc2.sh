#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
for tile in 4x16 6x12; do
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS"; do
  rm -rf $O/pmc_x
  MVDETR_MSDA_BWD_TILE=$tile rocprofv3 --pmc $set -d $O/pmc_x -o pmc -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
  echo "## tile=$tile"; (cd $R; python tools/rocpd_summary.py $O/pmc_x/pmc_results.db --filter "onepass<0" | grep -v "^==" | cut -c1-120)
done; done
rm -rf $O/pmc_x
