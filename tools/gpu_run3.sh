#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "tile or auto_dispatch or backward_checksums" > $O/pytest_tile.log 2>&1
echo "pytest rc=$?" >> $O/pytest_tile.log
python tools/microbench.py --iters 30 --skip-bwd > $O/micro3.log 2>&1
python tools/microbench.py --iters 10 --skip-bwd --config stress16 > $O/micro3_stress.log 2>&1
python tools/microbench.py --iters 10 --skip-bwd --config multiviewx > $O/micro3_mvx.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_sq3 -o pmc -- python $R/tools/microbench.py --iters 3 --skip-bwd > $O/pmc_sq3.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $O/pmc_fetch3 -o pmc -- python $R/tools/microbench.py --iters 3 --skip-bwd > $O/pmc_fetch3.log 2>&1
cd $R
tail -5 $O/pytest_tile.log; cat $O/micro3.log $O/micro3_stress.log $O/micro3_mvx.log
python tools/rocpd_summary.py $O/pmc_sq3/pmc_results.db $O/pmc_fetch3/pmc_results.db --filter msda_fwd_tile
