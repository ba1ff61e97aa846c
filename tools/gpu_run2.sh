#!/bin/bash
# GPU trip 2: tile-kernel parity, microbench, rocprofv3 kernel stats + a few PMC passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "tile or auto_dispatch or backward_checksums" > $O/pytest_tile.log 2>&1
echo "pytest rc=$?" >> $O/pytest_tile.log
python tools/microbench.py --iters 30 --skip-bwd > $O/micro2.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_micro -o micro -- python $R/tools/microbench.py --iters 20 --skip-bwd > $O/prof_micro.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $O/pmc_fetch -o pmc -- python $R/tools/microbench.py --iters 3 --skip-bwd > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $O/pmc_write -o pmc -- python $R/tools/microbench.py --iters 3 --skip-bwd > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_sq -o pmc -- python $R/tools/microbench.py --iters 3 --skip-bwd > $O/pmc_sq.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $O/pmc_ta -o pmc -- python $R/tools/microbench.py --iters 3 --skip-bwd > $O/pmc_ta.log 2>&1
cd $R
tail -5 $O/pytest_tile.log; cat $O/micro2.log; ls $O/prof_micro | head; find $O/prof_micro -name "*stats*" | head
