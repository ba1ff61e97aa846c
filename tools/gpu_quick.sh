#!/bin/bash
# quick iteration loop on the GPU box: tile parity tests, microbench, SQ counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "tile or auto_dispatch or wildtrack_forward" 2>&1 | tail -4
python tools/microbench.py --iters 30 --skip-bwd 2>&1 | grep -v amdgpu.ids | grep "msda\|#"
python tools/microbench.py --iters 10 --skip-bwd --config stress16 2>&1 | grep "msda_fwd\[realistic\]"
python tools/microbench.py --iters 10 --skip-bwd --config multiviewx 2>&1 | grep "msda_fwd\[realistic\]"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_q -o pmc -- python $R/tools/microbench.py --iters 3 --skip-bwd > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/pmc_q/pmc_results.db --filter msda_fwd_tile --per-dispatch | grep -A1 "SQ_" | grep "per-dispatch" | cut -c1-150
python tools/rocpd_summary.py $O/pmc_q/pmc_results.db --filter msda_fwd_tile | grep "SQ_\|tile" | cut -c1-120
rm -rf $O/pmc_q
