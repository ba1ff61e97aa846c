#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "backward or training or autograd or gradcheck" 2>&1 | tail -8
python tools/microbench.py --iters 20 2>&1 | grep -v amdgpu.ids | grep "bwd\|#"; python tools/microbench.py --iters 10 --config multiviewx 2>&1 | grep bwd; python tools/microbench.py --iters 5 --config stress16 2>&1 | grep bwd
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/bwd_trace -o t -- python $R/tools/microbench.py --iters 5 > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/bwd_trace/t_results.db --filter bwd | cut -c1-170
rm -rf $O/bwd_trace
