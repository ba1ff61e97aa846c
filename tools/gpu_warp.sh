#!/bin/bash
# warp iteration loop on the GPU box: parity tests, microbench, kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_warp_gpu.py -m gpu -x -q 2>&1 | tail -${TAIL:-6}
python tools/microbench.py --iters 20 --only warp 2>&1 | grep -v amdgpu.ids | grep "warp\|copy"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/warp_trace -o t -- python $R/tools/microbench.py --iters 10 --only warp > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/warp_trace/t_results.db | head -30
rm -rf $O/warp_trace
python tools/fuzz_warp.py --minutes ${FUZZ_MIN:-2} --seed 5 2>&1 | grep -v amdgpu.ids | tail -5
