#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
python -m pytest tests/test_warp_gpu.py tests/test_msda_gpu.py tests/test_cabi.py -m gpu -x -q 2>&1 | tail -5
python tools/microbench.py --iters 20 --skip-bwd 2>&1 | grep "msda_fwd\[uniform\|warp_bwd"
python tools/microbench.py --iters 20 --skip-bwd --config multiviewx 2>&1 | grep "msda_fwd\[uniform"
