#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_msda_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "backward or bwd or sweep or training or grad" 2>&1 | tail -3
python tools/microbench.py --iters 30 2>&1 | grep "msda_bwd"
bash tools/gpu_bwd_noise.sh
