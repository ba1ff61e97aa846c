#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_msda_gpu.py tests/test_msda_deterministic_gpu.py tests/test_fused_train_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/microbench.py --iters 30 2>&1 | grep "msda_bwd"
MVDETR_MSDA_BWD_DETERMINISTIC=1 python tools/microbench.py --iters 10 2>&1 | grep "msda_bwd"
python tools/experiments/fused_noise_sweep.py 2>&1 | grep noise
bash tools/gpu_bwd_noise.sh
