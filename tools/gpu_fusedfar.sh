#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_fused_train_gpu.py tests/test_msda_deterministic_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/microbench.py --iters 30 2>&1 | grep "msda_bwd_fused\|fused_train"
python tools/experiments/fused_noise_sweep.py 2>&1 | grep noise
