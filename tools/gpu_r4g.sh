#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_train_gpu.py -m gpu -x -q 2>&1 | tail -25
