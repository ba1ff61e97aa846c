cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_warp_gpu.py tests/test_frame_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/microbench.py --iters 20 2>&1 | grep "warp\|copy"
timeout 300 python tools/microbench.py --iters 5 --config stress16 --skip-bwd 2>&1 | grep "warp\|copy"
