#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_warp_gpu.py tests/test_frame_gpu.py tests/test_postprocess.py -m gpu -x -q 2>&1 | tail -8
python tools/microbench.py --iters 30 --skip-bwd 2>&1 | grep -v amdgpu.ids | grep "warp\|copy\|#"
python tools/microbench.py --iters 10 --skip-bwd --config stress16 2>&1 | grep "warp"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1
