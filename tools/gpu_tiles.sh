#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for tile in 4x16 5x16 6x12 8x8; do
  echo "## tile=$tile"
  MVDETR_MSDA_BWD_TILE=$tile python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "backward_encoder_shapes_vs_oracle or golden_backward" 2>&1 | tail -2
  MVDETR_MSDA_BWD_TILE=$tile python tools/microbench.py --iters 20 2>&1 | grep -v amdgpu.ids | grep "msda_bwd"
  MVDETR_MSDA_BWD_TILE=$tile python tools/microbench.py --iters 10 --config multiviewx 2>&1 | grep "msda_bwd"
done
