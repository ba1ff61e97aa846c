#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_msda_gpu.py tests/test_fused_train_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "backward or training or autograd or gradcheck or fused_train or pair or adjoint" 2>&1 | tail -4
for order in spread ranges; do
  echo "## order=$order"
  MVDETR_MSDA_BWD_ORDER=$order python tools/microbench.py --iters 20 2>&1 | grep -v amdgpu.ids | grep "msda_bwd"
  MVDETR_MSDA_BWD_ORDER=$order python tools/microbench.py --iters 10 --config multiviewx 2>&1 | grep "msda_bwd"
done
echo "## twopass"; MVDETR_MSDA_BWD_IMPL=twopass python tools/microbench.py --iters 20 2>&1 | grep "msda_bwd"
