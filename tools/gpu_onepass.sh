#!/bin/bash
# one-pass backward: parity + A/B against the two-kernel backward (rounds 2-4) on the GPU box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
if [ "$1" != "notest" ]; then
python -m pytest tests/test_msda_gpu.py tests/test_fused_train_gpu.py -m gpu -x -q -k "backward or training or autograd or gradcheck or fused_train or pair" 2>&1 | tail -12
fi
for lib in ${LIBS:-ops}; do
  echo "## lib=$lib"
  L=$R/mvdetr_amd/csrc/libmvdetr_$lib.so
  MVDETR_OPS_LIB=$L python tools/microbench.py --iters 20 2>&1 | grep -v amdgpu.ids | grep "msda_bwd"
  MVDETR_OPS_LIB=$L python tools/microbench.py --iters 10 --config multiviewx 2>&1 | grep "msda_bwd"
done
MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/libmvdetr_ops_optrace.so python tools/experiments/op_trace.py 2>&1 | grep -v amdgpu.ids | head -${TRACE_LINES:-34}
