#!/bin/bash
# round 4, call A: parity of the new raw layout / job map, A/B timings, phase stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "fused or slice" 2>&1 | tail -3
for jm in band blocks; do
  MVDETR_MSDA_JOBMAP=$jm python tools/experiments/fwd_variants.py --noise 0 1 --iters 40 2>&1 | grep -v amdgpu.ids
done
MVDETR_MSDA_JOBMAP=blocks python tools/experiments/fwd_variants.py --config multiviewx --noise 1 --iters 40 2>&1 | grep -v amdgpu.ids
MVDETR_MSDA_JOBMAP=band python tools/experiments/fwd_variants.py --config multiviewx --noise 1 --iters 40 2>&1 | grep -v amdgpu.ids
MVDETR_MSDA_JOBMAP=blocks python tools/experiments/fwd_variants.py --config multiviewx --batch 4 --noise 1 --iters 20 2>&1 | grep -v amdgpu.ids
MVDETR_MSDA_JOBMAP=band python tools/experiments/fwd_variants.py --config multiviewx --batch 4 --noise 1 --iters 20 2>&1 | grep -v amdgpu.ids
export MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/libmvdetr_ops_trace.so
for jm in band blocks; do
  MVDETR_MSDA_JOBMAP=$jm python tools/experiments/group_trace.py --noise 1 2>&1 | grep -v amdgpu.ids | tee $O/trace_$jm.txt
done
