#!/bin/bash
# after widening the fused training pair: whole GPU suite, soak, the training pair at the other BASELINE shapes
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/fuzz_parity.py --minutes 5 --seed 23 2>&1 | grep -v amdgpu.ids | tail -8
python tools/microbench.py --iters 10 --config stress16 2>&1 | grep "msda_bwd\|fused_train\|fused\[all"
python tools/microbench.py --iters 10 --config multiviewx 2>&1 | grep "msda_bwd\|fused_train\|fused\[all"
