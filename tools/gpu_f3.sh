#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python tools/microbench.py --iters 30 --skip-bwd 2>&1 | grep -v amdgpu.ids | grep "msda\|#"
