#!/bin/bash
# deterministic backward: parity tests + timing at Wildtrack size (default vs MVDETR_MSDA_BWD_DETERMINISTIC=1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_msda_deterministic_gpu.py -m gpu -x -q 2>&1 | tail -15


python tools/microbench.py --iters 30 2>&1 | grep "msda_bwd"
MVDETR_MSDA_BWD_DETERMINISTIC=1 python tools/microbench.py --iters 30 2>&1 | grep "msda_bwd"
