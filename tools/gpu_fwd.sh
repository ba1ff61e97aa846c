#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_msda_gpu.py tests/test_frame_gpu.py -m gpu -x -q -k "fused or tile or auto_dispatch or wildtrack_forward or frame or hot_path" 2>&1 | tail -4
python tools/microbench.py --iters 30 --skip-bwd 2>&1 | grep "msda_fwd\[realistic\] impl=tile\|fused\[all"
python tools/microbench.py --iters 5 --skip-bwd --config stress16 2>&1 | grep "msda_fwd\[realistic\] impl=tile\|msda_fwd\[uniform\] impl=tile\|fused\[all"
MVDETR_MSDA_GROUP=0 python tools/microbench.py --iters 5 --skip-bwd --config stress16 2>&1 | grep "msda_fwd\[realistic\] impl=tile\|fused\[all"
