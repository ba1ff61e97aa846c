#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
python -m pytest tests/test_msda_gpu.py tests/test_fullsize_gpu.py tests/test_frame_gpu.py tests/test_cabi.py -m gpu -x -q 2>&1 | tail -5
for c in wildtrack multiviewx; do
python tools/microbench.py --iters 20 --skip-bwd --config $c 2>&1 | grep "msda_fwd"
done
python tools/experiments/fwd_variants.py --config multiviewx --noise 1 --iters 30 2>&1 | grep level-outer | cut -c1-40,118-200
python tools/experiments/fwd_variants.py --config multiviewx --batch 4 --noise 1 --iters 20 2>&1 | grep level-outer | cut -c1-40,118-200
