set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
(MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py; MVDETR_MSDA_QUAD=0 timeout 300 python tools/experiments/fwd_ab.py --noise 1.0; MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py --config multiviewx --noise 1.0) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/fwd_ab_1.txt
timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r2/pytest_msda_1.txt
