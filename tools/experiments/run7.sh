cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
(MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py --noise 0 1.0 2.0; MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py --config multiviewx --noise 1.0) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/fwd_ab_s2.txt
