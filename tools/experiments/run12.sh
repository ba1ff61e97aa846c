cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
(MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py --noise 0 1.0 2.0; MVDETR_MSDA_QUAD=0 timeout 300 python tools/experiments/fwd_ab.py --noise 1.0; MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py --config multiviewx --noise 1.0) 2>&1 | grep -v amdgpu.ids | tee $O/fwd_ab_v5.txt
