cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
(timeout 300 python tools/experiments/fwd_ab.py --noise 0 1.0; timeout 300 python tools/experiments/fwd_ab.py --config multiviewx --noise 1.0; timeout 300 python tools/experiments/fwd_ab.py --config stress16 --noise 1.0 --iters 5)  2>&1 | grep -v amdgpu.ids | grep -v "q-major" | tee $O/fwd_ab_dma.txt
