cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
for v in 0 4; do echo "== GROUP_VARIANT=$v"; MVDETR_MSDA_GROUP_VARIANT=$v timeout 300 python tools/experiments/fwd_ab.py --noise 0 1.0 2>&1 | grep -v amdgpu.ids | grep -v "q-major"; done | tee $O/fwd_ab_split4.txt
timeout 600 python -m pytest tests/test_warp_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/microbench.py --iters 20 --skip-bwd 2>&1 | grep "warp\|copy"
