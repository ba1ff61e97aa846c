cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
timeout 600 python -m pytest tests/test_warp_gpu.py tests/test_frame_gpu.py tests/test_dist_gpu.py tests/test_layernorm_gpu.py tests/test_postprocess.py -x -q -m gpu 2>&1 | tail -4
(timeout 300 python tools/experiments/fwd_ab.py --config multiviewx --noise 1.0 --batch 4 --iters 10; timeout 300 python tools/experiments/fwd_ab.py --config wildtrack --noise 1.0 --batch 4 --iters 10) 2>&1 | grep -v "amdgpu.ids\|q-major" | tee $O/fwd_ab_b4.txt
timeout 600 python tools/microbench.py --iters 20 2>&1 | grep -v amdgpu.ids | tee $O/microbench_a.txt
