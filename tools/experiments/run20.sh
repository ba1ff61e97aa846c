cd $GRAFT_REPO_ROOT
timeout 300 python tools/experiments/fwd_ab.py --config stress16 --noise 1.0 --iters 5 2>&1 | grep -v "amdgpu.ids\|q-major"
timeout 600 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -k "many_camera or sweep or slice" 2>&1 | tail -3
