cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 2400 python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -25 | tee gpurun_out/r2/pytest_gpu.txt
