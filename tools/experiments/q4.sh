R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_fused_train_gpu.py -m gpu -x -q 2>&1 | tail -4
python tools/microbench.py --only msda --iters 10 --config stress16 2>&1 | grep "fused\|msda_bwd\[realistic"
