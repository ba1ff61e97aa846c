cd $GRAFT_REPO_ROOT
cp tools/experiments/lib_trace.so mvdetr_amd/csrc/libmvdetr_ops.so
timeout 200 python tools/experiments/group_trace.py 2>&1 | grep -v amdgpu.ids | tee $GRAFT_REPO_ROOT/gpurun_out/group_trace.txt
