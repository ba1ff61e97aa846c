set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
./tools/experiments/valu_lds_rate 2>&1 | head -20 | tee $O/valu_rate_wall.txt
MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py --noise 1.0 2>&1 | grep -v amdgpu.ids | tee $O/fwd_ab_double.txt
cp tools/experiments/libmvdetr_ops_single.so mvdetr_amd/csrc/libmvdetr_ops.so
MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py --noise 1.0 2>&1 | grep -v amdgpu.ids | tee $O/fwd_ab_single.txt
