cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
for v in "$@"; do
cp tools/experiments/lib_$v.so mvdetr_amd/csrc/libmvdetr_ops.so
echo "== $v"
MVDETR_MSDA_QUAD=1 timeout 300 python tools/experiments/fwd_ab.py --noise 0 1.0 2>&1 | grep -v amdgpu.ids
done | tee $O/fwd_ab_variants.txt
