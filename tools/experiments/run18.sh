cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $O
cp mvdetr_amd/csrc/libmvdetr_ops.so /tmp/lib_orig.so
for v in orig NOLDS NOSAMP NOCOPY ONETAP; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so mvdetr_amd/csrc/libmvdetr_ops.so; else cp tools/experiments/lib_abl_$v.so mvdetr_amd/csrc/libmvdetr_ops.so; fi
  echo "== $v"; timeout 200 python tools/experiments/fwd_ab.py --noise 0 2>&1 | grep "fused-sl noise\|unfused"
done | tee $O/fwd_ablation.txt
