cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-rooflines 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', r['avg_launch_us'], r['frac'], r['init_weights']['avg_launch_us'], d['hot_path']['ms_per_frame'])"; }
cp mvdetr_amd/csrc/libmvdetr_ops.so /tmp/base.so
run clamp3
cp tools/experiments/lib_s4.so mvdetr_amd/csrc/libmvdetr_ops.so; run clamp4
cp tools/experiments/lib_s2.so mvdetr_amd/csrc/libmvdetr_ops.so; run clamp2
cp /tmp/base.so mvdetr_amd/csrc/libmvdetr_ops.so; run clamp3_again
