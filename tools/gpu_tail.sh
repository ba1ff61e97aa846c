#!/bin/bash
# far-tap tail of msda_fwd_group2: flat list with prefetch (default) vs round 4's per-(camera, level) tail (libmvdetr_ops_oldtail.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "fused or tile or auto_dispatch or wildtrack_forward" 2>&1 | tail -2
for lib in "" mvdetr_amd/csrc/libmvdetr_ops_tail1.so; do
  echo "== lib=${lib:-default}"
  MVDETR_OPS_LIB=$lib python tools/microbench.py --iters 30 --skip-bwd 2>&1 | grep "msda_fwd\[realistic\] impl=tile\|fused\[all\|fused_train"
  MVDETR_OPS_LIB=$lib python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; print('roofline', r.get('avg_launch_us'), r['frac'], r['spread_sweep']['avg_launch_us'])
r = d['roofline_iid_offsets']; print('iid', r.get('avg_launch_us'), r['frac'], {k: v['avg_launch_us'] for k, v in r['spread_sweep'].items()})
print('value', d['value'], d['ms_per_step'])
"
done
