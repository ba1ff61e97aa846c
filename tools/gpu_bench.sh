#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
( time python bench.py "$@" ) > $O/bench_run.log 2>&1
grep '^{' $O/bench_run.log | tail -1 > $O/bench_line.json
python - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/bench_line.json'))
print('value',d['value'],'ms/step',d['ms_per_step'])
for k in ('roofline','roofline_warp','roofline_warp_bwd','roofline_msda_bwd'):
    if d.get(k): print(k,d[k]['kernel'],d[k]['avg_launch_us'],d[k]['frac'],d[k].get('init_weights'))
print('hot',d['hot_path']); print('startup',d.get('startup')); print('cpu',d['cpu_baseline'] and (d['cpu_baseline']['value'],d['cpu_baseline']['cores']))
print(d['roofline']['code_object'])
PY
tail -4 $O/bench_run.log | grep real
