#!/bin/bash
# r04 re-baseline: all GPU tests, MIOpen find-mode A/B of the bench start-up (separate user-db / cache dirs), microbench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) 2>&1 | tee $O/alltests.txt
for mode in 2 default; do
  export MIOPEN_USER_DB_PATH=/tmp/miopen_db_$mode MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen_cache_$mode
  mkdir -p $MIOPEN_USER_DB_PATH $MIOPEN_CUSTOM_CACHE_DIR
  if [ $mode = default ]; then unset MIOPEN_FIND_MODE; else export MIOPEN_FIND_MODE=$mode; fi
  ( time python bench.py --no-cpu-baseline ) > $O/bench_mode_$mode.log 2>&1
  grep '^{' $O/bench_mode_$mode.log | tail -1 > $O/bench_mode_$mode.json
  python - $O/bench_mode_$mode.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print('value',d['value'],'ms/step',d['ms_per_step'],'hot',d['hot_path']['ms_per_frame'])
for k in ('roofline','roofline_warp','roofline_warp_bwd','roofline_msda_bwd'):
    if d.get(k): print(k,d[k]['kernel'],d[k]['avg_launch_us'],d[k]['frac'])
print('startup',d.get('startup')); print(d['roofline']['code_object'])
PY
  tail -4 $O/bench_mode_$mode.log | grep real
done
unset MIOPEN_USER_DB_PATH MIOPEN_CUSTOM_CACHE_DIR MIOPEN_FIND_MODE
python tools/microbench.py --iters 20 2>&1 | grep -v amdgpu.ids | tee $O/microbench_r4h.txt
