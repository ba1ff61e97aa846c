#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_msda_gpu.py tests/test_frame_gpu.py -m gpu -x -q -k "fused or tile or auto_dispatch or wildtrack_forward or frame or hot_path" 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['hot_path'])"
python tools/microbench.py --iters 30 --skip-bwd 2>&1 | grep "msda_fwd\[realistic\] impl=tile\|fused"
python tools/microbench.py --iters 10 --skip-bwd --config multiviewx 2>&1 | grep "msda_fwd\[realistic\] impl=tile\|fused\[all"
