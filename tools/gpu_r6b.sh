#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/libmvdetr_ops_rftrace.so MVDETR_DEBUG_OCCUPANCY=1 python tools/experiments/rf_trace.py 2>&1 | tail -50
python -m pytest tests/test_knob_routes_gpu.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -40
