#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
export MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/libmvdetr_ops_trace.so
MVDETR_MSDA_GROUP_VAR=23 python tools/experiments/group_trace.py --noise 1 2>&1 | grep "pairs where"
