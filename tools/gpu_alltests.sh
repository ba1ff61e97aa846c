#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/alltests.txt
