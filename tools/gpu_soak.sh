#!/bin/bash
# soak: randomised parity over every MSDA route (incl. the deterministic backward) and the warp, then the noise sweep
R=$GRAFT_REPO_ROOT; cd $R
python tools/fuzz_parity.py --minutes ${1:-8} --seed ${2:-11} 2>&1 | grep -v amdgpu.ids | tail -12
python tools/fuzz_warp.py --minutes ${3:-3} --seed ${2:-11} 2>&1 | grep -v amdgpu.ids | tail -6
python tools/experiments/noise_sweep.py 2>&1 | grep -v amdgpu.ids
