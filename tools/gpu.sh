#!/bin/bash
# One launcher for everything that runs on the GPU box (replaces rounds 2-5's per-experiment gpu_*.sh scripts):
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/gpu.sh <task> [args] > gpurun_out/<task>.log 2>&1'
# Tasks:
#   tests [pytest args]         the GPU suite (default: everything marked gpu)
#   bench [bench.py args]       the contract benchmark, one JSON line + a short digest
#   micro [config] [iters]      kernel-level timings (tools/microbench.py) of wildtrack | multiviewx | stress16
#   ab_bwd                      MSDA backward, every MVDETR_MSDA_BWD_IMPL x {wildtrack, multiviewx} + kernel trace per route
#   ab_lib lib1.so lib2.so ...  the microbenchmark under other builds of the library (mvdetr_amd/csrc/<lib>; "" = default)
#   trace [microbench args]     rocprofv3 --kernel-trace summary of the microbenchmark
#   pmc <counters...>           one rocprofv3 --pmc pass over the microbenchmark (3 iterations), per-kernel sums
#   det                         the deterministic backward: tests + timings
#   soak [minutes] [seed]       randomised parity (tools/fuzz_parity.py, tools/fuzz_warp.py)
#   copycal                     FETCH_SIZE / WRITE_SIZE calibration: device copies with 4 / 8 / 16 bytes per lane (tools/copy_calibration.py)
#   profiles <tag>              round-end refresh (tools/refresh_profiles.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
task=$1; shift
lib_env() { [ -n "$1" ] && echo "MVDETR_OPS_LIB=$R/mvdetr_amd/csrc/$1"; }
trace_of() {   # trace_of <filter> <env assignments...> -- <command...>
  local filt=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( cd /tmp && export TMPDIR=/tmp && env "${envs[@]}" rocprofv3 --kernel-trace -d $O/_trace -o t -- "$@" > /dev/null 2>&1 )
  python tools/rocpd_summary.py $O/_trace/t_results.db --filter "$filt" | cut -c1-70,112-160; rm -rf $O/_trace
}
case $task in
tests) python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -8 ;;
bench)
  ( time python bench.py "$@" ) > $O/bench_run.log 2>&1
  grep '^{' $O/bench_run.log | tail -1 > $O/bench_line.json
  python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/bench_line.json")))
print("value", d["value"], d["unit"], " ms/step", d["ms_per_step"])
for k in ("roofline", "roofline_iid_offsets", "roofline_warp", "roofline_warp_bwd", "roofline_msda_bwd", "roofline_train_step"):
    if d.get(k):
        print(k, {x: d[k].get(x) for x in ("kernel", "avg_launch_us", "frac", "traffic", "traffic_ratio", "backward_us", "backward_frac")})
print("hot_path", d.get("hot_path")); print("startup", d.get("startup")); print("cpu_baseline", d.get("cpu_baseline"))
PY
  grep real $O/bench_run.log ;;
micro) python tools/microbench.py --config ${1:-wildtrack} --iters ${2:-30} 2>&1 | grep -v amdgpu.ids ;;
ab_bwd)
  for impl in split twopass onepass atomic; do
    for cfg in wildtrack multiviewx; do
      echo "## MVDETR_MSDA_BWD_IMPL=$impl  $cfg"
      MVDETR_MSDA_BWD_IMPL=$impl python tools/microbench.py --only msda --iters 40 --config $cfg 2>&1 | grep "msda_bwd"
    done
  done
  for impl in split twopass onepass; do
    echo "## kernels, MVDETR_MSDA_BWD_IMPL=$impl (wildtrack; the first msda_bwd launches of a kernel are the realistic input, the slow ones the uniform)"
    trace_of "msda_" MVDETR_MSDA_BWD_IMPL=$impl -- python $R/tools/microbench.py --only msda --iters 10
  done ;;
ab_lib)
  for lib in "" "$@"; do
    echo "## lib=${lib:-default}"
    env $(lib_env "$lib") python tools/microbench.py --only msda --iters 40 2>&1 | grep "msda_"
    env $(lib_env "$lib") python tools/microbench.py --only msda --iters 20 --config multiviewx 2>&1 | grep "msda_bwd\|msda_fwd\[realistic\]\|fused"
  done ;;
trace) trace_of "mvdetr" -- python $R/tools/microbench.py --iters 10 "$@" ;;
pmc)
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc "$@" -d $O/_pmc -o p -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1 )
  python tools/rocpd_summary.py $O/_pmc/p_results.db --filter mvdetr | cut -c1-200; rm -rf $O/_pmc ;;
det)
  python -m pytest tests/test_msda_deterministic_gpu.py -m gpu -x -q 2>&1 | tail -5
  MVDETR_MSDA_BWD_DETERMINISTIC=1 python tools/microbench.py --only msda --iters 30 2>&1 | grep "msda_bwd"
  trace_of "msda_" MVDETR_MSDA_BWD_DETERMINISTIC=1 -- python $R/tools/microbench.py --only msda --iters 10 ;;
soak)
  python tools/fuzz_parity.py --minutes ${1:-8} --seed ${2:-11} 2>&1 | grep -v amdgpu.ids | tail -12
  python tools/fuzz_warp.py --minutes ${3:-3} --seed ${2:-11} 2>&1 | grep -v amdgpu.ids | tail -6 ;;
copycal) python tools/copy_calibration.py "$@" ;;
profiles) bash tools/refresh_profiles.sh "$@" ;;
*) echo "unknown task '$task'"; sed -n 2,16p "$0"; exit 2 ;;
esac
