#!/bin/bash
# Profiles `python bench.py` with rocprofv3 on the GPU box and writes the judged summaries under
# gpurun_out/profile_<tag>/ (copy them into profiles/ afterwards):
#   <tag>_kernel_stats.txt    per-kernel calls / avg / min / max us            (--kernel-trace --stats)
#   <tag>_pmc_*.txt           per-kernel PMC averages, one pass per counter group (--pmc only)
#   <tag>_traffic.json        HBM-side bytes per call of every kernel group bench.py quotes: (2*FETCH_SIZE + WRITE_SIZE)*1024
# usage: bash tools/profile_bench.sh r01 [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profile_$TAG
mkdir -p $O
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-gemm-tuning $@"   # (tuning would fill the trace with candidate GEMMs)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/bench.py $ARGS > $O/bench_under_trace.log 2>&1
# the headline kernel alone: the same command restricted to the launches `roofline.avg_launch_us` averages (calibrated frames + hot-path
# passes); the full run above launches msda_fwd_group2 on five other inputs too (init weights, uncalibrated, spread sweep, iid, training)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_h -o t -- python $R/bench.py $ARGS --headline-only > $O/bench_under_trace_headline.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $O/pmc_fetch -o p -- python $R/bench.py $ARGS > $O/bench_under_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $O/pmc_write -o p -- python $R/bench.py $ARGS > $O/bench_under_pmc_write.log 2>&1
# the headline kernel's memory-side bytes from the launches `roofline` averages alone (the full run's msda_fwd_group2 rows mix in the
# spread sweep and the iid input, whose far taps fetch more)
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_h -o p -- python $R/bench.py $ARGS --headline-only > $O/bench_under_pmc_fetch_h.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_h -o p -- python $R/bench.py $ARGS --headline-only > $O/bench_under_pmc_write_h.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_sq -o p -- python $R/bench.py $ARGS > $O/bench_under_pmc_sq.log 2>&1
cd $R
python tools/rocpd_summary.py $O/trace/t_results.db > $O/${TAG}_kernel_stats.txt
python tools/rocpd_summary.py $O/trace_h/t_results.db --filter mvdetr > $O/${TAG}_kernel_stats_headline.txt
python tools/rocpd_summary.py $O/pmc_fetch/p_results.db --filter mvdetr > $O/${TAG}_pmc_fetch.txt
python tools/rocpd_summary.py $O/pmc_write/p_results.db --filter mvdetr > $O/${TAG}_pmc_write.txt
python tools/rocpd_summary.py $O/pmc_sq/p_results.db --filter mvdetr > $O/${TAG}_pmc_sq.txt
python - <<PY
import json, sqlite3
def avg(db, counter, kern):
    cur = sqlite3.connect(db).cursor()
    v = [r[0] for r in cur.execute("select value from counters_collection where counter_name=? and kernel_name like ?", (counter, f"%{kern}%"))]
    return sum(v) / len(v) if v else None
out = {}
# bench.py key -> the kernels of one call (per-launch averages are summed: a call launches each of them once)
groups = {"msda_fwd": ["msda_fwd_group"], "warp_fwd": ["warp_fwd_cl<"], "warp_fwd_nchw": ["warp_fwd_nchw_patch"],
          "warp_bwd": ["warp_bwd_scans", "warp_bwd_gather"],
          # (public contract, round 6's default: probe + value_tok + sampling; the grad_value-only one-pass instantiations are
          # OnePassCfg<4, 16, 6, 0, ..>, the deterministic launch is OnePassCfg<.., 1, 8>, .., true>)
          "msda_bwd": ["msda_locality_probe", "msda_bwd_value_tok<16, 0>", "msda_bwd_sampling"],
          "msda_bwd_deterministic": ["msda_det_absmax", "msda_bwd_onepass<0, mvdetr::OnePassCfg<4, 16, 6, 1", "msda_det_finish"],
          "msda_train": ["msda_fwd_group2<%7, 2, 2>", "msda_bwd_onepass<1, mvdetr::OnePassCfg<4, 16, 6, 0", "msda_bwd_fused_sampling"]}
for key, kerns in groups.items():
    h = "_h" if key == "msda_fwd" else ""          # the headline kernel: the --headline-only passes
    f = [avg("$O/pmc_fetch" + h + "/p_results.db", "FETCH_SIZE", k) for k in kerns]
    w = [avg("$O/pmc_write" + h + "/p_results.db", "WRITE_SIZE", k) for k in kerns]
    f = [x for x in f if x is not None]
    w = [x for x in w if x is not None]
    if f and w:
        out[key + "_fetch_size_kb_raw"] = sum(f)
        out[key + "_write_size_kb"] = sum(w)
        out[key + "_bytes_per_launch"] = int((2 * sum(f) + sum(w)) * 1024)
        out[key + "_kernels"] = kerns
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over: python bench.py $ARGS (msda_fwd: the same with "
               "--headline-only, i.e. only the launches roofline.avg_launch_us averages); "
               "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE tallies every 128-byte line request at 64 bytes "
               "(MI355X_MICROARCH.md, HBM section) -- calibrated in round 6 on 1 GiB device copies with 4, 8 and 16 bytes per lane and on "
               "sparse 8-byte reads (tools/copy_calibration.py, profiles/r06_copy_calibration.txt): the factor is 2 for every width; "
               "WRITE_SIZE is exact for whole lines and counts 32 bytes per 16-byte partial write. "
               "Infinity-Cache hits are included, so this is an upper bound on HBM bytes.")
json.dump(out, open("$O/${TAG}_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/trace $O/trace_h $O/pmc_fetch $O/pmc_write $O/pmc_fetch_h $O/pmc_write_h $O/pmc_sq
ls -la $O
