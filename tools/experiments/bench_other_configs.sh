cd $GRAFT_REPO_ROOT
for args in "--config stress16" "--config multiviewx --batch 4 --arch resnet50" "--config multiviewx"; do
  timeout 500 python bench.py $args --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-tuning 2>gpurun_out/bench_cfg_err.log | tail -1 > gpurun_out/bench_cfg.json
  python - <<'PY' || tail -3 gpurun_out/bench_cfg_err.log
import json
d=json.load(open("gpurun_out/bench_cfg.json"))
g=lambda k: (d.get(k) or {}).get("frac")
print(d["config"]["workload"][:40], d["value"], g("roofline"), g("roofline_warp"), g("roofline_warp_bwd"), g("roofline_msda_bwd"), (d.get("roofline_msda_bwd") or {}).get("kernel","")[:48])
PY
done
