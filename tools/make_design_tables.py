#!/usr/bin/env python3
"""Generates the measurement tables of DESIGN.md section 5 from the committed files under profiles/, so that a table row can
never disagree with the profile it cites (a hand-typed row said 91.7 us where the profile said 503.8 us).

    python tools/make_design_tables.py --tag r03            # print the block
    python tools/make_design_tables.py --tag r03 --write    # rewrite it inside DESIGN.md (between the GENERATED markers)

tests/test_design_tables.py fails when DESIGN.md's block differs from what this script produces for the committed profiles.
"""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN = "<!-- BEGIN GENERATED: tools/make_design_tables.py --tag {tag} (do not edit by hand) -->"
END = "<!-- END GENERATED -->"


def _load(tag, name):
    path = os.path.join(ROOT, "profiles", f"{tag}_{name}")
    return open(path).read() if os.path.exists(path) else None


def bench_tables(tag):
    raw = _load(tag, "bench.json")
    if raw is None:
        return [f"*(profiles/{tag}_bench.json is missing)*"]
    d = json.loads(raw.strip().splitlines()[-1])
    out = [f"**`python bench.py` on one MI355X** (`profiles/{tag}_bench.json`): **{d['value']} {d['unit']}** "
           f"({d['ms_per_step']} ms per frame, {d['steps']} timed steps, n_gpus {d['n_gpus']}, dtype {d['dtype']}); hot path alone "
           f"(warp + shadow transformer, features resident) **{d['hot_path']['ms_per_frame']} ms/frame**; CPU baseline "
           f"({d['cpu_baseline']['kind']}, {d['cpu_baseline']['cores']} threads) {d['cpu_baseline']['value']} frames/s."
           if d.get("cpu_baseline") else
           f"**`python bench.py`** (`profiles/{tag}_bench.json`): {d['value']} {d['unit']}.", ""]
    out += ["| bench.py key | kernel(s) | avg µs (HIP events) | algorithmic bytes / launch | GB/s | % of 8 TB/s | launches |",
            "|---|---|---|---|---|---|---|"]
    for key in ("roofline", "roofline_iid_offsets", "roofline_warp", "roofline_warp_bwd", "roofline_msda_bwd", "roofline_train_step"):
        r = d.get(key)
        if not r:
            continue
        out.append(f"| `{key}` | `{r['kernel']}` | {r['avg_launch_us']} | {r['algorithmic_bytes_per_launch']:,} | {r['achieved']:,} | "
                   f"{100 * r['frac']:.1f} | {r['launches_timed']} |")
    tr = [(key, d[key]) for key in ("roofline", "roofline_warp", "roofline_warp_bwd", "roofline_msda_bwd", "roofline_train_step")
          if d.get(key) and d[key].get("traffic")]
    if len(tr) > 1:
        out += ["", "Memory-side bytes per call (rocprofv3 PMC passes, `" + str(tr[0][1].get("traffic_source")) + "`; (2·FETCH_SIZE + "
                "WRITE_SIZE)·1024, Infinity-Cache hits included): "
                + "; ".join(f"`{k}` {r['traffic']:,} = {r['traffic'] / r['algorithmic_bytes_per_launch']:.2f}×" for k, r in tr) + "."]
    ts = d.get("roofline_train_step")
    if ts and ts.get("forward_us"):
        out += ["", f"`roofline_train_step` = fused forward (+ statistics) {ts['forward_us']} µs + fused backward {ts['backward_us']} µs "
                    f"({100 * ts['backward_frac']:.1f} % of the roofline on the unfused backward's byte count)."]
    r = d.get("roofline", {})
    extra = []
    if r.get("init_weights"):
        extra.append(f"`roofline` with the reference's zero offset weights: {r['init_weights']['avg_launch_us']} µs, "
                     f"{100 * r['init_weights']['frac']:.1f} %")
    if r.get("uncalibrated_offsets"):
        u = r["uncalibrated_offsets"]
        extra.append(f"`roofline` with the perturbation as rounds 2 – 4 quoted it (offset spreads {u['offset_std_px_per_layer']} px per layer instead of "
                     f"the calibrated 1 px; like for like with `BENCH_r03`): {u['avg_launch_us']} µs, {100 * u['frac']:.1f} %")
    if r.get("traffic"):
        extra.append(f"memory-side bytes per launch (PMC, `{r.get('traffic_source')}`): {r['traffic']:,} = "
                     f"{r['traffic'] / r['algorithmic_bytes_per_launch']:.2f}× the algorithmic bytes")
    if r.get("code_object"):
        c = r["code_object"]
        extra.append(f"code object of that instantiation: {c['num_regs']} registers, {c['scratch_bytes_per_lane']} B/lane scratch, "
                     f"{c['static_lds_bytes']} B static LDS")
    w = d.get("roofline_warp", {}).get("nchw_to_nchw")
    if w:
        extra.append(f"warp in the literal kornia layouts (NCHW → NCHW, `{w['kernel']}`): {w['avg_launch_us']} µs, {100 * w['frac']:.1f} %")
    wp = d.get("roofline_warp_bwd", {}).get("planned")
    if wp:
        extra.append(f"warp gradient with its geometry plan reused (`{wp['kernel']}` alone, the steady state of training without "
                     f"augmentation): {wp['avg_launch_us']} µs, {100 * wp['frac']:.1f} %")
    wt = d.get("roofline_warp_bwd", {}).get("tagged")
    if wt:
        extra.append(f"the same through the ONE-call entry with a version tag of the matrices (`mvdetr_warp_perspective_backward_tagged_f32`, "
                     f"ABI 12): {wt['avg_launch_us']} µs, {100 * wt['frac']:.1f} %")
    bd = d.get("roofline_msda_bwd", {}).get("deterministic")
    if bd:
        extra.append(f"MSDA backward in the opt-in bit-reproducible mode (`mvdetr_msda_set_backward_deterministic`, ABI 13): "
                     f"{bd['avg_launch_us']} µs, {100 * bd['frac']:.1f} %")
    if extra:
        out += ["", "; ".join(extra) + "."]
    sw = r.get("spread_sweep")
    if sw and sw.get("avg_launch_us"):
        out += ["", "`roofline.spread_sweep` — the same kernel in the model with the calibrated offset projections scaled to another spread "
                    "(`roofline` itself is quoted at 1 px): "
                + "; ".join(f"{k} {v} µs = {100 * sw['frac'][k]:.1f} %" for k, v in sw["avg_launch_us"].items()) + "."]
    si = d.get("roofline_iid_offsets", {}).get("spread_sweep")
    if si:
        out += ["", "`roofline_iid_offsets.spread_sweep` — SURVEY 8d's microbenchmark input (offsets iid per tap) at other spreads: "
                + "; ".join(f"{k} {v['avg_launch_us']} µs = {100 * v['frac']:.1f} %" for k, v in si.items())
                + f"; 1px {d['roofline_iid_offsets']['avg_launch_us']} µs = {100 * d['roofline_iid_offsets']['frac']:.1f} %."]
    return out


LINE = re.compile(r"^(?P<name>.+?)\s+avg\s+(?P<avg>[\d.]+) us\s+med\s+(?P<med>[\d.]+)\s+min\s+(?P<min>[\d.]+)\s+(?P<gbs>[\d.]+) GB/s \(alg\)\s+(?P<pct>[\d.]+)% of 8 TB/s")


def microbench_tables(tag):
    raw = _load(tag, "microbench.txt")
    if raw is None:
        return [f"*(profiles/{tag}_microbench.txt is missing)*"]
    out, rows, title = [], [], None

    def flush():
        if title and rows:
            out.extend(["", f"`tools/microbench.py` — {title} (`profiles/{tag}_microbench.txt`; HIP events, µs):", "",
                        "| op | avg | median | min | algorithmic GB/s | % of 8 TB/s |", "|---|---|---|---|---|---|"] + rows)
    for line in raw.splitlines():
        if line.startswith("#"):
            flush()
            title, rows = line[1:].strip(), []
            continue
        m = LINE.match(line.strip())
        if m:
            rows.append(f"| {m['name'].strip()} | {m['avg']} | {m['med']} | {m['min']} | {m['gbs']} | {m['pct']} |")
    flush()
    return out


def rocprof_table(tag, name, what):
    raw = _load(tag, name)
    if raw is None:
        return []
    rows = []
    for line in raw.splitlines():
        if not line.startswith("mvdetr::"):
            continue
        m = re.match(r"^(?P<k>.+?)\s+(?P<calls>\d+)\s+(?P<avg>[\d.]+)\s+(?P<min>[\d.]+)\s+(?P<max>[\d.]+)\s+\S+$", line.rstrip())
        if m:
            k = re.sub(r"\(.*", "", m["k"]).strip()
            rows.append(f"| `{k}` | {m['calls']} | {m['avg']} | {m['min']} | {m['max']} |")
    if not rows:
        return []
    return ["", f"rocprofv3 `--kernel-trace` over {what} (`profiles/{tag}_{name}`; this repo's kernels only, µs):", "",
            "| kernel | calls | avg | min | max |", "|---|---|---|---|---|"] + rows


def generate(tag):
    lines = [BEGIN.format(tag=tag), ""]
    lines += bench_tables(tag)
    lines += rocprof_table(tag, "kernel_stats_headline.txt", "`python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-tuning "
                           "--headline-only` (only the launches `roofline.avg_launch_us` averages: calibrated frames + hot-path passes)")
    lines += rocprof_table(tag, "kernel_stats.txt", "`python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gemm-tuning` (the forward "
                           "kernel's row mixes six inputs here: calibrated, init weights, uncalibrated, the spread sweep, iid, training)")
    lines += microbench_tables(tag)
    lines += rocprof_table(tag, "microbench_kernel_stats.txt", "`python tools/microbench.py --iters 10` (realistic AND uniform inputs: "
                           "`min` is the realistic case)")
    lines += ["", END]
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    block = generate(a.tag)
    if not a.write:
        print(block)
        return
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    pat = re.compile(r"<!-- BEGIN GENERATED: tools/make_design_tables\.py.*?<!-- END GENERATED -->", re.S)
    if not pat.search(text):
        raise SystemExit("DESIGN.md has no GENERATED block to replace")
    open(path, "w").write(pat.sub(lambda _: block, text))
    print(f"DESIGN.md: generated block rewritten from profiles/{a.tag}_*")


if __name__ == "__main__":
    main()
