#!/usr/bin/env python3
"""Benchmark of the multiview ground-plane fusion path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One step = one multiview frame: N_cam synthetic views [1, N_cam, 3, 720, 1280] already resident in
HBM -> ResNet-18 trunk + heads (PyTorch-ROCm) -> homography warp (HIP) -> shadow transformer with
3 x MSDeformAttn (HIP) -> BEV heat-map, eval mode, fp32, seeded random weights (no network for the
pretrained download or the datasets).  This is BASELINE.json configs[1], "Wildtrack 7-cam,
--world_feat deform_trans, ResNet18, 1 x MI355X".

Prints ONE JSON line (rank 0):
  metric/value     multiview frames/s, whole job (all ranks)
  roofline         (input: the model's sampling-offset / attention projections are seeded stand-ins for learned ones, calibrated
                   layer by layer on the model's own queries so that the learned part of the offsets has SURVEY 8d's spread of
                   1 px -- `config.offset_calibration` lists the spreads before; `roofline.uncalibrated_offsets` is the same
                   measurement on the uncalibrated perturbation rounds 2 - 4 quoted, `roofline.init_weights` on the reference's
                   zero-initialised projections)
                   the dominant HIP kernel of the path, MSDA forward: algorithmic bytes (SURVEY 8d:
                   4*(S*M*D + 3*Lq*M*L*P + Lq*M*D) per launch) / its average launch duration measured
                   here with HIP events on the launching stream, against the 8 TB/s HBM peak;
                   `traffic` = HBM-side bytes per launch from the committed rocprofv3 PMC passes
                   (profiles/*_traffic.json), null when that file is absent
  roofline_warp / roofline_warp_bwd / roofline_msda_bwd / roofline_train_step / roofline_iid_offsets
                   the other kernels SURVEY 8d names, same accounting (4*N*C*(h*w + H*W) bytes for the warp and its
                   gradient; 4*(Lq*M*D + 2*S*M*D + 6*Lq*M*L*P) for the MSDA backward), HIP events over 12 launches each,
                   measured on rank 0 after the timed region; roofline.code_object = registers / scratch (spill) bytes per
                   lane / static LDS of the forward instantiation that ran, read from the code object
  cpu_baseline     the oracle (the reference's CPU formulation: grid_sample-based deformable
                   attention + kornia-semantics warp + the same trunk) timed on this box's host cores:
                   a thread sweep (1, physical/2, physical) over the two hot ops on a bounded sample
                   (2 warm-ups, median of 3), then whole frames at the best thread count; baseline only
  hot_path         the same step without trunk and heads (warp + shadow transformer), for scale

Multi-GPU: `--parallel dp` (default) runs one independent frame per rank -- frames shard with no
data-path collective, weak scaling; `--parallel views` partitions the cameras of ONE frame over the
ranks: trunk, warp and token conv per view, then `--encoder sharded` (default; each rank runs the shadow
transformer on its own cameras' queries, all-gathering the projected values once per layer and all-reducing
the merge convolution's partial sums) or `--encoder replicated` (one all-gather of per-view world tokens, then
every rank runs the whole encoder).  RCCL over xGMI either way; scaling "strong" (one frame).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # synthetic-input builders shared with the parity tests

import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="wildtrack", choices=["wildtrack", "multiviewx", "stress16"])
    ap.add_argument("--arch", default="resnet18", choices=["resnet18", "resnet50"],
                    help="trunk (out of the path's scope; resnet50 + --config multiviewx --batch 4 is BASELINE configs[3])")
    ap.add_argument("--parallel", default="dp", choices=["dp", "views"])
    ap.add_argument("--encoder", default="sharded", choices=["sharded", "replicated"],
                    help="--parallel views only: shadow transformer partitioned by camera, or replicated")
    ap.add_argument("--batch", type=int, default=1, help="frames per step per rank (dp mode; the reference only supports 1)")
    ap.add_argument("--augment", action="store_true", help="random affine augmentation matrices instead of identity")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed frames and the hot-path runs: no init-weight / uncalibrated / spread-sweep runs and none of the "
                         "other kernels' rooflines -- what tools/profile_bench.sh traces so that rocprofv3's average of the forward kernel "
                         "covers the SAME launches as `roofline.avg_launch_us` (every other run in this file launches that kernel on another input)")
    ap.add_argument("--no-kernel-rooflines", action="store_true",
                    help="skip roofline_warp / roofline_warp_bwd / roofline_msda_bwd (measured after the timed region)")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="keep hipBLASLt's default solution per GEMM instead of PyTorch TunableOp's measured pick")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0)
    ap.add_argument("--msda-impl", default="auto", choices=["auto", "gather", "tile"])
    ap.add_argument("--offset-std-px", type=float, default=1.0,
                    help="std of the seeded perturbation of the learned sampling offsets, in pixels (SURVEY 8d: bias grid + "
                         "N(0, 1 px)); 0 keeps the reference's zero-initialised projections (every query samples one constant pattern)")
    return ap.parse_args()


def relaunch_under_torchrun(a):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU,
    the driver's own command shape) and hand back their exit code.  With fewer GPUs than ranks (a 1-GPU test box) the
    ranks share GPUs and exchange through gloo, which stages CUDA tensors through the host -- RCCL wants one GPU per rank."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < a.gpus:
        env.setdefault("MVDETR_DIST_BACKEND", "gloo")
    # --standalone: torchrun's own c10d rendezvous picks the port (no bind-close-rebind race of a port chosen here)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={a.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def perturb_sampling(model, std_px, seed=1234):
    """Seeded stand-in for LEARNED sampling offsets / attention logits.  The reference initialises both projections'
    weights to zero (ms_deform_attn.py:62-73), so with random-init weights every query would sample the same constant
    1..4 px pattern -- the most local input there is.  SURVEY 8d defines the measurement input as bias grid +
    N(0, 1 px): scale the offset projection so that its output has that spread on the model's own tokens."""
    if std_px <= 0 or not hasattr(model.world_feat, "encoder"):
        return None
    g = torch.Generator().manual_seed(seed)
    for layer in model.world_feat.encoder.layers:
        at = layer.self_attn
        with torch.no_grad():
            # queries are LayerNorm outputs + position/camera embeddings: per-channel rms ~ 1.4 => std(w . q) ~ 1.4 * sqrt(C) * std(w)
            at.sampling_offsets.weight.copy_(torch.randn(at.sampling_offsets.weight.shape, generator=g)
                                             * (std_px / (1.4 * at.d_model ** 0.5)))
            at.attention_weights.weight.copy_(torch.randn(at.attention_weights.weight.shape, generator=g)
                                              * (1.0 / (1.4 * at.d_model ** 0.5)))
    return std_px


def calibrate_sampling(attn_layers, run_once, std_px):
    """perturb_sampling assumes queries of per-channel rms ~1.4; the first layer's queries are the token convolution's
    output (not LayerNorm's), whose scale follows the trunk (a ResNet-50 trunk gave offsets several times the intended
    spread).  Layer by layer: run one frame with a forward pre-hook that measures the std of the learned part of that
    layer's offsets in pixels (sampling_offsets(query) minus its bias), then rescale its two projections so that the
    spread is ``std_px`` (logits: 1).  Returns the measured spreads before / after per layer (reported in the line)."""
    report = []
    for at in attn_layers:
        seen = {}

        def hook(mod, args, kwargs, seen=seen):
            q = args[0] if args else kwargs["query"]
            with torch.no_grad():
                seen["off"] = float(torch.nn.functional.linear(q, mod.sampling_offsets.weight).float().std())
                seen["logit"] = float(torch.nn.functional.linear(q, mod.attention_weights.weight).float().std())

        h = at.register_forward_pre_hook(hook, with_kwargs=True)
        try:
            run_once()
            torch.cuda.synchronize()
        finally:
            h.remove()
        before = seen.get("off")
        if before and before > 0 and seen.get("logit", 0) > 0:
            with torch.no_grad():
                at.sampling_offsets.weight.mul_(std_px / before)
                at.attention_weights.weight.mul_(1.0 / seen["logit"])
            at.cache_fused_projection(True)                 # (the cached permuted weight is rebuilt from the new values)
        report.append({"offset_std_px_before": round(before, 3) if before else None, "offset_std_px": std_px})
    return report


class KernelTimer:
    """HIP-event brackets around every MSDA forward launch (plain or fused entry point) on the
    launching (current) stream."""

    NAMES = ("ms_deform_attn_forward", "ms_deform_attn_forward_fused")

    def __init__(self, MSDA):
        self.events, self.enabled = [], False
        for name in self.NAMES:
            setattr(MSDA, name, self._wrap(getattr(MSDA, name)))

    def _wrap(self, fn):
        def timed(*a, **kw):
            if not self.enabled:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            # algorithmic bytes of THIS launch: value once + locations + weights + output, from the
            # call's own shapes (a rank of a query-sharded run has fewer queries than value tokens)
            B, S, M, D = a[0].shape
            L, Lq = a[1].shape[0], out.shape[1]
            P = 4
            self.events.append((e0, e1, 4 * B * (S * M * D + 3 * Lq * M * L * P + Lq * M * D)))
            return out
        return timed

    def average_us(self):
        """(average launch duration in us, launches, average algorithmic bytes per launch)"""
        if not self.events:
            return None, 0, None
        ts = [a.elapsed_time(b) * 1e3 for a, b, _ in self.events]
        return sum(ts) / len(ts), len(ts), sum(n for _, _, n in self.events) / len(self.events)


def time_launches(fn, launches=12, warmup=3):
    """Average / min duration in us of `fn` (one device call) over `launches` launches, each bracketed by HIP events on
    the current stream -- the stream the kernels are launched on."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = [e0.elapsed_time(e1) * 1e3 for e0, e1 in ev]
    return sum(ts) / len(ts), min(ts)


def roofline_entry(kernel, us, us_min, nbytes, launches, what, extra=None):
    ach = nbytes / (us * 1e-6) / 1e9
    d = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
         "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(nbytes),
         "avg_launch_us": round(us, 2), "min_launch_us": round(us_min, 2), "launches_timed": launches, "what": what}
    d.update(extra or {})
    return d


def other_kernel_rooflines(model, geom, feat, proj, MSDA, launches=12):
    """SURVEY 8d also asks for the warp and the MSDA backward: the other kernels of the path, on the same shapes and the
    same locality-realistic inputs, HIP events over `launches` launches each (rank 0, after the timed region)."""
    from helpers import encoder_msda_inputs
    from mvdetr_amd.ops import warp as warp_mod
    from mvdetr_amd.ops import warp_perspective
    out = {}
    N, C = geom.num_cam, feat.shape[1]
    h, w = feat.shape[-2:]
    H, W = model.Rworld_shape
    wbytes = 4 * N * C * (h * w + H * W)                                  # 4 N C (h w + H W)
    f1 = feat[:N]
    p1 = proj[:N]
    with torch.no_grad():
        nhwc = model.channels_last
        us, mn = time_launches(lambda: warp_perspective(f1, p1, (H, W), channels_last_out=nhwc), launches)
        out["roofline_warp"] = roofline_entry(
            warp_mod.last_kernel(), us, mn, wbytes, launches,
            "the warp as the model runs it: channels_last trunk features read in place -> token layout [N,H,W,C]"
            if nhwc else "NCHW features -> NCHW world features")
        f_nchw = f1.contiguous(memory_format=torch.contiguous_format)
        us, mn = time_launches(lambda: warp_perspective(f_nchw, p1, (H, W)), launches)
        out["roofline_warp"]["nchw_to_nchw"] = {
            "kernel": warp_mod.last_kernel(), "avg_launch_us": round(us, 2), "frac": round(wbytes / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
            "what": "the literal layouts of kornia.warp_perspective at mvdetr.py:194-195 (NCHW source and destination)"}
        go = torch.randn(N, H, W, C, device=feat.device)
        gs = torch.empty(N, h, w, C, device=feat.device)
        pm = p1.to(device=feat.device, dtype=torch.float32).contiguous()
        us, mn = time_launches(lambda: warp_mod._launch("backward", go, pm, N, C, h, w, H, W, 3, gs), launches)
        out["roofline_warp_bwd"] = roofline_entry(
            warp_mod.last_kernel(), us, mn, wbytes, launches,
            "gradient of the warp w.r.t. the features, channel-last on both sides (C ABI entry: candidate scans + gather + "
            "stragglers; deterministic, no atomics)")
        # what a training loop without augmentation runs in steady state: the autograd function keeps the gradient's geometry
        # ("plan": candidate scans, heavy-block and odd-pixel lists) per matrix tensor, so only the gather is launched
        plan = warp_mod._plans.get(pm, (N, C, h, w, H, W))
        us_p, mn_p = time_launches(lambda: warp_mod._launch("backward", go, pm, N, C, h, w, H, W, 3, gs, plan=plan), launches)
        out["roofline_warp_bwd"]["planned"] = {
            "kernel": warp_mod.last_kernel(), "avg_launch_us": round(us_p, 2), "min_launch_us": round(mn_p, 2),
            "frac": round(wbytes / (us_p * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
            "what": "mvdetr_warp_perspective_backward_planned_f32 with the plan of mvdetr_warp_backward_plan_f32 reused (what "
                    "WarpPerspectiveFunction.backward does while the matrices do not change)"}
        # ... and what a C-ABI caller gets with the one-call entry and a version tag of its matrices (ABI 12): the library keeps
        # the plan in its own per-(device, stream) scratch while the tag stays
        us_t, mn_t = time_launches(lambda: warp_mod._launch("backward", go, pm, N, C, h, w, H, W, 3, gs, tag=0x5eed), launches)
        out["roofline_warp_bwd"]["tagged"] = {
            "kernel": warp_mod.last_kernel(), "avg_launch_us": round(us_t, 2), "min_launch_us": round(mn_t, 2),
            "frac": round(wbytes / (us_t * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
            "what": "mvdetr_warp_perspective_backward_tagged_f32, same tag on every call: one launch (the gather) per call"}
        del go, gs
        # MSDA backward at this configuration's encoder shape, SURVEY 8d's input (bias grid + N(0, 1 px) offsets)
        wf = model.world_feat
        hh, ww = (int(x) for x in wf.spatial_shapes[0])
        M_, D_ = 8, wf.hidden_dim // 8
        value, shapes, lsi, loc, aw = [x.to(feat.device) for x in encoder_msda_inputs(N, hh, ww, M_, D_, 4, seed=0)]
        S = N * hh * ww
        gout = torch.randn(1, S, M_ * D_, device=feat.device)
        bbytes = 4 * (S * M_ * D_ + 2 * S * M_ * D_ + 6 * S * M_ * N * 4)   # 4 (Lq M D + 2 S M D + 6 Lq M L P)
        us, mn = time_launches(lambda: MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, gout, 64), launches)
        out["roofline_msda_bwd"] = roofline_entry(
            "msda_locality_probe + msda_bwd_value_tok + msda_bwd_sampling_resident (+ memset of grad_value)" if (D_ == 16 and N <= 7)
            else "msda_locality_probe + msda_bwd_value_tok + msda_bwd_sampling_groups (+ memset of grad_value)", us, mn, bbytes, launches,
            "MultiScaleDeformableAttention.ms_deform_attn_backward (public contract), SURVEY 8d's locality-realistic input: "
            "bias grid + N(0, 1 px) offsets, softmax(N(0,1)) weights")
        if D_ == 16:
            # the opt-in bit-reproducible mode (mvdetr_msda_set_backward_deterministic, ABI 13): one kernel, 64-bit fixed-point sums
            MSDA.set_backward_deterministic(True)
            try:
                us_d, mn_d = time_launches(lambda: MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, gout, 64), max(4, launches // 2))
            finally:
                MSDA.set_backward_deterministic(False)
            out["roofline_msda_bwd"]["deterministic"] = {
                "kernel": "msda_det_absmax + msda_bwd_onepass<deterministic> + msda_det_finish (+ memsets of grad_value and the 64-bit accumulators)",
                "avg_launch_us": round(us_d, 2), "min_launch_us": round(mn_d, 2),
                "frac": round(bbytes / (us_d * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                "what": "the same call with mvdetr_msda_set_backward_deterministic(1): grad_value bit-identical run to run"}
        del value, loc, aw, gout
        # the fused TRAINING pair (what MSDeformAttn.forward runs when gradients are needed): forward from the raw offsets /
        # logits + softmax statistics, backward to grad_value and the gradient of the raw tensor.  Bytes: SURVEY 8d's counts
        # of the unfused forward + backward it replaces, so that the fraction compares like for like (the pair itself never
        # materialises sampling_locations / attention_weights or their gradients).
        if MSDA.fused_train_supported(1, S, M_, D_, N, S, 4):
            from helpers import fused_train_inputs
            value, shapes, lsi, ref_lm, raw, _ = [x.to(feat.device) for x in fused_train_inputs(N, hh, ww, M_, D_, 4, seed=0)]
            gout = torch.randn(1, S, M_ * D_, device=feat.device)
            fbytes = 4 * (S * M_ * D_ + 3 * S * M_ * N * 4 + S * M_ * D_)
            # the inference kernel on the same iid input (SURVEY 8d's MICROBENCHMARK definition: every tap's offset drawn
            # independently; in the model the learned part of an offset is a linear function of the query, so neighbouring
            # queries' taps move together and the LDS reads conflict less -- `roofline` is the in-model figure)
            us_i, mn_i = time_launches(lambda: MSDA.ms_deform_attn_forward_fused(value, shapes, lsi, ref_lm, None, None, raw=raw,
                                                                                 ref_level_major=True, raw_level_outer=True), launches)
            out["roofline_iid_offsets"] = roofline_entry(
                MSDA.last_forward_kernel(), us_i, mn_i, fbytes, launches,
                "mvdetr_msda_forward_fused_f32 on SURVEY 8d's microbenchmark input: bias grid + iid N(0, 1 px) per tap, logits N(0, 1)")
            sweep = {}
            for px in (0.5, 2.0, 4.0):
                raw_s = fused_train_inputs(N, hh, ww, M_, D_, 4, seed=0, noise_px=px)[4].to(feat.device)
                us_s, _ = time_launches(lambda: MSDA.ms_deform_attn_forward_fused(value, shapes, lsi, ref_lm, None, None, raw=raw_s,
                                                                                   ref_level_major=True, raw_level_outer=True), 6)
                sweep[f"{px:g}px"] = {"avg_launch_us": round(us_s, 2), "frac": round(fbytes / (us_s * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}
                del raw_s
            out["roofline_iid_offsets"]["spread_sweep"] = sweep
            o_, st_ = MSDA.ms_deform_attn_forward_fused_train(value, shapes, lsi, ref_lm, raw)
            us_f, mn_f = time_launches(lambda: MSDA.ms_deform_attn_forward_fused_train(value, shapes, lsi, ref_lm, raw), launches)
            us_b, mn_b = time_launches(lambda: MSDA.ms_deform_attn_backward_fused(gout, value, shapes, lsi, ref_lm, raw, st_, o_), launches)
            out["roofline_train_step"] = roofline_entry(
                "msda_fwd_group2 (+ statistics) ; msda_bwd_onepass<fused, grad_value only> + msda_bwd_fused_sampling (+ memset of grad_value)"
                if (D_ == 16 and N in (6, 7)) else
                "inference forward of the shape + msda_softmax_stats ; msda_bwd_onepass<fused> (+ memset of grad_value)" if D_ == 16 else
                "inference forward of the shape + msda_softmax_stats ; msda_bwd_value_tok<32, fused> + msda_bwd_sampling_groups<fused> (+ memset of grad_value)",
                us_f + us_b, mn_f + mn_b, fbytes + bbytes, launches,
                "mvdetr_msda_forward_fused_train_f32 + mvdetr_msda_backward_fused_f32 on the same realistic input in raw form",
                {"forward_us": round(us_f, 2), "backward_us": round(us_b, 2),
                 "backward_frac": round(bbytes / (us_b * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)})
    return out


def write_tunableop_results(tun, path):
    """The in-memory TunableOp results of this process in the CSV format torch.cuda.tunable.read_file() takes
    (validator lines, then op signature, parameter signature, solution, time)."""
    lines = [f"Validator,{k},{v}" for k, v in tun.get_validators()]
    lines += [",".join(str(x) for x in row) for row in tun.get_results()]
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        f.write("\n".join(lines) + "\n")
    os.replace(tmp, path)


def load_traffic():
    """HBM-side bytes per call of the path's kernels from the newest committed PMC summary (profiles/*_traffic.json,
    written by tools/profile_bench.sh): ({"msda_fwd": bytes, "warp_fwd": ..., "warp_bwd": ..., "msda_bwd": ...}, file)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return {}, None
    try:
        d = json.load(open(files[-1]))
        return {k[:-len("_bytes_per_launch")]: v for k, v in d.items() if k.endswith("_bytes_per_launch")}, os.path.basename(files[-1])
    except Exception:
        return {}, None


def _median_time(fn, warmup=2, reps=3):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def cpu_baseline(model, imgs_cpu, proj_cpu, budget_s):
    """The oracle on the host.  (1) thread sweep over the two hot ops on a bounded sample -- MSDA core for ONE camera's
    queries against all cameras' values, warp of ONE view -- 2 warm-ups + median of 3 at 1, physical/2 and physical
    cores (all logical CPUs oversubscribe the oracle's OpenMP/oneDNN loops: 256 threads were slower than 1 on the
    r01 box); (2) whole frames at the best thread count, bounded by `budget_s`."""
    from helpers import encoder_msda_inputs
    from oracle import frame_oracle, torch_oracle
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        import psutil
        physical = min(logical, psutil.cpu_count(logical=False) or logical)
    except Exception:
        physical = max(1, logical // 2)
    wf = model.world_feat
    N = model.num_cam
    h, w = (int(x) for x in wf.spatial_shapes[0])
    value, shapes, _, loc, aw = encoder_msda_inputs(N, h, w, 8, wf.hidden_dim // 8, 4, seed=0)
    one = slice(0, h * w)                                   # one camera's queries
    loc1, aw1 = loc[:, one].contiguous(), aw[:, one].contiguous()
    feat1 = torch.randn(1, wf.hidden_dim, *model.Rimg_shape)
    sweep = {}
    for nt in sorted({1, max(1, physical // 2), physical}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            t_msda = _median_time(lambda: torch_oracle.msda_core(value, shapes, loc1, aw1))
            t_warp = _median_time(lambda: torch_oracle.warp_perspective(feat1, proj_cpu[:1], model.Rworld_shape))
        sweep[nt] = {"msda_core_one_camera_s": round(t_msda, 4), "warp_one_view_s": round(t_warp, 4)}
    best = min(sweep, key=lambda k: sweep[k]["msda_core_one_camera_s"] * N + sweep[k]["warp_one_view_s"] * N)
    torch.set_num_threads(best)
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = model.world_feat.encoder.reference_points.detach().cpu()

    def frame():
        t0 = time.perf_counter()
        with torch.no_grad():
            frame_oracle.forward(p, imgs_cpu, proj_cpu, model.Rworld_shape, ref, model.num_cam)
        return time.perf_counter() - t0

    # one warm-up frame, then up to three timed ones (a median, not one sample) while they fit the budget; a
    # frame longer than the whole budget leaves the warm-up frame as the only sample -- the `sample` string says which it was
    warm = frame()
    times = []
    while len(times) < 3 and (sum(times) + max(times) if times else warm) <= budget_s:
        times.append(frame())
    warm_only = not times
    times = times or [warm]
    med = sorted(times)[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": best, "kind": "port",
            "sample": f"median of {len(times)} whole frame(s) ({model.num_cam} views 3x{imgs_cpu.shape[-2]}x{imgs_cpu.shape[-1]} -> BEV) "
                      f"through oracle/frame_oracle.py on {best} thread(s) "
                      + ("(the warm-up frame itself: it alone exceeded --cpu-budget-s)" if warm_only else "after a warm-up frame")
                      + f", {sum(times):.1f} s timed within a budget of {budget_s:.0f} s; thread count chosen by the "
                      f"sweep below (hot ops on one camera's share, 2 warm-ups + median of 3)",
            "host": {"logical_cpus": logical, "physical_cores": physical}, "thread_sweep": {str(k): v for k, v in sweep.items()}}


class PhaseClock:
    """Wall-clock seconds of the run's phases (the JSON line's `phases_s`): where a fresh box spends its minutes."""

    def __init__(self):
        self.t0 = self.last = time.perf_counter()
        self.phases = {}

    def mark(self, name):
        now = time.perf_counter()
        self.phases[name] = round(self.phases.get(name, 0.0) + now - self.last, 2)
        self.last = now

    def total(self):
        return round(time.perf_counter() - self.t0, 2)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(a))
    clock = PhaseClock()
    from mvdetr_amd import dist as mdist
    rank, world, local_rank = mdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP kernels have no CPU fallback)")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True      # the reference's own setting (main.py:48): MIOpen picks convolutions by measurement
    gemm_tuning = False
    if not a.no_gemm_tuning:
        # the same for the shadow transformer's fp32 GEMMs: TunableOp times the hipBLASLt / rocBLAS solutions of each
        # shape once (during the warm-up steps) and keeps the fastest; same arithmetic type, nothing is written to disk.
        # All-or-nothing: if any step of the set-up fails, tuning is switched off again and the line says so.
        try:
            tun = torch.cuda.tunable
            tun.set_filename(os.path.join(tempfile.gettempdir(), f"mvdetr_bench_tunableop_{os.getpid()}.csv"))
            if hasattr(tun, "write_file_on_exit"):         # (not in every torch build; the results file is scratch anyway)
                tun.write_file_on_exit(False)
            tun.set_max_tuning_duration(200)
            tun.enable(True)
            tun.tuning_enable(True)
            gemm_tuning = bool(tun.is_enabled())
        except Exception as ex:                    # pragma: no cover
            try:
                torch.cuda.tunable.enable(False)
            except Exception:
                pass
            print(f"[bench] TunableOp unavailable, GEMMs keep the library defaults: {ex}", file=sys.stderr)

    import mvdetr_amd.ops  # noqa: F401
    import MultiScaleDeformableAttention as MSDA
    from mvdetr_amd import geometry
    from mvdetr_amd.model import build_model

    MSDA.set_forward_impl(a.msda_impl)
    timer = KernelTimer(MSDA)      # callers look the functions up on the module at call time
    clock.mark("import_and_init")

    geom = geometry.GEOMETRIES[a.config]
    model = build_model(a.config, seed=0, arch=a.arch)
    offset_std = perturb_sampling(model, a.offset_std_px)
    model = model.to(dev).eval()
    attn_layers = [layer.self_attn for layer in model.world_feat.encoder.layers] if hasattr(model.world_feat, "encoder") else []
    for at in attn_layers:
        at.cache_fused_projection(True)                     # inference with frozen weights (see MSDeformAttn)
    N, (Hi, Wi) = geom.num_cam, geom.input_img_shape
    g = torch.Generator().manual_seed(1000 + rank)
    Bf = a.batch if not (a.parallel == "views" and world > 1) else 1
    M = geometry.random_affine_mats(Bf, N, (Hi, Wi), seed=rank) if a.augment else torch.eye(3).repeat(Bf, N, 1, 1)

    if a.parallel == "views" and world > 1:
        runner = mdist.ViewShardedFrame(model, encoder=a.encoder)
        s, e = runner.range
        imgs = torch.randn(1, max(e - s, 0), 3, Hi, Wi, generator=torch.Generator().manual_seed(1000)).to(dev) \
            if e > s else torch.zeros(1, 0, 3, Hi, Wi, device=dev)
        step = lambda: runner(imgs, M)                       # noqa: E731
        frames_per_step, scaling = 1, "strong"
    else:
        imgs = torch.randn(Bf, N, 3, Hi, Wi, generator=g).to(dev)
        def step():
            with torch.no_grad():
                return model(imgs, M)
        frames_per_step, scaling = world * Bf, "weak"

    # the synthetic "learned" offsets are defined by their spread in pixels: calibrate it on this model's own
    # queries, layer by layer -- on ONE fixed seeded frame, so that every rank arrives at the same projections (a view-sharded
    # step has collectives in it: the calibration always runs the unsharded model of the rank)
    clock.mark("model_and_inputs")
    offset_calibration, uncalibrated = None, None

    def calibrate():
        nonlocal offset_calibration, uncalibrated
        if not (offset_std and attn_layers):
            return
        cal_imgs = torch.randn(1, N, 3, Hi, Wi, generator=torch.Generator().manual_seed(1000)).to(dev)
        cal_M = M[:1]

        def cal_run():
            with torch.no_grad():
                model(cal_imgs, cal_M)
        uncalibrated = [(at.sampling_offsets.weight.detach().clone(), at.attention_weights.weight.detach().clone()) for at in attn_layers]
        offset_calibration = calibrate_sampling(attn_layers, cal_run, offset_std)

    tuning_shared = None
    rank0_first = world > 1 and gemm_tuning
    if not rank0_first:
        calibrate()
        clock.mark("offset_calibration_incl_miopen_find")     # (the first frames of the process run here)
    else:
        # rank 0 goes first -- calibration frames and (dp) warm-up steps: TunableOp's measured GEMM picks go to ONE results file
        # and MIOpen's find results to its user database; the other ranks then read both instead of each spending minutes
        # measuring the same shapes at the same time.  `--parallel views`: a view-sharded step has collectives in it, so rank 0
        # cannot run IT alone -- but the calibration frame is the unsharded model and has none: rank 0 runs that one frame first
        # (the trunk's and the replicated encoder's shapes are then known; what only a shard sees is found by every rank in the
        # warm-up steps as before).  Every rank runs the same barrier sequence whatever fails in between.
        tun = torch.cuda.tunable
        shared = os.path.join(tempfile.gettempdir(), f"mvdetr_bench_tunableop_shared_{os.environ.get('MASTER_PORT', '0')}.csv")
        tuning_shared = True
        if rank == 0:
            try:
                calibrate()
                if a.parallel == "dp":
                    for _ in range(max(a.warmup, 1)):
                        step()
                elif offset_calibration is None:           # (nothing to calibrate: still one unsharded frame for the libraries)
                    with torch.no_grad():
                        model(torch.randn(1, N, 3, Hi, Wi, generator=torch.Generator().manual_seed(1000)).to(dev), M[:1])
                torch.cuda.synchronize()
                write_tunableop_results(tun, shared)
            except Exception as ex:                # pragma: no cover
                tuning_shared = False
                print(f"[bench] rank 0: writing the shared TunableOp results failed ({ex}); tuning per rank", file=sys.stderr)
        torch.distributed.barrier()
        if rank != 0:
            try:
                if not (os.path.exists(shared) and tun.read_file(shared)):
                    raise RuntimeError("no usable results file from rank 0")
                tun.tuning_enable(False)
            except Exception as ex:                # pragma: no cover
                tuning_shared = False
                print(f"[bench] rank {rank}: reading the shared TunableOp results failed ({ex}); tuning here", file=sys.stderr)
            calibrate()
        elif offset_calibration is None:
            calibrate()                            # (rank 0's first attempt failed before the calibration)
        clock.mark("offset_calibration_incl_miopen_find")
    clock.mark("gemm_tuning_share")
    for i in range(a.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            clock.mark("first_step_incl_miopen_find_and_gemm_tuning")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    clock.mark("warmup_rest")
    before_timed_s = clock.total()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = mdist.barrier_and_max(time.perf_counter() - t0, dev)
    timer.enabled = False
    clock.mark("timed_steps")
    k_us, k_n, k_bytes = timer.average_us()
    impl = MSDA.last_forward_kernel()
    fwd_resources = MSDA.last_forward_resources()          # registers / scratch (spill) bytes per lane / static LDS of that instantiation
    # ranks the collective library actually connected (one process may be all there is)
    seen = torch.ones(1, device=dev)
    if world > 1:
        torch.distributed.all_reduce(seen)
    n_ranks = int(seen.item())
    # where every rank ran: (rank, device index, devices visible, device uuid-ish name, host)
    import socket
    me = {"rank": rank, "device": dev.index, "gpus_visible": torch.cuda.device_count(),
          "device_name": torch.cuda.get_device_name(dev), "host": socket.gethostname(),
          "bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None)}
    placement = [me]
    if world > 1:
        placement = [None] * world
        torch.distributed.all_gather_object(placement, me)
    distinct_devices = len({(p["host"], p["device"]) for p in placement})

    # ---- the hot path alone (warp + shadow transformer), same inputs ------------------------------------
    hot_ms = None
    if a.parallel == "dp":
        with torch.no_grad():
            feat = model.features(imgs)
            proj = model.frame_proj_mats(M, dev)
            for _ in range(3):
                model.hot_path(feat, proj)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                model.hot_path(feat, proj)
            torch.cuda.synchronize()
            hot_ms = (time.perf_counter() - t1) / a.steps * 1e3 / Bf

    # ---- the same kernel on the reference's initial (zero) offset / attention weights, for comparison ---------------
    init_us = None
    if a.parallel == "dp" and offset_std and attn_layers and not a.headline_only:
        saved = [(at.sampling_offsets.weight.detach().clone(), at.attention_weights.weight.detach().clone()) for at in attn_layers]
        with torch.no_grad():
            for at in attn_layers:
                at.sampling_offsets.weight.zero_()
                at.attention_weights.weight.zero_()
                at.cache_fused_projection(True)
            model.hot_path(feat, proj)
            n0 = len(timer.events)
            timer.enabled = True
            for _ in range(5):
                model.hot_path(feat, proj)
            torch.cuda.synchronize()
            timer.enabled = False
            ts = [e0.elapsed_time(e1) * 1e3 for e0, e1, _ in timer.events[n0:]]
            init_us = sum(ts) / len(ts)
            del timer.events[n0:]
            for at, (ow, aw) in zip(attn_layers, saved):
                at.sampling_offsets.weight.copy_(ow)
                at.attention_weights.weight.copy_(aw)
                at.cache_fused_projection(True)

    # ---- and on the UNCALIBRATED perturbation (rounds 2 - 4a quoted `roofline` on it: like for like with those lines) ----
    uncal_us = None
    if a.parallel == "dp" and uncalibrated and attn_layers and not a.headline_only:
        saved = [(at.sampling_offsets.weight.detach().clone(), at.attention_weights.weight.detach().clone()) for at in attn_layers]
        with torch.no_grad():
            for at, (ow, aw) in zip(attn_layers, uncalibrated):
                at.sampling_offsets.weight.copy_(ow)
                at.attention_weights.weight.copy_(aw)
                at.cache_fused_projection(True)
            model.hot_path(feat, proj)
            n0 = len(timer.events)
            timer.enabled = True
            for _ in range(5):
                model.hot_path(feat, proj)
            torch.cuda.synchronize()
            timer.enabled = False
            ts = [e0.elapsed_time(e1) * 1e3 for e0, e1, _ in timer.events[n0:]]
            uncal_us = sum(ts) / len(ts)
            del timer.events[n0:]
            for at, (ow, aw) in zip(attn_layers, saved):
                at.sampling_offsets.weight.copy_(ow)
                at.attention_weights.weight.copy_(aw)
                at.cache_fused_projection(True)

    # ---- how the same kernel degrades with the spread of the learned offsets: the calibrated (1 px) offset projections scaled
    #      to 0.5 / 1 / 2 / 4 px, in the model (exact for the first layer; later layers' queries move a little with it) ----------
    spread_sweep = None
    if a.parallel == "dp" and offset_calibration and attn_layers and not a.no_kernel_rooflines and not a.headline_only:
        saved = [at.sampling_offsets.weight.detach().clone() for at in attn_layers]
        spread_sweep = {}
        with torch.no_grad():
            for px in (0.5, 1.0, 2.0, 4.0):
                for at, ow in zip(attn_layers, saved):
                    at.sampling_offsets.weight.copy_(ow * (px / offset_std))
                    at.cache_fused_projection(True)
                model.hot_path(feat, proj)
                n0 = len(timer.events)
                timer.enabled = True
                for _ in range(4):
                    model.hot_path(feat, proj)
                torch.cuda.synchronize()
                timer.enabled = False
                ts = [e0.elapsed_time(e1) * 1e3 for e0, e1, _ in timer.events[n0:]]
                del timer.events[n0:]
                spread_sweep[f"{px:g}px"] = round(sum(ts) / len(ts), 2)
            for at, ow in zip(attn_layers, saved):
                at.sampling_offsets.weight.copy_(ow)
                at.cache_fused_projection(True)

    clock.mark("hot_path_and_init_weight_runs")
    if rank != 0:
        return
    alg_bytes = int(k_bytes) if k_bytes else None          # of the launches actually timed (rank 0's)
    full_frame = a.config == "wildtrack" and Bf == 1 and not (a.parallel == "views" and world > 1)
    traffic_all, traffic_src = load_traffic() if full_frame else ({}, None)
    traffic = traffic_all.get("msda_fwd")
    achieved = alg_bytes / (k_us * 1e-6) / 1e9 if (k_us and alg_bytes) else None
    res = {
        "metric": "multiview frames/s (7-cam Wildtrack) + MSDeformAttn HBM GB/s vs roofline",
        "value": round(frames_per_step * a.steps / elapsed, 3), "unit": "frames/s",
        "n_gpus": min(n_ranks, distinct_devices), "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
        "scaling": scaling if distinct_devices >= n_ranks else None,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.config} {N}-cam frame, --world_feat deform_trans, {a.arch} trunk: "
                               f"{N}x3x{Hi}x{Wi} -> {geom.feat_channels}-ch world feat {geom.Rworld_shape[0]}x{geom.Rworld_shape[1]} "
                               f"-> BEV (BASELINE.json configs[1])" if a.config == "wildtrack" else f"{a.config} {N}-cam frame",
                   "frames_per_step": frames_per_step, "batch_per_rank": Bf, "parallelism": f"{a.parallel}{world}" + (f"-{a.encoder}" if a.parallel == "views" and world > 1 else ""),
                   "augment": bool(a.augment), "gemm_tuning": gemm_tuning,
                   "backend": (os.environ.get("MVDETR_DIST_BACKEND") or "nccl (RCCL)") if world > 1 else None,
                   "gpus_visible": torch.cuda.device_count(), "ranks": n_ranks, "distinct_devices": distinct_devices,
                   "oversubscribed": distinct_devices < n_ranks,
                   "rank_placement": [{k: p[k] for k in ("rank", "device", "gpus_visible", "bus_id")} for p in placement],
                   "gemm_tuning_shared_from_rank0": tuning_shared,
                   "weights": "seeded random" + (f"; sampling-offset / attention projections perturbed (seeded) to ~{offset_std:g} px offset std, "
                                                 "SURVEY 8d's locality-realistic input" if offset_std else "; reference init (zero offset weights)"),
                   "offset_calibration": offset_calibration},
        "roofline": {"bound": "hbm", "kernel": impl, "achieved": round(achieved, 1) if achieved else None,
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4) if achieved else None,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_us": round(k_us, 2) if k_us else None, "launches_timed": k_n,
                     "code_object": fwd_resources,
                     "spread_sweep": ({"what": "avg launch us of the same kernel in the model with the calibrated offset projections scaled "
                                               "to the given spread (frac = algorithmic bytes / us / 8 TB/s); `roofline` itself is quoted "
                                               "at 1 px and stays so",
                                       "avg_launch_us": spread_sweep,
                                       "frac": {k: round(alg_bytes / (v * 1e-6) / 1e9 / PEAK_HBM_GBS, 4) for k, v in spread_sweep.items()}}
                                      if spread_sweep and alg_bytes else None),
                     "input": (f"learned-like offsets: bias grid + the model's own offset projection of its queries, calibrated per layer to "
                               f"a spread of {offset_std:g} px (SURVEY 8d: bias grid + N(0, 1 px); config.offset_calibration has the spreads before)" if offset_std
                               else "reference init: constant bias-grid offsets"),
                     "uncalibrated_offsets": ({"avg_launch_us": round(uncal_us, 2), "frac": round(alg_bytes / (uncal_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                                               "offset_std_px_per_layer": [c["offset_std_px_before"] for c in offset_calibration],
                                               "what": "same kernel and frame with the perturbation as rounds 2 - 4 quoted it (projection scaled for "
                                                       "queries of rms 1.4, not measured): like for like with BENCH_r02 / r03"}
                                              if (uncal_us and alg_bytes and offset_calibration) else None),
                     "init_weights": ({"avg_launch_us": round(init_us, 2), "frac": round(alg_bytes / (init_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                                       "what": "same kernel, zero offset / attention weights (every query samples the constant bias grid)"}
                                      if (init_us and alg_bytes) else None)},
        "hot_path": {"ms_per_frame": round(hot_ms, 3) if hot_ms else None,
                     "frames_per_s": round(1e3 / hot_ms, 1) if hot_ms else None,
                     "what": "warp_perspective + DeformTransWorldFeat (3 x MSDeformAttn), features resident"},
    }
    if a.parallel == "dp" and not a.no_kernel_rooflines and not a.headline_only and hasattr(model.world_feat, "encoder"):
        res.update(other_kernel_rooflines(model, geom, feat, proj, MSDA))
        for key, tkey in (("roofline_warp", "warp_fwd"), ("roofline_warp_bwd", "warp_bwd"), ("roofline_msda_bwd", "msda_bwd"),
                          ("roofline_train_step", "msda_train")):
            if key in res and tkey in traffic_all:           # (same source and caveats as roofline.traffic)
                res[key]["traffic"], res[key]["traffic_source"] = traffic_all[tkey], traffic_src
        if "roofline_msda_bwd" in res and isinstance(res["roofline_msda_bwd"].get("deterministic"), dict) and "msda_bwd_deterministic" in traffic_all:
            res["roofline_msda_bwd"]["deterministic"]["traffic"] = traffic_all["msda_bwd_deterministic"]
        if "roofline_warp" in res and "warp_fwd_nchw" in traffic_all:
            res["roofline_warp"]["nchw_to_nchw"]["traffic"] = traffic_all["warp_fwd_nchw"]
        clock.mark("other_kernel_rooflines")
    # memory-side bytes over algorithmic bytes, next to every `traffic` (how much of what moved was re-fetched or written twice)
    def add_ratio(e):
        if isinstance(e, dict):
            if e.get("traffic") and e.get("algorithmic_bytes_per_launch"):
                e["traffic_ratio"] = round(e["traffic"] / e["algorithmic_bytes_per_launch"], 2)
            for v in e.values():
                add_ratio(v)
    for k, e in res.items():
        if k.startswith("roofline"):
            add_ratio(e)
    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(model, imgs[:1].cpu(), model.frame_proj_mats(M[:1]), a.cpu_budget_s)
        clock.mark("cpu_baseline")
    else:
        res["cpu_baseline"] = None
    res["startup"] = {"seconds_before_timed_region": before_timed_s, "phases_s": clock.phases, "total_s": clock.total(),
                      "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE")}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
