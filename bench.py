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
  roofline         the dominant HIP kernel of the path, MSDA forward: algorithmic bytes (SURVEY 8d:
                   4*(S*M*D + 3*Lq*M*L*P + Lq*M*D) per launch) / its average launch duration measured
                   here with HIP events on the launching stream, against the 8 TB/s HBM peak;
                   `traffic` = HBM-side bytes per launch from the committed rocprofv3 PMC passes
                   (profiles/*_traffic.json), null when that file is absent
  cpu_baseline     the oracle (the reference's CPU formulation: grid_sample-based deformable
                   attention + kornia-semantics warp + the same trunk) timed on this box's host cores
                   for a bounded number of frames; baseline only
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
    ap.add_argument("--parallel", default="dp", choices=["dp", "views"])
    ap.add_argument("--encoder", default="sharded", choices=["sharded", "replicated"],
                    help="--parallel views only: shadow transformer partitioned by camera, or replicated")
    ap.add_argument("--batch", type=int, default=1, help="frames per step per rank (dp mode; the reference only supports 1)")
    ap.add_argument("--augment", action="store_true", help="random affine augmentation matrices instead of identity")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="keep hipBLASLt's default solution per GEMM instead of PyTorch TunableOp's measured pick")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--msda-impl", default="auto", choices=["auto", "gather", "tile"])
    return ap.parse_args()


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


def load_traffic():
    """HBM-side bytes per MSDA-forward launch from the newest committed PMC summary, or None."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d.get("msda_fwd_bytes_per_launch"), os.path.basename(files[-1])
    except Exception:
        return None, None


def cpu_baseline(model, imgs_cpu, proj_cpu, budget_s):
    """Full frame on the host through the oracle (reference CPU formulation), bounded wall time."""
    from oracle import frame_oracle
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = model.world_feat.encoder.reference_points.detach().cpu()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    def one():
        t0 = time.perf_counter()
        with torch.no_grad():
            frame_oracle.forward(p, imgs_cpu, proj_cpu, model.Rworld_shape, ref, model.num_cam)
        return time.perf_counter() - t0

    first = one()
    times = [first]
    if first < budget_s / 3:                    # cheap enough: treat the first call as warm-up
        times = []
        while len(times) < 5 and sum(times) + first <= budget_s:
            times.append(one())
    frames, t_total = len(times), sum(times)
    res = {"value": round(frames / t_total, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{frames} full frame(s) ({model.num_cam} views 3x{imgs_cpu.shape[-2]}x{imgs_cpu.shape[-1]} -> BEV), "
                     f"oracle/frame_oracle.py on {torch.get_num_threads()} host threads, {t_total:.1f} s"}
    # the two hot ops alone (the reference's CPU formulation), all cores and -- because the reference pins
    # OMP_NUM_THREADS=1 (main.py:3) -- one thread; one call each, a few seconds in total
    from helpers import encoder_msda_inputs
    from oracle import torch_oracle
    wf = model.world_feat
    h, w = (int(x) for x in wf.spatial_shapes[0])
    value, shapes, _, loc, aw = encoder_msda_inputs(model.num_cam, h, w, 8, wf.hidden_dim // 8, 4, seed=0)
    feat = torch.randn(model.num_cam, wf.hidden_dim, *model.Rimg_shape)
    ops = {}
    with torch.no_grad():
        torch_oracle.msda_core(value, shapes, loc, aw)          # warm-up (allocator, thread pool)
    for tag, nthreads in (("all_cores", cores), ("1_thread", 1)):
        torch.set_num_threads(nthreads)
        with torch.no_grad():
            t0 = time.perf_counter()
            torch_oracle.msda_core(value, shapes, loc, aw)
            t1 = time.perf_counter()
            torch_oracle.warp_perspective(feat, proj_cpu, model.Rworld_shape)
            t2 = time.perf_counter()
        ops[tag] = {"threads": nthreads, "msda_core_s": round(t1 - t0, 3), "warp_s": round(t2 - t1, 3)}
    torch.set_num_threads(cores)
    res["ops"] = ops
    return res


def main():
    a = parse()
    from mvdetr_amd import dist as mdist
    rank, world, local_rank = mdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP kernels have no CPU fallback)")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True      # the reference's own setting (main.py:48): MIOpen picks convolutions by measurement
    if not a.no_gemm_tuning:
        # the same for the shadow transformer's fp32 GEMMs: TunableOp times the hipBLASLt / rocBLAS solutions of each
        # shape once (during the warm-up steps) and keeps the fastest; same arithmetic type, nothing is written to disk
        try:
            import tempfile
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.set_filename(os.path.join(tempfile.gettempdir(), f"mvdetr_bench_tunableop_{os.getpid()}.csv"))
            torch.cuda.tunable.write_file_on_exit(False)
            torch.cuda.tunable.set_max_tuning_duration(200)
        except Exception as ex:                    # pragma: no cover
            print(f"[bench] TunableOp unavailable: {ex}", file=sys.stderr)

    import mvdetr_amd.ops  # noqa: F401
    import MultiScaleDeformableAttention as MSDA
    from mvdetr_amd import geometry
    from mvdetr_amd.model import build_model

    MSDA.set_forward_impl(a.msda_impl)
    timer = KernelTimer(MSDA)      # callers look the functions up on the module at call time

    geom = geometry.GEOMETRIES[a.config]
    model = build_model(a.config, seed=0).to(dev).eval()
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

    for _ in range(a.warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = mdist.barrier_and_max(time.perf_counter() - t0, dev)
    timer.enabled = False
    k_us, k_n, k_bytes = timer.average_us()
    impl = MSDA.last_forward_impl()

    # ---- the hot path alone (warp + shadow transformer), same inputs ------------------------------------
    hot_ms = None
    if a.parallel == "dp":
        with torch.no_grad():
            feat = model.features(imgs)
            proj = model.frame_proj_mats(M).to(dev)
            for _ in range(3):
                model.hot_path(feat, proj)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                model.hot_path(feat, proj)
            torch.cuda.synchronize()
            hot_ms = (time.perf_counter() - t1) / a.steps * 1e3 / Bf

    if rank != 0:
        return
    alg_bytes = int(k_bytes) if k_bytes else None          # of the launches actually timed (rank 0's)
    full_frame = a.config == "wildtrack" and Bf == 1 and not (a.parallel == "views" and world > 1)
    traffic, traffic_src = load_traffic() if full_frame else (None, None)
    achieved = alg_bytes / (k_us * 1e-6) / 1e9 if (k_us and alg_bytes) else None
    res = {
        "metric": "multiview frames/s (7-cam Wildtrack) + MSDeformAttn HBM GB/s vs roofline",
        "value": round(frames_per_step * a.steps / elapsed, 3), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.config} {N}-cam frame, --world_feat deform_trans, ResNet18 trunk: "
                               f"{N}x3x{Hi}x{Wi} -> {geom.feat_channels}-ch world feat {geom.Rworld_shape[0]}x{geom.Rworld_shape[1]} "
                               f"-> BEV (BASELINE.json configs[1])" if a.config == "wildtrack" else f"{a.config} {N}-cam frame",
                   "frames_per_step": frames_per_step, "batch_per_rank": Bf, "parallelism": f"{a.parallel}{world}" + (f"-{a.encoder}" if a.parallel == "views" and world > 1 else ""),
                   "augment": bool(a.augment), "gemm_tuning": not a.no_gemm_tuning,
                   "weights": "seeded random"},
        "roofline": {"bound": "hbm", "kernel": f"msda_forward[{impl}]", "achieved": round(achieved, 1) if achieved else None,
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4) if achieved else None,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_us": round(k_us, 2) if k_us else None, "launches_timed": k_n},
        "hot_path": {"ms_per_frame": round(hot_ms, 3) if hot_ms else None,
                     "frames_per_s": round(1e3 / hot_ms, 1) if hot_ms else None,
                     "what": "warp_perspective + DeformTransWorldFeat (3 x MSDeformAttn), features resident"},
    }
    if world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(model, imgs[:1].cpu(), model.frame_proj_mats(M[:1]), a.cpu_budget_s)
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
