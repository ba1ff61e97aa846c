"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing: camera partitioning, the single
all-gather of per-view world tokens, and the bench's max-over-ranks timing rule."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvdetr_amd import dist as mdist


def test_partition_views():
    assert mdist.partition_views(7, 8) == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 7)]
    assert mdist.partition_views(16, 8) == [(2 * i, 2 * i + 2) for i in range(8)]
    assert mdist.partition_views(7, 2) == [(0, 4), (4, 7)]
    assert mdist.partition_views(6, 4) == [(0, 2), (2, 4), (4, 5), (5, 6)]
    assert mdist.partition_views(7, 1) == [(0, 7)]
    for n in range(1, 20):
        for w in range(1, 9):
            parts = mdist.partition_views(n, w)
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - s for s, e in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_views, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    try:
        r, w, _ = mdist.init_from_env()
        assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
        from mvdetr_amd.world_feat import DeformTransWorldFeat
        torch.manual_seed(0)
        B, C, H, W = 2, 16, 8, 12
        wf = DeformTransWorldFeat(num_views, (H, W), C, hidden_dim=C, nhead=2, dim_feedforward=32,
                                  reference_points=torch.zeros(num_views * (H // 2) * (W // 2), num_views, 4, 2))
        x = torch.randn(B, num_views, C, H, W, generator=torch.Generator().manual_seed(5))
        full, h, w_ = wf.tokens(x)                                     # every rank can build the unsharded answer
        s, e = mdist.partition_views(num_views, world)[rank]
        local = wf.tokens(x[:, s:e])[0] if e > s else torch.zeros(B, 0, C)
        gathered = mdist.all_gather_view_tokens(local, num_views)
        ok = gathered.shape == full.shape and torch.equal(gathered, full)
        # channel-last input takes the same route
        gathered2 = mdist.all_gather_view_tokens(
            wf.tokens(x[:, s:e].permute(0, 1, 3, 4, 2).contiguous())[0] if e > s else torch.zeros(B, 0, C), num_views)
        ok = ok and torch.allclose(gathered2, full, atol=1e-6)
        # timing rule: the slowest rank defines the step time
        t = mdist.barrier_and_max(0.25 * (rank + 1), torch.device("cpu"))
        ok = ok and abs(t - 0.25 * world) < 1e-12
        q.put((rank, bool(ok), ""))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("num_views", [7, 2, 1])
def test_view_sharded_all_gather_world2(num_views):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_views, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in results:
        assert ok, f"rank {rank}: {msg}"


# ---- query-sharded encoder ---------------------------------------------------
# On the CPU the attention core is the library's own host path (csrc/host_path.cpp), i.e. these tests run the product
# end to end; `core="oracle"` swaps the extension's forward entry for the oracle (test infrastructure) as a cross-check
# of the schedule that is independent of the library.

def _oracle_core_patch(monkeypatch=None):
    """monkeypatch=None only inside a spawned worker process (nothing to restore there)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import torch_oracle
    from mvdetr_amd.ops import MultiScaleDeformableAttention as MSDA

    def forward(value, shapes, lsi, loc, aw, im2col_step):
        return torch_oracle.msda_core(value, shapes, loc, aw)
    if monkeypatch is None:
        MSDA.ms_deform_attn_forward = forward
    else:
        monkeypatch.setattr(MSDA, "ms_deform_attn_forward", forward)


def _small_world_feat(num_views, C=16, H=8, W=12, seed=0):
    from mvdetr_amd.world_feat import DeformTransWorldFeat
    torch.manual_seed(seed)
    h, w = H // 2, W // 2
    ref = torch.rand(num_views * h * w, num_views, 4, 2, generator=torch.Generator().manual_seed(3))
    wf = DeformTransWorldFeat(num_views, (H, W), C, hidden_dim=C, nhead=2, dim_feedforward=32, reference_points=ref)
    # the zero-initialised offset/attention weights would make every query of a camera behave alike
    for layer in wf.encoder.layers:
        torch.nn.init.normal_(layer.self_attn.sampling_offsets.weight, std=0.3)
        torch.nn.init.normal_(layer.self_attn.attention_weights.weight, std=0.3)
    return wf.eval(), h, w


@pytest.mark.parametrize("core", ["host", "oracle"])
@pytest.mark.parametrize("num_views,world", [(7, 3), (7, 8), (6, 4), (3, 1), (16, 8)])
def test_query_sharded_fusion_lockstep(num_views, world, core, monkeypatch):
    """Emulated ranks in one process: per-layer value exchange + partial merge == unsharded fuse()."""
    if core == "oracle":
        _oracle_core_patch(monkeypatch)
    wf, h, w = _small_world_feat(num_views)
    B, C = 2, wf.hidden_dim
    tokens = torch.randn(B, num_views * h * w, C, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        want = wf.fuse(tokens, B, h, w)
        ranks = [mdist.QueryShardedFusion(wf, r, world) for r in range(world)]
        src = [tokens[:, rk.own_slice(h, w)] for rk in ranks]
        for i in range(wf.encoder.num_layers):
            value = torch.cat([rk.layer_value(i, s) for rk, s in zip(ranks, src)], dim=1)
            assert value.shape == tokens.shape
            src = [rk.layer_update(i, s, value, h, w) for rk, s in zip(ranks, src)]
        total = sum(rk.merge_partial(s, B, h, w) for rk, s in zip(ranks, src))
        got = ranks[0].merge_finish(total)
    assert got.shape == want.shape
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5), float((got - want).abs().max())


def _sharded_worker(rank, world, port, num_views, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MVDETR_HOST_THREADS="2")
    try:
        mdist.init_from_env()                 # (the attention core is the library's host path: nothing is patched)
        wf, h, w = _small_world_feat(num_views)
        B, C = 2, wf.hidden_dim
        tokens = torch.randn(B, num_views * h * w, C, generator=torch.Generator().manual_seed(9))
        with torch.no_grad():
            want = wf.fuse(tokens, B, h, w)
        fusion = mdist.QueryShardedFusion(wf, rank, world)
        got = fusion(tokens[:, fusion.own_slice(h, w)].contiguous(), B, h, w)
        ok = got.shape == want.shape and torch.allclose(got, want, atol=2e-5, rtol=1e-5)
        q.put((rank, bool(ok), "" if ok else f"max err {float((got - want).abs().max())}"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("num_views", [7, 1])
def test_query_sharded_fusion_world2(num_views):
    """The same through real collectives (gloo, world size 2; with one view rank 1 is idle)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, num_views, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in results:
        assert ok, f"rank {rank}: {msg}"


# ---- world size 8 through real collectives (only world 2 had ever run them) -----------------------
def _run_world(target, world, args, timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in sorted(results):
        assert ok, f"rank {rank}: {msg}"
    return results


@pytest.mark.parametrize("num_views", [7, 16])
def test_view_sharded_all_gather_world8(num_views):
    """7 views over 8 ranks (rank 7 idle: a zero-length shard in the padded all-gather) and 16 views, 2 per rank."""
    _run_world(_worker, 8, (num_views,))


@pytest.mark.parametrize("num_views", [7, 16])
def test_query_sharded_fusion_world8(num_views):
    _run_world(_sharded_worker, 8, (num_views,))


def _frame_worker(rank, world, port, num_views, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MVDETR_HOST_THREADS="1", OMP_NUM_THREADS="1")
    try:
        import dataclasses
        from mvdetr_amd import geometry
        from mvdetr_amd.model import build_model
        torch.set_num_threads(1)
        mdist.init_from_env()
        name = f"mini{num_views}"
        geometry.GEOMETRIES[name] = dataclasses.replace(geometry.MINI, name=name, num_cam=num_views)
        model = build_model(name, seed=0).eval()                          # CPU: the library's host path end to end
        with torch.no_grad():
            for i, layer in enumerate(model.world_feat.encoder.layers):
                layer.self_attn.sampling_offsets.weight.normal_(0, 0.02, generator=torch.Generator().manual_seed(10 + i))
                layer.self_attn.attention_weights.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(20 + i))
        Hi, Wi = geometry.MINI.input_img_shape
        imgs = torch.randn(1, num_views, 3, Hi, Wi, generator=torch.Generator().manual_seed(3))
        M = geometry.random_affine_mats(1, num_views, (Hi, Wi), seed=2, translate=0.05, scale=(0.9, 1.1))
        msgs = []
        with torch.no_grad():
            want = model(imgs, M)[0]
            for encoder in ("sharded", "replicated"):
                runner = mdist.ViewShardedFrame(model, encoder=encoder)
                s, e = runner.range
                assert (s, e) == mdist.partition_views(num_views, world)[rank]
                # count the collectives of one frame: replicated = ONE token all-gather and nothing else (no all-reduce to
                # agree on the token grid, also with an idle rank); sharded = one value all-gather per encoder layer + the
                # all-reduce of the merge convolution's partial sums
                calls = {"all_reduce": 0, "all_gather_into_tensor": 0}
                real = {k: getattr(dist, k) for k in calls}

                def counted(name):
                    def f(*a, **kw):
                        calls[name] += 1
                        return real[name](*a, **kw)
                    return f
                for k in calls:
                    setattr(dist, k, counted(k))
                try:
                    got = runner(imgs[:, s:e], M)
                finally:
                    for k in calls:
                        setattr(dist, k, real[k])
                n_layers = len(model.world_feat.encoder.layers)
                want_calls = {"all_reduce": 0, "all_gather_into_tensor": 1} if encoder == "replicated" else \
                             {"all_reduce": 1, "all_gather_into_tensor": n_layers}
                assert calls == want_calls, (encoder, calls, want_calls)
                err = max((got[0] - want[0]).abs().max().item(), (got[1] - want[1]).abs().max().item())
                msgs.append(f"{encoder} {err:.2e} {calls}")
                assert err < 5e-5, msgs
        q.put((rank, True, "; ".join(msgs)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("num_views", [7, 16])
def test_view_sharded_frame_world8_both_encoder_modes(num_views):
    """Whole frames (trunk, warp, token conv per view; then the query-sharded or the replicated shadow transformer) over 8
    gloo ranks equal the single-process frame: 7 views -> rank 7 idles through every collective, 16 views -> 2 each."""
    _run_world(_frame_worker, 8, (num_views,), timeout=600)
