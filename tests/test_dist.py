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
