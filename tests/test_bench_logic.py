"""CPU checks of bench.py's host logic: the self-launch command for --gpus N, and that the seeded stand-in for learned
sampling offsets really produces the spread SURVEY 8d names (bias grid + ~N(0, 1 px)) on the model's own tokens."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_n_relaunches_under_torchrun(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    rc = bench.relaunch_under_torchrun(types.SimpleNamespace(gpus=4))
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"      # torchrun picks the port itself
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]      # the ranks see the same flags
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    if torch.cuda.device_count() < 4:
        assert seen["env"]["MVDETR_DIST_BACKEND"] == "gloo"      # fewer GPUs than ranks: RCCL needs one GPU per rank


def test_perturbed_projections_give_about_one_pixel_of_offset_spread():
    from mvdetr_amd.world_feat import DeformTransWorldFeat
    torch.manual_seed(0)
    N, H, W, C = 3, 16, 24, 128
    h, w = H // 2, W // 2
    ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
    ref = torch.stack([xs / w, ys / h], -1).reshape(-1, 1, 1, 2).repeat(N, N, 4, 1)
    wf = DeformTransWorldFeat(N, (H, W), C, hidden_dim=C, reference_points=ref).eval()
    assert bench.perturb_sampling(types.SimpleNamespace(world_feat=wf), 1.0) == 1.0
    spreads, logit_spreads = [], []
    hooks = []
    for layer in wf.encoder.layers:
        at = layer.self_attn
        hooks.append(at.sampling_offsets.register_forward_hook(
            lambda m, i, o: spreads.append(float((o - m.bias).std()))))
        hooks.append(at.attention_weights.register_forward_hook(
            lambda m, i, o: logit_spreads.append(float((o - m.bias).std()))))
    x = torch.randn(1, N, C, H, W)
    with torch.no_grad():
        wf(x)                                                     # CPU tensors: the library's host path
    for hk in hooks:
        hk.remove()
    assert len(spreads) == 3
    # (pixels: the module divides the raw offsets by the level size, so the Linear's output IS in pixels)
    assert all(0.6 < s < 1.6 for s in spreads), spreads
    assert all(0.4 < s < 1.6 for s in logit_spreads), logit_spreads
    # and with std 0 the reference's zero-initialised projections are left alone
    wf2 = DeformTransWorldFeat(N, (H, W), C, hidden_dim=C, reference_points=ref)
    assert bench.perturb_sampling(types.SimpleNamespace(world_feat=wf2), 0.0) is None
    assert float(wf2.encoder.layers[0].self_attn.sampling_offsets.weight.detach().abs().max()) == 0.0


def test_calibration_brings_every_layers_offset_spread_to_the_target():
    """bench.calibrate_sampling: layer by layer, the learned part of the offsets gets the spread SURVEY 8d names (1 px) whatever the
    scale of the queries -- the first layer's are the token convolution's output, whose scale follows the trunk (a ResNet-50
    trunk gave 6.4 px with the uncalibrated projection)."""
    from mvdetr_amd.world_feat import DeformTransWorldFeat
    torch.manual_seed(0)
    N, H, W, C = 3, 16, 24, 128
    h, w = H // 2, W // 2
    ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
    ref = torch.stack([xs / w, ys / h], -1).reshape(-1, 1, 1, 2).repeat(N, N, 4, 1)
    wf = DeformTransWorldFeat(N, (H, W), C, hidden_dim=C, reference_points=ref).eval()
    bench.perturb_sampling(types.SimpleNamespace(world_feat=wf), 1.0)
    x = 5.0 * torch.randn(1, N, C, H, W)                          # features five times the scale the perturbation assumes
    layers = [layer.self_attn for layer in wf.encoder.layers]

    def run():
        with torch.no_grad():
            wf(x)

    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None                 # (CPU run)
    try:
        report = bench.calibrate_sampling(layers, run, 1.0)
    finally:
        torch.cuda.synchronize = real_sync
    assert len(report) == 3 and report[0]["offset_std_px_before"] > 2.0      # the first layer WAS far off
    spreads = []
    hooks = [at.sampling_offsets.register_forward_hook(lambda m, i, o: spreads.append(float((o - m.bias).std()))) for at in layers]
    run()
    for hk in hooks:
        hk.remove()
    assert all(abs(s - 1.0) < 0.05 for s in spreads), spreads
