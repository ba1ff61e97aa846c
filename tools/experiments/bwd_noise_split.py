"""Which kernel of the public backward pays for wider offsets: run under rocprofv3 --kernel-trace (tools/gpu_bwd_noise.sh)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mvdetr_amd.ops  # noqa
import MultiScaleDeformableAttention as MSDA
from helpers import encoder_msda_inputs
s = float(sys.argv[1])
v, sh, lsi, loc, aw = [x.cuda() for x in encoder_msda_inputs(7, 60, 180, 8, 16, 4, B=1, seed=0, noise_px=s)]
go = torch.randn(1, 75600, 128, device="cuda")
for _ in range(8):
    MSDA.ms_deform_attn_backward(v, sh, lsi, loc, aw, go, 64)
torch.cuda.synchronize()
