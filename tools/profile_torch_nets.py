#!/usr/bin/env python3
"""HuBERT / TextEncoder / flow / RMVPE-U-Net only (the PyTorch-ROCm driven parts) at the cfg-2 shape, for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch, torch.nn.functional as F
from rvc_amd import _native
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.hubert import HubertModelWithFinalProj
from rvc_amd.lib.algorithm.encoders import text_encoder
from rvc_amd.lib.algorithm.residuals import flow_reverse
from rvc_amd.lib.algorithm.weights import fold_weight_norm
from rvc_amd.lib.predictors.RMVPE import RMVPE0Predictor
dev = "cuda:0"
if os.environ.get("CUDNN_BENCHMARK"):
    torch.backends.cudnn.benchmark = True
which = sys.argv[1] if len(sys.argv) > 1 else "all"
hub = HubertModelWithFinalProj(S.make_hubert_state_dict(1), device=dev)
w = {k: v.to(dev) for k, v in fold_weight_norm(S.make_synth_checkpoint(48000, "HiFi-GAN", 0)["weight"]).items() if not k.startswith("dec.")}
rm = RMVPE0Predictor(device=dev, state_dict=S.make_rmvpe_state_dict(0))
wav = torch.randn(1, 512000, device=dev) * 0.1
T = 3198
phone = torch.randn(1, T, 768, device=dev); pitch = torch.randint(1, 255, (1, T), device=dev); lens = torch.tensor([T], device=dev)
g = torch.randn(1, 256, 1, device=dev); zp = torch.randn(1, 192, T, device=dev); mask = torch.ones(1, 1, T, device=dev)
mel = torch.randn(1, 128, 3232, device=dev)
import time
for it in range(3):
    if it == 2:
        torch.cuda.synchronize(); time.sleep(0.6)
    with torch.no_grad():
        if which in ("all", "hubert"): hub(wav)
        if which in ("all", "enc"): text_encoder(w, phone, pitch, lens)
        if which in ("all", "flow"): flow_reverse(w, zp, mask, g, full=True)
        if which in ("all", "rmvpe"): rm.mel2hidden(mel, 3201)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
with torch.no_grad():
    if which in ("all", "hubert"): hub(wav)
    if which in ("all", "enc"): text_encoder(w, phone, pitch, lens)
    if which in ("all", "flow"): flow_reverse(w, zp, mask, g, full=True)
    if which in ("all", "rmvpe"): rm.mel2hidden(mel, 3201)
e1.record(); torch.cuda.synchronize()
print(f"{which}: {e0.elapsed_time(e1):.2f} ms (one more pass, HIP events)")
