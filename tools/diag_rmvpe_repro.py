#!/usr/bin/env python3
"""Is the RMVPE path bit-reproducible ACROSS processes?  Prints checksums of the U-Net features (gi), the salience and the f0 contour of
one fixed 30 s clip, evaluated three times in this process; run it in several fresh processes and compare the lines."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import numpy as np, torch
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.predictors.RMVPE import RMVPE0Predictor
dev = "cuda:0"
pred = RMVPE0Predictor(device=dev, state_dict=S.make_rmvpe_state_dict(0))
a = torch.from_numpy(S.synth_audio(480000, seed=0)).float().to(dev)
h = lambda t: hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]
for rep in range(3):
    gi, n = pred.front_half_device(a)
    sal = pred.gru_head(gi, n)[0]
    f0 = pred.decode(sal)
    print(f"pid {os.getpid()} rep {rep}: gi {h(gi)} salience {h(sal)} f0 {h(f0)}", flush=True)
