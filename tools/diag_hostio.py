#!/usr/bin/env python3
"""Is the host-array boundary (pinned double slots, asynchronous copies) stable?  Alternates resident / host-array batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import numpy as np, torch
from rvc_amd.infer.infer import VoiceConverter
from rvc_amd.lib import synthetic as S
dev = "cuda:0"
vc = VoiceConverter(device=dev)
vc.load_checkpoint_dict(S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0))
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
vc.vc.set_index(torch.from_numpy(S.synth_index(100_000, seed=0)).to(dev))
host = [S.synth_audio(480000, seed=i) for i in range(4)]
res = [torch.from_numpy(a).to(dev) for a in host]
kw = dict(index_path="", index_rate=0.75, protect=0.5, sid=0)
n = int(os.environ.get("N", "8"))
vc.convert_batch(res * 1, inflight=2, **kw); vc.convert_batch(host * 1, inflight=2, **kw)
for rep in range(4):
    for name, pool in (("resident", res), ("host", host)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = vc.convert_batch([pool[j % 4] for j in range(n)], inflight=2, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"rep {rep} {name:8s}: {dt / n * 1e3:6.2f} ms / utterance", flush=True)
