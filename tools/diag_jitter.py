#!/usr/bin/env python3
"""Where do the slow 8-step bench runs come from?  One process, cfg 2, two utterances in flight: 12 timed groups of 8 utterances each,
wall time per group, with and without Python's cyclic GC."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd.infer.infer import VoiceConverter
from rvc_amd.lib import synthetic as S
dev = "cuda:0"
vc = VoiceConverter(device=dev)
vc.load_checkpoint_dict(S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0))
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
vc.vc.set_index(torch.from_numpy(S.synth_index(100000, seed=0)).to(dev))
audios = [torch.from_numpy(S.synth_audio(480000, seed=j)).to(dev) for j in range(4)]
kw = dict(index_path="", index_rate=0.75, protect=0.5, sid=0)
run = lambda n: vc.convert_batch([audios[j % 4] for j in range(n)], inflight=2, **kw)
run(10)
for label in ("gc on", "gc off", "gc on"):
    if label == "gc off":
        gc.collect(); gc.disable()
    else:
        gc.enable()
    ts = []
    for _ in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(8)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 8 * 1e3)
    print(f"{label}: ms per utterance in groups of 8: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
