#!/usr/bin/env python3
"""One warm BASELINE cfg-2 utterance through Pipeline.pipeline for rocprofv3 --kernel-trace: two warm-up utterances,
an idle gap (so tools/summarize_trace.py ... lastgap isolates the last one), then the profiled utterance."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd.infer.infer import VoiceConverter
from rvc_amd.lib import synthetic as S

dev = "cuda:0"
vc = VoiceConverter(device=dev)
vc.load_checkpoint_dict(S.make_synth_checkpoint(48000, sys.argv[1] if len(sys.argv) > 1 else "HiFi-GAN", seed=0))
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
vc.vc.set_index(S.synth_index(100_000, seed=0))
audio = torch.from_numpy(S.synth_audio(480_000, seed=0)).to(dev)
for _ in range(2):
    vc.convert_array(audio, index_rate=0.75)
torch.cuda.synchronize(); time.sleep(0.6)
t0 = time.perf_counter()
vc.convert_array(audio, index_rate=0.75)
torch.cuda.synchronize()
print(f"profiled utterance: {1e3 * (time.perf_counter() - t0):.2f} ms")
