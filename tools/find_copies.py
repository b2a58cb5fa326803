#!/usr/bin/env python3
"""Where do the small torch launches of one cfg-2 conversion come from?  torch.profiler with stacks: device-to-device copies,
elementwise and other ATen kernels grouped by the rvc_amd source line that issued them."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from rvc_amd.infer.infer import VoiceConverter
from rvc_amd.lib import synthetic as S
dev = "cuda:0"
vc = VoiceConverter(device=dev)
vc.load_checkpoint_dict(S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0))
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
vc.vc.set_index(S.synth_index(100_000, seed=0))
audio = S.synth_audio(480_000, seed=0)
x = torch.from_numpy(audio).to(dev)
for _ in range(2): out = vc.convert_array(x, index_rate=0.75)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    out = vc.convert_array(x, index_rate=0.75)
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"): continue
    dt = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    if not ev.kernels: continue
    where = next((f for f in (ev.stack or []) if "rvc_amd" in f), "?")
    key = (ev.name, where.split("rvc_amd/")[-1][:70])
    by[key][0] += len(ev.kernels); by[key][1] += dt
tot_n = sum(v[0] for v in by.values()); tot_t = sum(v[1] for v in by.values())
print(f"ATen-issued device kernels in one conversion: {tot_n} launches, {tot_t/1e3:.2f} ms of device time")
for (name, where), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t/1e3:7.3f} ms {n:4d} x  {name:32s} {where}")
