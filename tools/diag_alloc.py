#!/usr/bin/env python3
"""Does the caching allocator reach a steady state under convert_batch?  cfg-2 utterances, `INFLIGHT` in flight (env, default 3): reserved
bytes and the number of device allocations (hipMalloc calls) after every batch of 20 utterances, ten batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd.infer.infer import VoiceConverter
from rvc_amd.lib import synthetic as S
dev = "cuda:0"
inflight = int(os.environ.get("INFLIGHT", "3"))
vc = VoiceConverter(device=dev)
vc.load_checkpoint_dict(S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0))
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
vc.vc.set_index(torch.from_numpy(S.synth_index(100_000, seed=0)).to(dev))
audios = [torch.from_numpy(S.synth_audio(480000, seed=i)).to(dev) for i in range(4)]
kw = dict(index_path="", index_rate=0.75, protect=0.5, sid=0)
prev = 0
for b in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vc.convert_batch([audios[j % 4] for j in range(20)], inflight=inflight, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m = torch.cuda.memory_stats(dev)
    n = m.get("num_device_alloc", 0)
    print(f"batch {b}: {dt / 20 * 1e3:6.2f} ms per utterance; reserved {m['reserved_bytes.all.current'] / 1e9:6.2f} GB, allocated peak {m['allocated_bytes.all.peak'] / 1e9:6.2f} GB, "
          f"device allocations so far {n} (+{n - prev}), frees {m.get('num_device_free', 0)}, inactive split {m['inactive_split_bytes.all.current'] / 1e9:5.2f} GB", flush=True)
    prev = n
