import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from rvc_amd import _native
from rvc_amd.infer.infer import VoiceConverter
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.algorithm.synthesizers import Synthesizer
real_draw = Synthesizer._draw
def zero_draw(self, noise, b, t, t_dec=None):
    return {k: (torch.zeros_like(v) if v is not None else None) for k, v in real_draw(self, noise, b, t, t_dec).items()}
Synthesizer._draw = zero_draw
DEV = "cuda:0"
vc = VoiceConverter(device=DEV)
vc.load_checkpoint_dict(S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0, smooth_pitch=bool(int(os.environ.get("PEAKED", "1")))))
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0, peaked=bool(int(os.environ.get("PEAKED", "1")))))
vc.vc.set_index(S.synth_index(20_000, seed=0))
audios = [torch.from_numpy(S.synth_audio(n, seed=i)).to(DEV) for i, n in enumerate((48_000, 80_000, 64_000, 48_000, 100_000))]
def rms(x): return float(np.sqrt(np.mean(np.square(x.astype(np.float64)))))
taps = {}
vc.vc.debug_taps = None
seq1 = [vc.convert_array(a, index_rate=0.75).clone() for a in audios]
torch.cuda.synchronize()
def report(tag, outs):
    d = [(a - b).cpu().numpy() for a, b in zip(seq1, outs)]
    print(tag, [f"{rms(x):.1e}" for x in d], "first idx > 1e-5:", [int(np.argmax(np.abs(x) > 1e-5)) if (np.abs(x) > 1e-5).any() else -1 for x in d], flush=True)
for r in range(3):
    report("seq   ", [vc.convert_array(a, index_rate=0.75).clone() for a in audios])
for r in range(4):
    for inflight in (2, 3):
        par = vc.convert_batch(audios, inflight=inflight, index_rate=0.75)
        torch.cuda.synchronize()
        report(f"par {inflight} ", par)
