"""``Synthesizer`` with the reference's constructor and ``infer`` signature
(rvc/lib/algorithm/synthesizers.py:40-66, 223-260), driving the TextEncoder and flow on PyTorch-ROCm and
the vocoder through librvc_amd's HIP kernels.

It follows the call sequence of ``VoiceConverter.setup_network`` (rvc/infer/infer.py:476-485) unchanged:

    net_g = Synthesizer(*cpt["config"], use_f0=..., text_enc_hidden_dim=..., vocoder=...)
    del net_g.enc_q
    net_g.load_state_dict(cpt["weight"], strict=False)
    net_g = net_g.to(device).float()
    net_g.eval()

Noise (SURVEY §7 hard part 1) -- ``infer(..., noise=...)``:
    None    draw on the device with torch's HIP generator (the reference's behaviour on a GPU)
    "cpu"   draw from torch's CPU generator in the reference's order and shapes, then upload: with equal
            seeds the output matches the reference's CPU path (parity mode)
    dict    explicit tensors: {"z": [B,192,T], "src_rand": ..., "src_randn": ...}
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from rvc_amd.lib.algorithm.encoders import prepare_attention_weights, text_encoder
from rvc_amd.lib.algorithm.residuals import flow_reverse, prepare_flow_weights
from rvc_amd.lib.algorithm.weights import fold_weight_norm


class Synthesizer:
    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, spk_embed_dim, gin_channels, sr,
                 use_f0, text_enc_hidden_dim=768, vocoder="HiFi-GAN", randomized=True, checkpointing=False, **kwargs):
        if not use_f0:
            # the reference's non-f0 generator is dead code in this fork
            raise NotImplementedError("models without pitch guidance are not supported (nor by the reference fork)")
        self.inter_channels, self.hidden_channels = inter_channels, hidden_channels
        self.n_heads, self.n_layers, self.kernel_size = n_heads, n_layers, kernel_size
        self.resblock_kernel_sizes = list(resblock_kernel_sizes)
        self.resblock_dilations = list(resblock_dilation_sizes[0])
        self.upsample_rates, self.upsample_kernel_sizes = list(upsample_rates), list(upsample_kernel_sizes)
        self.upsample_initial_channel = upsample_initial_channel
        self.gin_channels, self.sr, self.vocoder = gin_channels, int(sr), vocoder
        self.use_f0 = use_f0
        self.upp = int(np.prod(upsample_rates))
        self.enc_q = None  # training-only posterior encoder; the reference deletes it (infer.py:482)
        self.device = torch.device("cpu")
        self.w: Dict[str, torch.Tensor] = {}
        self._dec_weights: Optional[Dict[str, torch.Tensor]] = None
        self.dec = None
        self.dec_weight_dtype = "f32"   # "bf16": BASELINE cfg 4, the vocoder's weights are bf16 values

    # ---- nn.Module-like surface used by the reference's loader ----
    def load_state_dict(self, state_dict, strict: bool = False):
        w = fold_weight_norm({k: v for k, v in state_dict.items() if not k.startswith("enc_q.")})
        self._dec_weights = {k[4:]: v for k, v in w.items() if k.startswith("dec.")}
        if self.dec_weight_dtype == "bf16":   # after weight-norm folding: what a bf16 copy of the folded tensor holds
            self._dec_weights = {k: v.float().bfloat16().float() for k, v in self._dec_weights.items()}
        self.w = {k: v for k, v in w.items() if not k.startswith("dec.")}
        self.dec = None
        if self.device.type == "cuda":
            self._to_device()
        return self

    def _to_device(self):
        from rvc_amd import _native
        self.w = {k: v.to(self.device) for k, v in self.w.items()}
        # derived tensors are built HERE, once, on the loading thread: forwards run concurrently on several host threads
        # and streams (VoiceConverter.convert_batch) and only ever read self.w
        prepare_attention_weights(self.w, self.n_layers)
        prepare_flow_weights(self.w)
        torch.cuda.synchronize(self.device)   # visible to every stream that will use them
        if self._dec_weights is not None and self.dec is None:
            with torch.cuda.device(self.device):
                self.dec = _native.Decoder(self.vocoder, self.sr, self._dec_weights, in_channels=self.inter_channels,
                                           upsample_initial_channel=self.upsample_initial_channel,
                                           gin_channels=self.gin_channels, upsample_rates=self.upsample_rates,
                                           upsample_kernel_sizes=self.upsample_kernel_sizes,
                                           res_kernel_sizes=self.resblock_kernel_sizes,
                                           res_dilations=self.resblock_dilations, weight_storage=self.dec_weight_dtype)

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("rvc_amd.Synthesizer runs on a HIP device only: the vocoder has no CPU path "
                               "(the CPU restatement lives in oracle/ and is test infrastructure)")
        self.device = device
        self._to_device()
        return self

    def float(self):
        return self

    def eval(self):
        return self

    def remove_weight_norm(self):  # folded at load
        return self

    # ---- noise ----
    def _adain_shapes(self, b, t):
        """RefineGAN's 24 AdaIN draws, in call order (refinegan.py:111 via :155-162, :168-170)."""
        shapes, length, ch = [], t, self.upsample_initial_channel
        for r in self.upsample_rates:
            length *= r
            ch //= 2
            shapes += [(b, ch, length)] * (2 * len(self.resblock_kernel_sizes))
        return shapes

    def _draw(self, noise, b, t, t_dec=None):
        """Every random tensor of one infer call, in the reference's order.  ``t_dec``: frames the vocoder sees when
        ``rate`` cuts the head off (synthesizers.py:247-251) -- z is drawn for all ``t`` frames BEFORE the cut, the
        vocoder's draws happen after it, at the shorter length."""
        t_dec = t if t_dec is None else t_dec
        L = t_dec * self.upp
        dim = 9 if self.vocoder == "MRF HiFi-GAN" else 1
        refine = self.vocoder == "RefineGAN"
        dev = self.device
        if isinstance(noise, dict):
            return {k: (v.to(dev) if v is not None else None) for k, v in noise.items()}
        if noise == "cpu":
            z = torch.randn(b, self.inter_channels, t)                        # synthesizers.py:245
            if self.vocoder == "HiFi-GAN":
                src_rand = torch.rand(1, 1, 1)                                # hifigan.py:189 (zeroed, consumes state)
                src_randn = torch.randn(b, L, 1)                              # hifigan.py:223
            else:
                src_rand = torch.rand(b, dim)                                 # hifigan_mrf.py:143 / refinegan.py:229
                src_randn = torch.randn(b, L, dim)                            # hifigan_mrf.py:172 / refinegan.py:258
            out = {"z": z.to(dev), "src_rand": src_rand.to(dev), "src_randn": src_randn.to(dev)}
            if refine:
                out["adain_randn"] = torch.cat([torch.randn(*sh).reshape(-1) for sh in self._adain_shapes(b, t_dec)]).to(dev)
            return out
        out = {"z": torch.randn(b, self.inter_channels, t, device=dev),
               "src_rand": torch.rand(b, dim, device=dev),
               "src_randn": torch.randn(b, L, dim, device=dev)}
        if refine:
            out["adain_randn"] = torch.randn(sum(int(np.prod(sh)) for sh in self._adain_shapes(b, t_dec)), device=dev)
        return out

    @torch.no_grad()
    def infer(self, phone, phone_lengths, pitch=None, nsff0=None, sid=None, rate=None, noise=None, *,
              phone_lengths_host=None):
        if self.dec is None:
            raise RuntimeError("Synthesizer.infer before load_state_dict(...).to('cuda:N')")
        w = self.w
        g = F.embedding(sid, w["emb_g.weight"]).unsqueeze(-1)
        m_p, logs_p, x_mask = text_encoder(w, phone, pitch, phone_lengths, hidden=self.hidden_channels,
                                           out_channels=self.inter_channels, n_heads=self.n_heads,
                                           n_layers=self.n_layers, kernel_size=self.kernel_size,
                                           lengths_host=phone_lengths_host)
        b, _, t = m_p.shape
        head = int(t * (1.0 - rate.item())) if rate is not None else 0   # synthesizers.py:247-251
        nz = self._draw(noise, b, t, t - head)
        z_p = (m_p + torch.exp(logs_p) * nz["z"] * 0.66666) * x_mask
        if rate is not None:
            z_p, x_mask = z_p[:, :, head:], x_mask[:, :, head:]
            nsff0 = nsff0[:, head:]
        full = phone_lengths_host is not None and rate is None and all(int(n) == t for n in phone_lengths_host)
        z = flow_reverse(w, z_p, x_mask, g, half=self.inter_channels // 2, hidden=self.hidden_channels, full=full)
        o = self.dec.forward((z * x_mask).contiguous(), nsff0.float().contiguous(), g[:, :, 0].contiguous(),
                             src_randn=nz["src_randn"].contiguous(), src_rand=nz.get("src_rand"),
                             adain_randn=nz.get("adain_randn"))
        return o, x_mask, (z, z_p, m_p, logs_p)
