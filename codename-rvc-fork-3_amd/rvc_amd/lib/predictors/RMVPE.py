"""RMVPE F0 estimator on the GPU (rvc/lib/predictors/RMVPE.py).

* log-mel front end  -> librvc_amd ``rvc_logmel_rmvpe`` (HIP; RMVPE.py:342-417, reflect pad of :452-455 fused)
* E2E network        -> PyTorch-ROCm, functional over the reference's state dict (RMVPE.py:289-339), BatchNorm
                        (eval) folded into the convolutions at load
* salience decode    -> vectorised on the device in float64 (RMVPE.py:459-512 runs a Python loop per frame)
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

N_MELS, N_CLASS = 128, 360


def _fold_bn(conv_w: torch.Tensor, sd, bn: str, out_dim: int = 0, eps: float = 1e-5):
    scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + eps)
    shift = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * scale
    shape = [1] * conv_w.dim()
    shape[out_dim] = -1
    return (conv_w.double() * scale.view(shape)).float(), shift.float()


class RMVPE0Predictor:
    def __init__(self, model_path=None, device=None, state_dict: Dict[str, torch.Tensor] | None = None):
        self.device = torch.device(device if device is not None else "cuda:0")
        self.w: Dict[str, torch.Tensor] = {}
        if state_dict is None and model_path is not None:
            state_dict = torch.load(model_path, map_location="cpu", weights_only=True)
        if state_dict is not None:
            self.load_state_dict(state_dict)
        cents = 20 * np.arange(N_CLASS) + 1997.3794084376191  # RMVPE.py:441-442
        self.cents_mapping = torch.from_numpy(np.pad(cents, (4, 4))).to(self.device)

    def load_state_dict(self, sd):
        w = {}
        sd = {k: v.float() if v.is_floating_point() else v for k, v in sd.items()}

        def block(p):
            w[p + ".c1.w"], w[p + ".c1.b"] = _fold_bn(sd[p + ".conv.0.weight"], sd, p + ".conv.1")
            w[p + ".c2.w"], w[p + ".c2.b"] = _fold_bn(sd[p + ".conv.3.weight"], sd, p + ".conv.4")
            if p + ".shortcut.weight" in sd:
                w[p + ".sc.w"], w[p + ".sc.b"] = sd[p + ".shortcut.weight"], sd[p + ".shortcut.bias"]

        bn = "unet.encoder.bn"
        s = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
        w["in.scale"] = s.float()
        w["in.shift"] = (sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * s).float()
        for i in range(5):
            for m in range(4):
                block(f"unet.encoder.layers.{i}.conv.{m}")
        for i in range(4):
            for m in range(4):
                block(f"unet.intermediate.layers.{i}.conv.{m}")
        for i in range(5):
            p = f"unet.decoder.layers.{i}"
            w[p + ".up.w"], w[p + ".up.b"] = _fold_bn(sd[p + ".conv1.0.weight"], sd, p + ".conv1.1", out_dim=1)
            for m in range(4):
                block(f"{p}.conv2.{m}")
        w["cnn.w"], w["cnn.b"] = sd["cnn.weight"], sd["cnn.bias"]
        for sfx in ("", "_reverse"):
            for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                w["gru." + n + sfx] = sd["fc.0.gru." + n + sfx]
        w["fc.w"], w["fc.b"] = sd["fc.1.weight"], sd["fc.1.bias"]
        self.w = {k: v.to(self.device).contiguous() for k, v in w.items()}
        # K10 (conv2d.hip) taps of every 3x3 / 1x1 block conv whose input channel count the kernel takes (all but the 1 -> 16 one)
        self.wp = {}
        if self.device.type == "cuda":
            from rvc_amd import _native
            for k, v in w.items():
                if k.endswith((".c1.w", ".c2.w", ".sc.w", "cnn.w")) and v.shape[1] % 8 == 0:
                    self.wp[k] = _native.conv2d_pack_weight(v, self.device)
        # K10b (conv2dbf.hip): the 3x3 convs as exact bf16x3 products on the bf16 matrix cores (every level of the U-Net; the 1x1 shortcuts
        # and shapes it does not take stay on K10)
        self.wb = {}
        if self.device.type == "cuda":
            for k, v in w.items():
                if k.endswith((".c1.w", ".c2.w", "cnn.w")) and v.shape[-1] == 3 and _native.conv2d_bf16x3_packable(v.shape[1], v.shape[0]):
                    self.wb[k] = _native.conv2d_bf16x3_pack_weight(v, self.device)
        # BiGRU: one GEMM for the input projections of both directions, recurrence in librvc_amd (gru.hip)
        self.w["gru.wih"] = torch.cat([self.w["gru.weight_ih_l0"], self.w["gru.weight_ih_l0_reverse"]], 0).contiguous()
        self.w["gru.bih"] = torch.cat([self.w["gru.bias_ih_l0"], self.w["gru.bias_ih_l0_reverse"]], 0).contiguous()
        self.w["gru.whhT"] = torch.stack([self.w["gru.weight_hh_l0"].t(), self.w["gru.weight_hh_l0_reverse"].t()], 0).contiguous()
        self.w["gru.bhh"] = torch.stack([self.w["gru.bias_hh_l0"], self.w["gru.bias_hh_l0_reverse"]], 0).contiguous()
        return self

    def _conv3(self, x, name, c_out, relu=False, res=None):
        """One 3x3 conv + folded-BN bias (+ ReLU, + skip path): K10b where it takes the shape, else K10."""
        from rvc_amd import _native
        u = self.wb.get(name + ".w")
        if u is not None and _native.conv2d_bf16x3_supported(x.shape[1], c_out, x.shape[2], x.shape[3]):
            return _native.conv2d_bf16x3_forward(x, u, self.w[name + ".b"], c_out, relu=relu, res=res)
        return _native.conv2d_forward(x, self.wp[name + ".w"], self.w[name + ".b"], c_out, 3, relu=relu, res=res)

    def _block(self, x, p):
        w = self.w
        if x.is_cuda and p + ".c1.w" in self.wp and x.shape[-1] in (4, 8, 16, 32, 64, 128):
            # the whole block in librvc_amd: conv + folded-BN bias + ReLU (+ skip path) per launch (conv2d.hip)
            from rvc_amd import _native
            c_mid, c_out = w[p + ".c1.w"].shape[0], w[p + ".c2.w"].shape[0]
            x = x.contiguous()
            y = self._conv3(x, p + ".c1", c_mid, relu=True)
            res = _native.conv2d_forward(x, self.wp[p + ".sc.w"], w[p + ".sc.b"], c_out, 1) if p + ".sc.w" in self.wp else x
            return self._conv3(y, p + ".c2", c_out, relu=True, res=res)
        if x.is_cuda and x.shape[-1] * x.shape[-2] % 4 == 0:
            # conv by MIOpen; bias + ReLU (+ residual) as ONE pass of librvc_amd K8 instead of three PyTorch launches
            from rvc_amd import _native
            y = _native.bias_relu_add_(F.conv2d(x, w[p + ".c1.w"], None, 1, 1), w[p + ".c1.b"])
            res = F.conv2d(x, w[p + ".sc.w"], w[p + ".sc.b"]) if p + ".sc.w" in w else x
            return _native.bias_relu_add_(F.conv2d(y, w[p + ".c2.w"], None, 1, 1), w[p + ".c2.b"], res.contiguous())
        y = F.relu(F.conv2d(x, w[p + ".c1.w"], w[p + ".c1.b"], 1, 1))
        y = F.relu(F.conv2d(y, w[p + ".c2.w"], w[p + ".c2.b"], 1, 1))
        if p + ".sc.w" in w:
            return y + F.conv2d(x, w[p + ".sc.w"], w[p + ".sc.b"])
        return y + x

    @torch.no_grad()
    def mel2hidden(self, mel: torch.Tensor, n_frames: int) -> torch.Tensor:
        """mel [B,128,T32] (frame axis already reflect-padded to a multiple of 32) -> salience [B,n_frames,360]."""
        return self.gru_head(self.unet_features(mel), n_frames)

    @torch.no_grad()
    def unet_features(self, mel: torch.Tensor) -> torch.Tensor:
        """The throughput-bound half: U-Net + output conv + the GRU's input projections -> gi [B,T32,2,768]."""
        w = self.w
        x = mel.transpose(-1, -2).unsqueeze(1)
        x = x * w["in.scale"] + w["in.shift"]
        skips = []
        for i in range(5):
            for m in range(4):
                x = self._block(x, f"unet.encoder.layers.{i}.conv.{m}")
            skips.append(x)
            x = F.avg_pool2d(x, (2, 2))
        for i in range(4):
            for m in range(4):
                x = self._block(x, f"unet.intermediate.layers.{i}.conv.{m}")
        for i in range(5):
            p = f"unet.decoder.layers.{i}"
            if x.is_cuda:
                from rvc_amd import _native
                x = _native.bias_relu_add_(F.conv_transpose2d(x, w[p + ".up.w"], None, stride=(2, 2), padding=(1, 1),
                                                              output_padding=(1, 1)).contiguous(), w[p + ".up.b"])
            else:
                x = F.relu(F.conv_transpose2d(x, w[p + ".up.w"], w[p + ".up.b"], stride=(2, 2), padding=(1, 1),
                                              output_padding=(1, 1)))
            x = torch.cat((x, skips[-1 - i]), dim=1)
            for m in range(4):
                x = self._block(x, f"{p}.conv2.{m}")
        if x.is_cuda and "cnn.w" in self.wp and x.shape[-1] in (4, 8, 16, 32, 64, 128):
            from rvc_amd import _native
            x = self._conv3(x.contiguous(), "cnn", w["cnn.w"].shape[0])
        else:
            x = F.conv2d(x, w["cnn.w"], w["cnn.b"], 1, 1)
        x = x.transpose(1, 2).flatten(-2)
        return F.linear(x, w["gru.wih"], w["gru.bih"]).view(x.shape[0], x.shape[1], 2, 768)

    @torch.no_grad()
    def gru_head(self, gi: torch.Tensor, n_frames: int) -> torch.Tensor:
        """The latency-bound half: the BiGRU recurrence (8 workgroups, ~1.7 us per step) + classifier -> salience."""
        from rvc_amd import _native
        w = self.w
        x = _native.bigru_forward(gi, w["gru.whhT"], w["gru.bhh"])
        return torch.sigmoid(F.linear(x, w["fc.w"], w["fc.b"]))[:, :n_frames]

    @torch.no_grad()
    def decode(self, hidden: torch.Tensor, thred=0.03) -> torch.Tensor:
        """salience [T,360] (device) -> f0 [T] float64 (device); RMVPE.py:459-512."""
        center = torch.argmax(hidden, dim=1)
        sal = F.pad(hidden, (4, 4))
        idx = center[:, None] + torch.arange(9, device=hidden.device)[None, :]  # (center+4) - 4 .. +4
        todo = torch.gather(sal, 1, idx)                     # float32, like the reference's todo_salience
        prod = todo.double() * self.cents_mapping[idx]       # float32 * float64 -> float64 (NumPy promotion)

        def pairwise9(t):  # NumPy's pairwise reduction order for 9 contiguous elements
            return (((t[:, 0] + t[:, 1]) + (t[:, 2] + t[:, 3])) + ((t[:, 4] + t[:, 5]) + (t[:, 6] + t[:, 7]))) + t[:, 8]

        cents = pairwise9(prod) / pairwise9(todo).double()   # weight_sum is a float32 sum in the reference
        cents = torch.where(hidden.max(dim=1).values <= thred, torch.zeros_like(cents), cents)
        f0 = 10 * torch.pow(2.0, cents / 1200)
        return torch.where(f0 == 10, torch.zeros_like(f0), f0)

    @torch.no_grad()
    def infer_from_audio_device(self, audio: torch.Tensor, thred=0.03) -> torch.Tensor:
        """audio [n] or [1,n] float32 on the device -> f0 float64 on the device."""
        from rvc_amd import _native
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        mel, n_frames = _native.logmel_rmvpe(audio.float(), pad_to=32)
        hidden = self.mel2hidden(mel, n_frames)
        return self.decode(hidden[0], thred)

    @torch.no_grad()
    def front_half_device(self, audio: torch.Tensor):
        """log-mel + U-Net + GRU input projections; returns (gi, n_frames) for `back_half_device`."""
        from rvc_amd import _native
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        mel, n_frames = _native.logmel_rmvpe(audio.float(), pad_to=32)
        return self.unet_features(mel), n_frames

    @torch.no_grad()
    def back_half_device(self, gi: torch.Tensor, n_frames: int, thred=0.03, taps=None) -> torch.Tensor:
        hidden = self.gru_head(gi, n_frames)[0]
        if taps is not None:      # tests: the salience behind the contour (device tensor [n_frames, 360])
            taps["salience"] = hidden
        return self.decode(hidden, thred)

    def infer_from_audio(self, audio, thred=0.03) -> np.ndarray:
        """Reference signature (RMVPE.py:472-485): NumPy audio in, NumPy float64 f0 out."""
        a = torch.from_numpy(np.ascontiguousarray(audio)).float().to(self.device)
        return self.infer_from_audio_device(a, thred).cpu().numpy()
