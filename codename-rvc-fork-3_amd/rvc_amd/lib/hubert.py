"""HuBERT-base / ContentVec feature extractor on PyTorch-ROCm.

Replaces ``HubertModelWithFinalProj`` (rvc/lib/utils.py:31-34, a ``transformers.HubertModel``
subclass) at its one call site ``model(feats)["last_hidden_state"]`` (rvc/infer/pipeline.py:450)
and the v1-only ``model.final_proj`` (:451-453).  No ``transformers`` import: the network is
restated functionally over a state dict in transformers naming, weight-norm of
the positional conv folded once at load.  On the device the feature-extractor convs 0-3 run in
librvc_amd's K11 (gemmbf.hip), the attention in K7b (attention.hip) and the transformer layers'
projections + LayerNorms in K12 (linbf.hip), all as exact bf16x3 splits on the bf16 matrix cores;
everything stays fp32-valued (README.md:22).  CPU tensors (host-logic tests only) take plain torch ops.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from rvc_amd.lib.algorithm.weights import fold_weight_norm

CONV_STRIDES = (5, 2, 2, 2, 2, 2, 2)


class HubertModelWithFinalProj:
    def __init__(self, state_dict: Dict[str, torch.Tensor] | None = None, device="cpu", n_layers=12, n_heads=12):
        self.n_layers, self.n_heads = n_layers, n_heads
        self.device = torch.device(device)
        self.w: Dict[str, torch.Tensor] = {}
        # reference-RNG compatibility: transformers' encoder draws torch.rand([]) per layer for LayerDrop even
        # in eval mode, advancing the CPU generator the Synthesizer's noise comes from afterwards
        self.consume_layerdrop_rng = False
        self.native_min_rows = 900      # frames from which the transformer layers run on K12 (linbf.hip); tests set it to 1
        self.frame_convs = True         # feature extractor on time-major frames (K13 + K12) for single clips; tests compare both routes
        self.native_posconv = True      # positional conv in K14 for single clips; tests compare both routes
        self.frame_convs_min_samples = 400   # (the receptive field of one output frame)
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # --- the nn.Module surface the reference touches (infer.py:72-74) ---
    def load_state_dict(self, sd, strict: bool = True):
        w = fold_weight_norm(sd)
        self.w = {k: v.to(self.device) for k, v in w.items() if v.is_floating_point()}
        # feature-extractor convs 0-3 (1 -> 512 with 10 taps, then 512 -> 512, 3 taps, stride 2; 96 k .. 13 k columns for a 30 s clip) run in librvc_amd's K11
        # (gemmbf.hip: exact bf16x3 splits on the bf16 matrix cores, GELU in the epilogue): 1.44-1.73x MIOpen's NHWC igemm + its
        # transposes + the GELU pass (tools/bench_gemmbf.py), 0.55 ms per 30 s utterance; layers 4-6 (6 k columns and fewer) and the
        # fp32 projections stay on the libraries, which are faster at those sizes.  K11 launches one 8-wave block per CU and asks
        # for the CU's whole LDS: a workgroup issuing bf16 matrix instructions must not share a CU with the vocoder's fp32
        # Winograd kernel (profiles/r03_mfma_cohabitation.txt), and with two utterances in flight HuBERT overlaps the other
        # utterance's vocoder.
        self._conv_bf = {}
        if self.device.type == "cuda":
            from rvc_amd import _native
            for i in (0, 1, 2, 3):    # layer 0: 1 -> 512 channels, 10 taps, stride 5 (MIOpen: im2col + GEMM, 0.43 ms; here one k16 step)
                cw = self.w[f"feature_extractor.conv_layers.{i}.conv.weight"]
                if cw.shape[0] % 128 == 0 and (cw.shape[1] % 16 == 0 or (cw.shape[1] == 1 and cw.shape[2] <= 16)):
                    self._conv_bf[i] = _native.gemm_bf16x3_pack_weight(cw, self.device)
        self._pack_frame_convs()
        self._pack_posconv()
        self._lin = {}   # K12: fragment slabs of the four projections of every layer (device only)
        self._qkv = {}
        for i in range(self.n_layers):
            L = f"encoder.layers.{i}.attention"
            hd = self.w[L + ".q_proj.weight"].shape[0] // self.n_heads
            scale = hd ** -0.5
            # one [3D, D] projection; the 1/sqrt(d) query scale is applied to the product, like transformers
            self._qkv[i] = (torch.cat([self.w[L + ".q_proj.weight"], self.w[L + ".k_proj.weight"],
                                       self.w[L + ".v_proj.weight"]], 0).contiguous(),
                            torch.cat([self.w[L + ".q_proj.bias"], self.w[L + ".k_proj.bias"],
                                       self.w[L + ".v_proj.bias"]], 0).contiguous(), scale)
        self._pack_linears()
        return self

    def _pack_frame_convs(self):
        """Feature-extractor layers 1-6 as K12 GEMMs over time-major frames (rvc_conv1d_frames_bf16x3): the slab of
        W[o][k * C + c] = conv.weight[o][c][k]; layer 0 runs in K13 from its fp32 taps.  One clip at a time (batch 1)."""
        self._conv_fr = {}
        if self.device.type != "cuda":
            return
        from rvc_amd import _native
        n = len(CONV_STRIDES)
        ws = [self.w[f"feature_extractor.conv_layers.{i}.conv.weight"] for i in range(n)]
        c = ws[0].shape[0]
        ok = ws[0].shape[1] == 1 and ws[0].shape[2] == 10 and c % 128 == 0 and \
            all(w.shape[0] == c and w.shape[1] == c and (w.shape[2] * c) % 32 == 0 for w in ws[1:]) and \
            not any(f"feature_extractor.conv_layers.{i}.conv.bias" in self.w for i in range(n))
        if not ok:
            return
        for i in range(1, n):
            self._conv_fr[i] = _native.gemm_bf16x3_pack_weight(ws[i].permute(0, 2, 1).reshape(c, -1).contiguous(), self.device)

    def _pack_posconv(self):
        """The positional conv (Conv1d(D, D, 128, padding 64, groups 16)) as K14's fragment slab (posconv.hip): 48 or 64 channels per group."""
        self._posconv = None
        if self.device.type != "cuda":
            return
        from rvc_amd import _native
        pw = self.w["encoder.pos_conv_embed.conv.weight"]
        d, cg, taps = pw.shape
        if d % cg == 0 and cg in (48, 64) and taps <= 128:
            self._posconv = (_native.posconv_bf16x3_pack_weight(pw, d // cg, self.device), d // cg, taps)

    def _frame_convs_max_samples(self) -> int:
        """The longest clip the time-major route takes: K12 addresses an operand through one buffer descriptor (< 2 GiB), and the
        largest operand is layer 0's output, 3 planes x frames padded to 128 x C x 2 B (512 channels: ~218 s of audio).  Longer
        unsegmented clips fall through to the channel-major route below (Pipeline never sends more than x_max = 41 s at once)."""
        c = self.w["feature_extractor.conv_layers.0.conv.weight"].shape[0]
        frames = ((1 << 31) - (1 << 20)) // (6 * c) // 128 * 128 - 128
        return frames * CONV_STRIDES[0]

    def _features_native(self, wav):
        """conv_layers[0..6] of one clip -> [1, frames, C] fp32, time-major (what feature_projection consumes)."""
        from rvc_amd import _native as N
        w = self.w
        xs, n = N.hubert_conv0_frames_bf16x3(wav[0], w["feature_extractor.conv_layers.0.conv.weight"],
                                             w["feature_extractor.conv_layers.0.layer_norm.weight"],
                                             w["feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5, stride=CONV_STRIDES[0])
        last = len(CONV_STRIDES) - 1
        for i in range(1, last + 1):
            cw = w[f"feature_extractor.conv_layers.{i}.conv.weight"]
            xs, n = N.conv1d_frames_bf16x3(xs, n, self._conv_fr[i], None, cw.shape[0], cw.shape[2], CONV_STRIDES[i],
                                           "gelu_planes" if i < last else "gelu_f32")
        return xs[None]

    def _pack_linears(self):
        """K12 (linbf.hip) takes the nn.Linear weights as bf16x3 matrix-instruction fragments: q/k/v (fused), attention output,
        feed-forward 1 / 2 of every layer -- 42 MB per layer for HuBERT-base.  Shapes the kernel does not take (output features not
        a multiple of 128, input features not a multiple of 96 = 32 x 3 K parts) leave the layer on the library GEMMs."""
        self._lin = {}
        if self.device.type != "cuda":
            return
        from rvc_amd import _native
        for i in range(self.n_layers):
            L = f"encoder.layers.{i}"
            ws = [self._qkv[i][0], self.w[L + ".attention.out_proj.weight"], self.w[L + ".feed_forward.intermediate_dense.weight"],
                  self.w[L + ".feed_forward.output_dense.weight"]]
            d = ws[1].shape[0]
            if any(w.shape[0] % 128 or w.shape[1] % 96 for w in ws) or d not in (256, 768, 1024):
                self._lin = {}
                return
            self._lin[i] = [_native.gemm_bf16x3_pack_weight(w, self.device) for w in ws]

    def to(self, device):
        self.device = torch.device(device)
        self.w = {k: v.to(self.device) for k, v in self.w.items()}
        self._qkv = {i: (a.to(self.device), b.to(self.device), s) for i, (a, b, s) in self._qkv.items()}
        self._conv_bf = {i: a.to(self.device) for i, a in self._conv_bf.items()} if self.device.type == "cuda" else {}
        self._pack_frame_convs()
        self._pack_posconv()
        self._pack_linears()
        return self

    def float(self):
        return self

    def eval(self):
        return self

    def final_proj(self, x):
        return F.linear(x, self.w["final_proj.weight"], self.w["final_proj.bias"])

    @torch.no_grad()
    def __call__(self, wav: torch.Tensor):
        w = self.w
        if wav.is_cuda and wav.shape[0] == 1 and self._conv_fr and self.frame_convs and \
                self.frame_convs_min_samples <= wav.shape[1] <= self._frame_convs_max_samples():
            return self._encode(self._features_native(wav))
        x = wav[:, None, :]
        for i, s in enumerate(CONV_STRIDES):
            cw = w[f"feature_extractor.conv_layers.{i}.conv.weight"]
            if i in self._conv_bf and x.is_cuda and (i > 0 or x.shape[0] == 1):
                from rvc_amd import _native
                x = _native.conv1d_bf16x3(x, self._conv_bf[i], None, cw.shape[0], cw.shape[2], stride=s, act="gelu" if i > 0 else "none")
                if i > 0:
                    continue
            else:
                x = F.conv1d(x, cw, None, stride=s)
            if i == 0 and x.is_cuda and x.is_contiguous():
                from rvc_amd import _native      # GroupNorm(512, 512) + GELU in one launch (K8), float64 statistics
                x = _native.rownorm_gelu_(x, w["feature_extractor.conv_layers.0.layer_norm.weight"],
                                          w["feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5)
                continue
            if i == 0:
                x = F.group_norm(x, x.shape[1], w["feature_extractor.conv_layers.0.layer_norm.weight"],
                                 w["feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5)
            x = F.gelu(x)
        return self._encode(x.transpose(1, 2))

    def _encode(self, x):
        """feature_projection + encoder of transformers' HubertModel on [batch, frames, C] features."""
        w = self.w
        x = F.layer_norm(x, (x.shape[-1],), w["feature_projection.layer_norm.weight"],
                         w["feature_projection.layer_norm.bias"], 1e-5)
        x = F.linear(x, w["feature_projection.projection.weight"], w["feature_projection.projection.bias"])
        d = x.shape[-1]
        b, t, _ = x.shape
        if x.is_cuda and b == 1 and self._posconv is not None and self.native_posconv:
            # K14: gelu(grouped conv) over the time-major frames; the add + encoder.layer_norm in K12's fused pass
            from rvc_amd import _native as N
            a_pc, groups, taps = self._posconv
            x2 = x.reshape(t, d).contiguous()
            pos = N.posconv_gelu_bf16x3(x2, a_pc, w["encoder.pos_conv_embed.conv.bias"], groups, taps, taps // 2)
            if d in (256, 768, 1024):
                x, _ = N.bias_residual_layernorm_bf16x3(pos[None], None, x2, w["encoder.layer_norm.weight"], w["encoder.layer_norm.bias"],
                                                        1e-5, want_planes=False)
                x = x[None]
            else:
                x = F.layer_norm(x + pos[None], (d,), w["encoder.layer_norm.weight"], w["encoder.layer_norm.bias"], 1e-5)
        else:   # padding / groups from the weight's own shape, like the native route (HubertPositionalConvEmbedding: padding = taps // 2,
            pw = w["encoder.pos_conv_embed.conv.weight"]   # one frame dropped when taps is even -- HubertSamePadLayer)
            taps = pw.shape[2]
            pos = F.conv1d(x.transpose(1, 2), pw, w["encoder.pos_conv_embed.conv.bias"], padding=taps // 2, groups=d // pw.shape[1])
            if taps % 2 == 0:
                pos = pos[:, :, :-1]
            x = x + F.gelu(pos).transpose(1, 2)
            x = F.layer_norm(x, (d,), w["encoder.layer_norm.weight"], w["encoder.layer_norm.bias"], 1e-5)
        h = self.n_heads
        hd = d // h
        # K12 fills the chip from ~900 frames up (18 s of audio; 1599 frames: 1.23-1.37 x hipBLASLt per projection); below that its
        # 128-frame tiles leave most CUs idle and every launch costs its 24-chunk K loop whatever the frame count, so short clips
        # keep the library GEMMs (cfg 1's 599 frames: 13.8 against 12.9 ms per utterance)
        if x.is_cuda and self._lin and b * t >= self.native_min_rows:
            return {"last_hidden_state": self._encoder_native(x, b, t, d, h)}
        for i in range(self.n_layers):
            L = f"encoder.layers.{i}"
            if self.consume_layerdrop_rng:
                torch.rand([])
            wqkv, bqkv, scale = self._qkv[i]
            qkv = F.linear(x, wqkv, bqkv)
            if qkv.is_cuda:   # librvc_amd K7: reads the fused projection in place, writes the layout out_proj consumes
                from rvc_amd import _native
                a = _native.attention_qkv(qkv, h, scale)
            else:             # CPU tensors only occur in the host-logic tests
                qkv = qkv.view(b, t, 3, h, hd).permute(2, 0, 3, 1, 4)
                a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], scale=scale).transpose(1, 2).reshape(b, t, d)
            a = F.linear(a, w[L + ".attention.out_proj.weight"], w[L + ".attention.out_proj.bias"])
            x = F.layer_norm(x + a, (d,), w[L + ".layer_norm.weight"], w[L + ".layer_norm.bias"], 1e-5)
            f = F.gelu(F.linear(x, w[L + ".feed_forward.intermediate_dense.weight"],
                                w[L + ".feed_forward.intermediate_dense.bias"]))
            f = F.linear(f, w[L + ".feed_forward.output_dense.weight"], w[L + ".feed_forward.output_dense.bias"])
            x = F.layer_norm(x + f, (d,), w[L + ".final_layer_norm.weight"], w[L + ".final_layer_norm.bias"], 1e-5)
        return {"last_hidden_state": x}

    def _encoder_native(self, x, b, t, d, h):
        """The twelve post-LN transformer layers (transformers' HubertEncoderLayer) in librvc_amd: activations travel as fp32 rows
        AND as three bf16 planes (their exact split); per layer K12 q/k/v -> K7b attention -> split -> K12 out (K in 3 parts) ->
        fused sum + bias + residual + LayerNorm (+ split) -> K12 ff1 + GELU (-> planes) -> K12 ff2 (3 parts) -> fused LayerNorm."""
        from rvc_amd import _native as N
        w = self.w
        n = b * t
        x = x.reshape(n, d).contiguous()
        xs = N.split_rows_bf16x3(x)
        a_s = N.planes_empty(n, d, x.device)
        d_ff = w["encoder.layers.0.feed_forward.intermediate_dense.weight"].shape[0]
        fs = N.planes_empty(n, d_ff, x.device)
        parts = torch.empty((3, n, d), dtype=torch.float32, device=x.device)
        qkv = torch.empty((n, 3 * d), dtype=torch.float32, device=x.device)
        x1, x1s = torch.empty_like(x), N.planes_empty(n, d, x.device)
        for i in range(self.n_layers):
            L = f"encoder.layers.{i}"
            if self.consume_layerdrop_rng:
                torch.rand([])
            a_qkv, a_o, a_1, a_2 = self._lin[i]
            _, bqkv, scale = self._qkv[i]
            N.linear_bf16x3_presplit(xs, a_qkv, bqkv, n, 3 * d, "f32", out=qkv)
            a = N.attention_qkv(qkv.view(b, t, 3 * d), h, scale)
            N.split_rows_bf16x3(a.view(n, d), out=a_s)
            N.linear_bf16x3_presplit(a_s, a_o, None, n, d, "parts", 3, out=parts)
            N.bias_residual_layernorm_bf16x3(parts, w[L + ".attention.out_proj.bias"], x, w[L + ".layer_norm.weight"],
                                             w[L + ".layer_norm.bias"], 1e-5, y=x1, ys=x1s)
            N.linear_bf16x3_presplit(x1s, a_1, w[L + ".feed_forward.intermediate_dense.bias"], n, d_ff, "gelu_planes", out=fs)
            N.linear_bf16x3_presplit(fs, a_2, None, n, d, "parts", 3, out=parts)
            N.bias_residual_layernorm_bf16x3(parts, w[L + ".feed_forward.output_dense.bias"], x1, w[L + ".final_layer_norm.weight"],
                                             w[L + ".final_layer_norm.bias"], 1e-5, y=x, ys=xs)
        return x.view(b, t, d)
