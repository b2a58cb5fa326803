#!/usr/bin/env python3
"""HuBERT's GEMM-shaped layers at the cfg-2 sizes: torch fp32 (hipBLASLt / MIOpen) vs gemmbf.hip (exact bf16x3 splits on the
bf16 matrix cores).  HIP events, median of 5 batches of 20 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch, torch.nn.functional as F
from rvc_amd import _native
dev = "cuda:0"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def timed(fn, batches=5, reps=20):
    for _ in range(3): fn()
    out = []
    for _ in range(batches):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]

T = 1599
for name, k, m, act in (("qkv", 768, 2304, "none"), ("out_proj", 768, 768, "none"), ("ff1+gelu", 768, 3072, "gelu"), ("ff2", 3072, 768, "none")):
    x = torch.randn(T, k, device=dev); w = torch.randn(m, k, device=dev) * k ** -0.5; b = torch.randn(m, device=dev)
    a = _native.gemm_bf16x3_pack_weight(w, dev)
    lib = timed(lambda: F.gelu(F.linear(x, w, b)) if act == "gelu" else F.linear(x, w, b))
    own = timed(lambda: _native.linear_bf16x3(x, a, b, m, act=act))
    xs = _native.split_rows_bf16x3(x)
    mode, parts = {"qkv": ("f32", 1), "out_proj": ("parts", 3), "ff1+gelu": ("gelu_planes", 1), "ff2": ("parts", 3)}[name]
    parts = int(os.environ.get("K_PARTS", parts)) if mode == "parts" else 1
    buf = _native.linear_bf16x3_presplit(xs, a, b if mode != "parts" else None, T, m, mode, parts)
    pre = timed(lambda: _native.linear_bf16x3_presplit(xs, a, b if mode != "parts" else None, T, m, mode, parts, out=buf))
    extra = ""
    if mode == "parts":
        res = torch.randn(T, m, device=dev); gam = torch.ones(m, device=dev)
        y, ys = _native.bias_residual_layernorm_bf16x3(buf, b, res, gam, gam, 1e-5)
        ln = timed(lambda: _native.bias_residual_layernorm_bf16x3(buf, b, res, gam, gam, 1e-5, y=y, ys=ys))
        lnlib = timed(lambda: F.layer_norm(res + x[:, :m] if k >= m else res, (m,), gam, gam, 1e-5))
        extra = f" + fused reduce/bias/residual/LayerNorm/split {ln*1e3:5.1f} us (torch add + layer_norm {lnlib*1e3:5.1f})"
    gf = 2.0 * T * k * m / 1e9
    print(f"linear {name:9s} {T} x {k:4d} -> {m:4d}: pre-split K12 ({mode}, {parts} K part(s)) {pre*1e3:7.1f} us ({gf/pre:6.1f} TF/s fp32-equivalent, "
          f"{6*gf/pre:7.1f} TF/s on the bf16 pipe) x{lib/pre:.2f} vs torch{extra}", flush=True)
    print(f"linear {name:9s} {T} x {k:4d} -> {m:4d}: torch fp32 {lib*1e3:7.1f} us ({gf/lib:6.1f} TF/s) | bf16x3 {own*1e3:7.1f} us ({gf/own:6.1f} TF/s fp32-equivalent, "
          f"{6*gf/own:7.1f} TF/s on the bf16 pipe) x{lib/own:.2f}", flush=True)
if os.environ.get('LINEAR_ONLY'): sys.exit(0)
L = 102399
for i, (k, s) in enumerate(((3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2))):
    x = torch.randn(1, 512, L, device=dev); w = torch.randn(512, 512, k, device=dev) * (512 * k) ** -0.5
    a = _native.gemm_bf16x3_pack_weight(w, dev)
    lib = timed(lambda: F.gelu(F.conv1d(x, w, None, stride=s)))
    own = timed(lambda: _native.conv1d_bf16x3(x, a, None, 512, k, stride=s, act="gelu"))
    Lo = (L - k) // s + 1
    gf = 2.0 * 512 * 512 * k * Lo / 1e9
    # K12 over time-major frames (rvc_conv1d_frames_bf16x3): planes in, GELU + planes out
    xs = _native.split_rows_bf16x3(x[0].t().contiguous())
    af = _native.gemm_bf16x3_pack_weight(w.permute(0, 2, 1).reshape(512, -1).contiguous(), dev)
    fr = timed(lambda: _native.conv1d_frames_bf16x3(xs, L, af, None, 512, k, s, "gelu_planes"))
    print(f"conv layer {i + 1} 512->512 k {k} s {s} L {L:6d}: torch fp32 conv + gelu {lib*1e3:7.1f} us ({gf/lib:6.1f} TF/s) | K11 channel-major {own*1e3:7.1f} us ({gf/own:6.1f} TF/s fp32-equivalent) x{lib/own:.2f}"
          f" | K12 over time-major frames {fr*1e3:7.1f} us ({gf/fr:6.1f} TF/s fp32-equivalent, {6*gf/fr:6.1f} on the bf16 pipe) x{lib/fr:.2f}", flush=True)
    L = Lo
# layer 0: conv + GroupNorm + GELU
wav = torch.randn(1, 1, 512000, device=dev) * 0.3
w0 = torch.randn(512, 1, 10, device=dev) * 0.4
gam = torch.randn(512, device=dev)
a0 = _native.gemm_bf16x3_pack_weight(w0, dev)
lib0 = timed(lambda: F.gelu(F.group_norm(F.conv1d(wav, w0, None, stride=5), 512, gam, gam, 1e-5)))
old0 = timed(lambda: _native.rownorm_gelu_(_native.conv1d_bf16x3(wav, a0, None, 512, 10, stride=5, act="none"), gam, gam, 1e-5))
new0 = timed(lambda: _native.hubert_conv0_frames_bf16x3(wav[0, 0], w0, gam, gam, 1e-5, stride=5))
print(f"layer 0 (conv 1->512 k 10 s 5 + GroupNorm + GELU) on 512000 samples: torch {lib0*1e3:7.1f} us | K11 + K8 (channel-major fp32) {old0*1e3:7.1f} us | "
      f"K13 (time-major planes, conv evaluated twice) {new0*1e3:7.1f} us", flush=True)
