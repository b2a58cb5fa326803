"""Parity of every HIP kernel against the oracle, through the C ABI (ctypes -> librvc_amd.so).

Run on the MI355X box: python -m pytest tests -m gpu.  Tolerances are stated at each check:
integer outputs (neighbour ids) are exact up to documented near-ties; waveforms are gated at the
north_star's 1e-3 RMS with the observed error asserted far below it.
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import REPO, load_golden, rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from rvc_amd import _native
    return _native


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def conv1d_f64(x, w, b=None, padding=0, dilation=1, stride=1):
    """F.conv1d in float64, evaluated on the device as one float64 matrix product per tap (rocBLAS dgemm): the same sums as
    F.conv1d(x.double(), w.double(), ...) on the host, which takes 20-45 s per full-length shape on the box's cores and was a
    third of this file's run time.  Returns a float64 tensor on the host."""
    dev0 = torch.device("cuda:0")
    xd, wd = x.to(dev0).double(), w.to(dev0).double()
    batch, c_in, length = xd.shape
    c_out, _, k = wd.shape
    xp = F.pad(xd, (padding, padding))
    l_out = (length + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    y = torch.zeros(batch, c_out, l_out, dtype=torch.float64, device=dev0)
    for t in range(k):
        seg = xp[:, :, t * dilation: t * dilation + (l_out - 1) * stride + 1: stride]
        y += torch.matmul(wd[:, :, t], seg)
    if b is not None:
        y += b.to(dev0).double()[None, :, None]
    return y.cpu()


# ---- K3 conv1d ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("c_in,c_out,k,dil,length,batch", [
    (32, 32, 3, 1, 1000, 1), (32, 32, 11, 5, 1537, 2), (64, 64, 7, 3, 700, 1), (128, 128, 11, 5, 513, 1),
    (256, 256, 3, 3, 300, 2), (192, 512, 7, 1, 100, 1), (64, 64, 3, 5, 31, 1), (128, 128, 7, 1, 4097, 1),
])
def test_conv1d_matches_torch_fp32(native, dev, c_in, c_out, k, dil, length, batch):
    g = torch.Generator().manual_seed(c_in * 1000 + k * 10 + dil)
    x = torch.randn(batch, c_in, length, generator=g)
    w = torch.randn(c_out, c_in, k, generator=g) / (c_in * k) ** 0.5
    b = torch.randn(c_out, generator=g)
    res = torch.randn(batch, c_out, length, generator=g)
    acc = torch.randn(batch, c_out, length, generator=g)
    ref = (F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), padding=(k - 1) // 2 * dil, dilation=dil)
           + res.double() + acc.double()) / 3
    wp = native.conv1d_pack_weight(w, dev)
    y = native.conv1d_forward(x.to(dev), wp, b.to(dev), c_out, k, dil, 0.1, res=res.to(dev), acc=acc.to(dev),
                              out_scale=1 / 3)
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5, err  # fp32 fma chain vs float64 reference, |ref| ~ 1
    # plain conv: no activation, no residual
    y2 = native.conv1d_forward(x.to(dev), wp, None, c_out, k, dil, 1.0)
    ref2 = conv1d_f64(x, w, None, padding=(k - 1) // 2 * dil, dilation=dil)
    assert (y2.cpu().double() - ref2).abs().max().item() <= 2e-5


@pytest.mark.parametrize("c_in,c_out,k,dil,length,batch", [
    (32, 32, 3, 1, 1000, 1), (32, 32, 11, 5, 1537, 2), (64, 64, 7, 3, 700, 1), (128, 128, 11, 5, 513, 1), (128, 128, 11, 1, 4096, 1),
    (256, 256, 3, 3, 300, 2), (64, 64, 3, 5, 31, 1), (128, 128, 7, 1, 4097, 1), (256, 256, 11, 3, 2051, 1), (32, 32, 7, 5, 9999, 1),
    (64, 128, 3, 1, 777, 1), (128, 64, 11, 1, 1234, 2),
])
def test_conv1d_winograd_matches_float64(native, dev, c_in, c_out, k, dil, length, batch):
    """The fast form of the ResBlock convs (wino.hip: grouped F(4,3), residuals.py:75-86 layers) against F.conv1d in
    float64, with the fused input activation, bias, residual, running sum and scale; lengths that are not multiples of a
    tile (4 d), every dilation, the 32-channel (1 x 4 waves) and 64+-channel (2 x 2) block shapes."""
    g = torch.Generator().manual_seed(c_in * 1000 + k * 10 + dil)
    x = torch.randn(batch, c_in, length, generator=g)
    w = torch.randn(c_out, c_in, k, generator=g) / (c_in * k) ** 0.5
    b = torch.randn(c_out, generator=g)
    res = torch.randn(batch, c_out, length, generator=g)
    acc = torch.randn(batch, c_out, length, generator=g)
    ref = (conv1d_f64(F.leaky_relu(x.double(), 0.1), w, b, padding=(k - 1) // 2 * dil, dilation=dil) + res.double()
           + acc.double()) / 3
    u = native.conv1d_wino_pack_weight(w, dev)
    got = native.conv1d_wino_forward(x.to(dev), u, b.to(dev), c_out, k, dil, 0.1, res=res.to(dev), acc=acc.to(dev), out_scale=1 / 3).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err <= 6e-5, err                         # |y| ~ 1, up to 2816 terms: a few fp32 ulps through the transforms
    plain = native.conv1d_wino_forward(x.to(dev), u, None, c_out, k, dil, 1.0).cpu()
    ref2 = conv1d_f64(x, w, None, padding=(k - 1) // 2 * dil, dilation=dil)
    assert (plain.double() - ref2).abs().max().item() <= 6e-5
    direct = native.conv1d_forward(x.to(dev), native.conv1d_pack_weight(w, dev), None, c_out, k, dil, 1.0).cpu()
    assert (plain - direct).abs().max().item() <= 6e-5


@pytest.mark.parametrize("c_in,c_out,k,dil,length,batch", [
    (64, 64, 7, 3, 700, 1), (128, 128, 11, 5, 513, 1), (128, 128, 11, 1, 4096, 1), (128, 128, 7, 1, 4097, 1),
    (256, 256, 11, 3, 2051, 1), (64, 128, 7, 1, 777, 1), (128, 64, 11, 1, 1234, 2), (64, 64, 11, 5, 9999, 1),
    (256, 256, 7, 5, 300, 2), (128, 128, 11, 3, 31, 1), (64, 64, 7, 1, 16384, 1),
    (128, 256, 11, 5, 1237, 2), (256, 256, 7, 1, 50, 1), (192, 128, 7, 3, 8191, 1),     # 128-row blocks: ragged, short, odd c_in
    (16, 128, 7, 1, 3000, 1), (48, 128, 11, 3, 5000, 1), (128, 128, 11, 1, 383760, 1),   # one chunk; an odd chunk count; the benchmarked stage-1 shape
    # three taps (F(4,3), six points, winobf2.hip only): every dilation, ragged lengths, batch, one chunk, odd chunk count, full length
    (128, 128, 3, 1, 4096, 1), (128, 128, 3, 3, 4097, 1), (256, 256, 3, 5, 2051, 2), (128, 128, 3, 5, 31, 1), (256, 128, 3, 1, 777, 1),
    (16, 128, 3, 1, 3000, 1), (48, 128, 3, 3, 5000, 1), (256, 256, 3, 1, 50, 2), (128, 128, 3, 2, 9999, 1), (128, 128, 3, 1, 383760, 1),
])
def test_conv1d_winograd_bf16x3_matches_float64(native, dev, c_in, c_out, k, dil, length, batch):
    """winobf.hip / winobf2.hip: the F(4,4) form of the 7- / 11-tap ResBlock convs and the F(4,3) form of the 3-tap ones at
    >= 128 channels (residuals.py:75-86) on the bf16 matrix cores, every
    fp32 operand split exactly into three bf16 and the six products of order <= 2^-16 accumulated in fp32.  Against
    F.conv1d in float64 with the fused activation, bias, residual, running sum and scale; every dilation, lengths that are
    not multiples of a tile, blocks with ragged tile counts, batch > 1.  The gate is the fp32 Winograd form's (6e-5 at
    |y| ~ 1), and the relative RMS error must not exceed 1.5x the fp32 Winograd form's on the same input."""
    g = torch.Generator().manual_seed(c_in * 1000 + k * 10 + dil)
    x = torch.randn(batch, c_in, length, generator=g)
    w = torch.randn(c_out, c_in, k, generator=g) / (c_in * k) ** 0.5
    b = torch.randn(c_out, generator=g)
    res = torch.randn(batch, c_out, length, generator=g)
    acc = torch.randn(batch, c_out, length, generator=g)
    ref = (conv1d_f64(F.leaky_relu(x.double(), 0.1), w, b, padding=(k - 1) // 2 * dil, dilation=dil) + res.double()
           + acc.double()) / 3
    u = native.conv1d_winobf_pack_weight(w, dev)
    got = native.conv1d_winobf_forward(x.to(dev), u, b.to(dev), c_out, k, dil, 0.1, res=res.to(dev), acc=acc.to(dev), out_scale=1 / 3).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err <= 6e-5, err
    ref2 = conv1d_f64(x, w, None, padding=(k - 1) // 2 * dil, dilation=dil)
    plain = native.conv1d_winobf_forward(x.to(dev), u, None, c_out, k, dil, 1.0).cpu()
    assert (plain.double() - ref2).abs().max().item() <= 6e-5
    fp32w = native.conv1d_wino_forward(x.to(dev), native.conv1d_wino_pack_weight(w, dev), None, c_out, k, dil, 1.0).cpu()
    direct = native.conv1d_forward(x.to(dev), native.conv1d_pack_weight(w, dev), None, c_out, k, dil, 1.0).cpu()
    rel = lambda t: ((t.double() - ref2).pow(2).mean().sqrt() / ref2.pow(2).mean().sqrt()).item()
    r_bf, r_w, r_d = rel(plain), rel(fp32w), rel(direct)
    print(f"C {c_in}->{c_out} k {k} d {dil} L {length}: relative RMS error vs float64: bf16x3 Winograd {r_bf:.2e}, fp32 Winograd {r_w:.2e}, "
          f"fp32 direct {r_d:.2e}")
    assert r_bf <= 1.5 * r_w


@pytest.mark.parametrize("c,k,dil,length,batch", [
    (32, 3, 1, 4096, 1), (32, 3, 3, 4097, 1), (32, 3, 5, 1000, 2), (32, 7, 1, 5003, 1), (32, 7, 3, 749, 2), (32, 7, 5, 16384, 1),
    (32, 11, 1, 2051, 1), (32, 11, 3, 9999, 1), (32, 11, 5, 513, 2), (32, 11, 5, 31, 1), (32, 3, 1, 5, 1), (32, 7, 5, 244, 1),
    (64, 3, 1, 4096, 1), (64, 3, 5, 777, 2), (64, 7, 1, 5003, 1), (64, 7, 3, 1234, 1), (64, 11, 1, 2051, 1), (64, 11, 5, 9999, 2),
    (64, 11, 3, 117, 1), (64, 7, 5, 50, 1),
    (128, 3, 1, 4096, 1), (128, 3, 3, 1237, 2), (128, 3, 5, 61, 1), (128, 7, 1, 2051, 1), (128, 7, 5, 777, 1),
    (32, 7, 3, 1535040, 1), (32, 11, 5, 1535040, 1), (64, 7, 1, 767520, 1), (64, 3, 5, 767520, 1), (128, 3, 3, 383760, 1),   # the benchmarked stage shapes
])
def test_resblock_pair_bf16x3_matches_float64(native, dev, c, k, dil, length, batch):
    """resblock_bf.hip (K3f): one (dilated conv -> conv) pair of ResBlock.forward (residuals.py:75-86) in one launch, direct form on
    the bf16 matrix cores with every fp32 operand -- taps, activations and the intermediate -- split exactly into three bf16.
    Against the same pair in float64 (F.conv1d), with biases, the residual, the running sum and the final scale; every
    dilation, lengths shorter than one tile / not a multiple of anything, several tiles per block (persistent loop), batch > 1.
    Gate: 6e-5 at |y| ~ 1 (the single convs' gate); the relative RMS error must stay at the fp32 direct form's level."""
    g = torch.Generator().manual_seed(c * 1000 + k * 10 + dil)
    x = torch.randn(batch, c, length, generator=g)
    w1 = torch.randn(c, c, k, generator=g) / (c * k) ** 0.5
    w2 = torch.randn(c, c, k, generator=g) / (c * k) ** 0.5
    b1, b2 = torch.randn(c, generator=g), torch.randn(c, generator=g)
    acc = torch.randn(batch, c, length, generator=g)

    def pair64(xx, bb1, bb2):
        t = conv1d_f64(F.leaky_relu(xx.double(), 0.1), w1, bb1, padding=(k - 1) // 2 * dil, dilation=dil)
        return conv1d_f64(F.leaky_relu(t, 0.1), w2, bb2, padding=(k - 1) // 2) + xx.double()

    u = native.resblock_bf16x3_pack_weight(w1, w2, dev)
    xd = x.to(dev)
    ref = (pair64(x, b1.double(), b2.double()) + acc.double()) / 3
    got = native.resblock_bf16x3_forward(xd, u, b1.to(dev), b2.to(dev), k, dil, 0.1, acc=acc.to(dev), out_scale=1 / 3).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err <= 6e-5, err
    ref2 = pair64(x, None, None)
    plain = native.resblock_bf16x3_forward(xd, u, None, None, k, dil, 0.1).cpu()
    assert (plain.double() - ref2).abs().max().item() <= 6e-5
    # the same pair as two launches of the fp32 direct-form kernel (conv.hip): the error level this kernel has to hold
    pw1, pw2 = native.conv1d_pack_weight(w1, dev), native.conv1d_pack_weight(w2, dev)
    t32 = native.conv1d_forward(xd, pw1, None, c, k, dil, 0.1)
    d32 = native.conv1d_forward(t32, pw2, None, c, k, 1, 0.1, res=xd).cpu()
    rel = lambda t: ((t.double() - ref2).pow(2).mean().sqrt() / ref2.pow(2).mean().sqrt()).item()
    r_bf, r_d = rel(plain), rel(d32)
    print(f"C {c} k {k} d {dil} L {length} B {batch}: relative RMS error vs float64: fused bf16x3 pair {r_bf:.2e}, two fp32 direct convs {r_d:.2e}")
    assert r_bf <= 1.5 * r_d + 1e-8
    again = native.resblock_bf16x3_forward(xd, u, None, None, k, dil, 0.1).cpu()
    assert torch.equal(again, plain)                      # bit-reproducible


@pytest.mark.parametrize("c,k,dil,length,batch", [
    (32, 3, 1, 4096, 1), (32, 3, 5, 1000, 2), (32, 7, 3, 749, 2), (32, 7, 5, 16384, 1), (32, 11, 1, 2051, 1), (32, 11, 5, 513, 2),
    (32, 11, 5, 31, 1), (32, 3, 1, 5, 1), (64, 3, 5, 777, 2), (64, 7, 1, 5003, 1), (64, 11, 1, 2051, 1), (64, 11, 5, 9999, 2),
    (64, 11, 3, 117, 1), (128, 3, 3, 1237, 2), (128, 7, 1, 2051, 1), (128, 7, 5, 777, 1), (128, 7, 3, 50, 1),
    (32, 11, 5, 1535040, 1), (64, 11, 3, 767520, 1), (128, 7, 5, 383760, 1),   # the cfg-4 stage shapes
])
def test_resblock_pair_bf16_taps_matches_float64(native, dev, c, k, dil, length, batch):
    """K3f with ONE-TERM taps (rvc_resblock_bf16w_*): BASELINE cfg 4's "alt ResBlock kernel path" -- the MRF layer
    (hifigan_mrf.py:13-83 = residuals.py:75-86) with bf16-stored weights.  A bf16-valued tap is the first term of its own split, so
    three products per multiply-add (w x_0 + w x_1 + w x_2) are exact to the same 2^-23 as the six of the fp32-tap form.  Reference:
    the pair in float64 on the bf16-ROUNDED taps (the cfg-4 oracle runs fp32 math on the same rounded weights); the result
    must also be what the three-term kernel gives on fragments of those rounded taps (its second and third fragments are zero) to
    fp32 summation-order noise, and be bit-reproducible."""
    g = torch.Generator().manual_seed(c * 1000 + k * 10 + dil + 7)
    x = torch.randn(batch, c, length, generator=g)
    w1 = (torch.randn(c, c, k, generator=g) / (c * k) ** 0.5)
    w2 = (torch.randn(c, c, k, generator=g) / (c * k) ** 0.5)
    w1r, w2r = w1.bfloat16().float(), w2.bfloat16().float()
    b1, b2 = torch.randn(c, generator=g), torch.randn(c, generator=g)
    acc = torch.randn(batch, c, length, generator=g)

    def pair64(xx, bb1, bb2):
        t = conv1d_f64(F.leaky_relu(xx.double(), 0.1), w1r, bb1, padding=(k - 1) // 2 * dil, dilation=dil)
        return conv1d_f64(F.leaky_relu(t, 0.1), w2r, bb2, padding=(k - 1) // 2) + xx.double()

    u1 = native.resblock_bf16x3_pack_weight(w1, w2, dev, bf16_taps=True)       # unrounded in: the pack rounds (RNE)
    assert u1.numel() * 3 == native.resblock_bf16x3_pack_weight(w1r, w2r, dev).numel()
    xd = x.to(dev)
    ref = (pair64(x, b1.double(), b2.double()) + acc.double()) / 3
    got = native.resblock_bf16x3_forward(xd, u1, b1.to(dev), b2.to(dev), k, dil, 0.1, acc=acc.to(dev), out_scale=1 / 3, bf16_taps=True).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err <= 6e-5, err
    ref2 = pair64(x, None, None)
    plain = native.resblock_bf16x3_forward(xd, u1, None, None, k, dil, 0.1, bf16_taps=True).cpu()
    assert (plain.double() - ref2).abs().max().item() <= 6e-5
    u3 = native.resblock_bf16x3_pack_weight(w1r, w2r, dev)
    three = native.resblock_bf16x3_forward(xd, u3, None, None, k, dil, 0.1).cpu()
    rel = lambda t: ((t.double() - ref2).pow(2).mean().sqrt() / ref2.pow(2).mean().sqrt()).item()
    r1, r3 = rel(plain), rel(three)
    print(f"C {c} k {k} d {dil} L {length} B {batch}: relative RMS error vs float64 on the rounded taps: one-term {r1:.2e}, three-term {r3:.2e}")
    assert r1 <= 1.5 * r3 + 1e-8
    again = native.resblock_bf16x3_forward(xd, u1, None, None, k, dil, 0.1, bf16_taps=True).cpu()
    assert torch.equal(again, plain)                      # bit-reproducible


@pytest.mark.parametrize("c,k,dil,length,batch", [
    (128, 11, 1, 4096, 1), (128, 11, 3, 4097, 1), (128, 11, 5, 777, 2), (128, 3, 1, 1000, 1), (128, 7, 5, 31, 1), (128, 11, 5, 63, 1),
    (256, 3, 1, 2051, 1), (256, 7, 3, 1237, 2), (256, 11, 5, 5003, 1), (256, 11, 1, 64, 1), (256, 7, 1, 65, 1), (256, 11, 3, 9999, 1),
    (128, 11, 5, 383760, 1), (256, 11, 3, 38376, 1), (256, 7, 5, 38376, 1),   # the cfg-4 stage shapes
])
def test_conv1d_bf16_taps_direct_matches_float64(native, dev, c, k, dil, length, batch):
    """K3d (convbf1.hip, rvc_conv1d_bf16w_*): one square conv of the MRF layer (hifigan_mrf.py:13-83 = residuals.py:75-86) with bf16-stored
    taps in direct form -- one-term taps x exact bf16x3 activations -- at the channel counts the fused pair cannot hold (C = 256;
    C = 128 with 11 taps).  Against F.conv1d in float64 on the bf16-ROUNDED taps with the fused activation, bias, residual, running sum
    and scale; every dilation, lengths shorter than a tile / not a multiple of anything (the element-wise store path), several tiles
    per block (persistent loop), batch > 1, res aliasing y (how the decoder calls it).  Gate: the bf16x3 Winograd form's (6e-5 at
    |y| ~ 1); the relative RMS error must not exceed 1.5 x the Winograd form's on fragments of the same rounded taps; bit-reproducible."""
    g = torch.Generator().manual_seed(c * 1000 + k * 10 + dil + 3)
    x = torch.randn(batch, c, length, generator=g)
    w = torch.randn(c, c, k, generator=g) / (c * k) ** 0.5
    wr = w.bfloat16().float()
    b = torch.randn(c, generator=g)
    res = torch.randn(batch, c, length, generator=g)
    acc = torch.randn(batch, c, length, generator=g)
    ref = (conv1d_f64(F.leaky_relu(x.double(), 0.1), wr, b, padding=(k - 1) // 2 * dil, dilation=dil) + res.double() + acc.double()) / 3
    u = native.conv1d_bf16w_pack_weight(w, dev)            # unrounded in: the pack rounds (RNE)
    xd = x.to(dev)
    got = native.conv1d_bf16w_forward(xd, u, b.to(dev), k, dil, 0.1, res=res.to(dev), acc=acc.to(dev), out_scale=1 / 3).cpu()
    err = (got.double() - ref).abs().max().item()
    assert err <= 6e-5, err
    ref2 = conv1d_f64(x, wr, None, padding=(k - 1) // 2 * dil, dilation=dil)
    plain = native.conv1d_bf16w_forward(xd, u, None, k, dil, 1.0).cpu()
    assert (plain.double() - ref2).abs().max().item() <= 6e-5
    wino = native.conv1d_winobf_forward(xd, native.conv1d_winobf_pack_weight(wr, dev), None, c, k, dil, 1.0).cpu()
    rel = lambda t: ((t.double() - ref2).pow(2).mean().sqrt() / ref2.pow(2).mean().sqrt()).item()
    r1, rw = rel(plain), rel(wino)
    print(f"C {c} k {k} d {dil} L {length} B {batch}: relative RMS error vs float64 on the rounded taps: direct one-term {r1:.2e}, bf16x3 Winograd {rw:.2e}")
    assert r1 <= 1.5 * rw + 1e-8
    # res aliasing y (conv2 of the second and third dilation in the decoder's schedule), twice: bit-reproducible
    y1 = res.to(dev).clone()
    native.conv1d_bf16w_forward(xd, u, b.to(dev), k, dil, 0.1, res=y1, out=y1)
    y2 = res.to(dev).clone()
    native.conv1d_bf16w_forward(xd, u, b.to(dev), k, dil, 0.1, res=y2, out=y2)
    assert torch.equal(y1, y2)
    ref3 = conv1d_f64(F.leaky_relu(x.double(), 0.1), wr, b, padding=(k - 1) // 2 * dil, dilation=dil) + res.double()
    assert (y1.cpu().double() - ref3).abs().max().item() <= 6e-5


@pytest.mark.parametrize("c_in,c_out,rate,ksize,nc_k,nc_stride,length,batch", [
    (512, 256, 12, 24, 0, 1, 200, 1), (512, 256, 12, 24, 0, 1, 3198, 1),            # stage 0 of the 48 k vocoder (its noise conv runs separately)
    (256, 128, 10, 20, 8, 4, 777, 2), (256, 128, 10, 20, 8, 4, 38376, 1),           # stage 1: 44 folded noise rows
    (128, 64, 2, 4, 4, 2, 5003, 1), (64, 32, 2, 4, 1, 1, 4097, 2), (64, 32, 2, 4, 1, 1, 63, 1),   # stages 2 / 3
    (512, 256, 10, 16, 0, 1, 333, 1), (256, 128, 8, 16, 8, 4, 1000, 1),             # the 40 k / 32 k vocoders' first stages (ksize < 2 rate)
    (128, 64, 2, 4, 0, 1, 1, 1), (64, 32, 2, 4, 1, 1, 767520, 1),                   # one input position; the benchmarked last stage
])
def test_upsample_bf16x3_matches_float64(native, dev, c_in, c_out, rate, ksize, nc_k, nc_stride, length, batch):
    """K3u (upsbf.hip, rvc_upsample_bf16x3_*): one upsampling step of the NSF / MRF vocoder -- leaky ReLU, ConvTranspose1d in polyphase form,
    the noise conv of har_source folded in as extra GEMM rows (more than 4 taps; the bias then rides a row of ones) or evaluated in the
    epilogue (hifigan_nsf.py:184-193) -- on the bf16 matrix cores with exact bf16x3
    operands, against F.conv_transpose1d + F.conv1d in float64.  Every stage shape of the 48 k / 40 k / 32 k vocoders, ragged lengths,
    a single input position, batch > 1.  Gate: 2e-5 of the largest value; relative RMS at the fp32 kernel's level; bit-reproducible."""
    g = torch.Generator().manual_seed(c_in + rate * 7 + length)
    pad = (ksize - rate) // 2
    x = torch.randn(batch, c_in, length, generator=g)
    w = torch.randn(c_in, c_out, ksize, generator=g) / (2 * c_in) ** 0.5
    b = torch.randn(c_out, generator=g)
    l_out = (length - 1) * rate - 2 * pad + ksize
    ref = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=rate, padding=pad)
    har, nw, nc_pad = None, None, 0
    if nc_k:
        nc_pad = 0 if nc_stride == 1 else (nc_k - nc_stride) // 2
        lh = l_out * nc_stride
        har = torch.randn(batch, lh, generator=g)
        nw = torch.randn(c_out, 1, nc_k, generator=g) * 0.3
        ref = ref + F.conv1d(har.double()[:, None], nw.double(), None, stride=nc_stride, padding=nc_pad)[:, :, :l_out]
    assert ref.shape == (batch, c_out, l_out)
    packed = native.upsample_bf16x3_pack_weight(w, nw, b, rate, nc_stride, dev)
    args = (x.to(dev), har.to(dev) if har is not None else None, packed, c_out, rate, ksize, pad, nc_stride, nc_pad, 0.1)
    got = native.upsample_bf16x3_forward(*args)
    assert got.shape == ref.shape
    err = (got.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    rel = ((got.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    lib = F.conv_transpose1d(F.leaky_relu(x.to(dev), 0.1), w.to(dev), b.to(dev), stride=rate, padding=pad)
    if nc_k:
        lib = lib + F.conv1d(har.to(dev)[:, None], nw.to(dev), None, stride=nc_stride, padding=nc_pad)[:, :, :l_out]
    rel_lib = ((lib.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"upsample {c_in}->{c_out} x{rate} k{ksize} noise {nc_k}/{nc_stride} L {length} B {batch}: relative RMS error vs float64: bf16x3 {rel:.2e}, torch fp32 {rel_lib:.2e}")
    assert rel <= max(1.5 * rel_lib, 3e-7)
    assert torch.equal(native.upsample_bf16x3_forward(*args), got)


@pytest.mark.parametrize("c,form", [(32, "three"), (64, "three"), (128, "three"), (32, "one"), (64, "one"), (128, "one"), (128, "direct"), (256, "direct")])
def test_resblock_pair_fresh_buffers_right_after_load(c, form):
    """ADVICE round 5 (resblock_bf.hip:290): K3f is the default path of the narrow stages and carries a race that was removed by
    reordering the stager prologue without being explained (profiles/r05_rbf_notes.txt item 5: wrong second tiles on the FIRST
    launches into FRESH output buffers; K3d, convbf1.hip, shares the stager design and is held to the same test).  tools/stress_rbf.py
    reproduces that regime in a child process whose first GPU work it is:
    every benchmarked (taps, dilation) of this channel count, 50 launches each into 50 never-written buffers (25 untouched, 25
    NaN-poisoned), all bit-equal to the first and the first equal to the fp32 direct-form pair."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "stress_rbf.py"), str(c), form, "50"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_short_clip_vocoders_and_pipeline_bit_stable_under_allocator_churn():
    """tools/stress_pipeline.py: the 40 k / 48 k / 32 k NSF and the 48 k MRF vocoder on 2-3 s inputs, 150 forwards each with the caching
    allocator's blocks moved around in between (fresh buffers every few iterations), every output BIT-EQUAL to the first; then the whole
    pipeline on the 2 s / 40 kHz case that failed one full-suite run of round 6 at 1.1e-2 (reruns: 2.5e-6), 12 runs, each within 1e-4 of
    the first (observed: exactly equal).  A sporadic kernel-level corruption on short clips -- few tiles per block, the regime the
    fresh-buffer test above does not cover -- would show here."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "stress_pipeline.py"), "150", "12"], capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_resblock_pair_runtime_switch(native, dev):
    """rvc_resblock_bf16x3_set_enabled(0): the operator's fall-back without a rebuild -- a decoder finalized while the switch is off runs
    its narrow stages on the unfused kernels (launch_resblock_layer / winobf / wino) and must give the waveform of the default
    handle to fp32 accumulation noise; handles finalized earlier are not affected; the switch is restored."""
    from rvc_amd.lib import synthetic as S
    from rvc_amd.lib.algorithm.weights import fold_weight_norm
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
    folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
    T = 300
    g = torch.Generator(device=dev).manual_seed(5)
    args = (torch.randn(1, 192, T, device=dev, generator=g), torch.full((1, T), 220.0, device=dev), torch.randn(1, 256, device=dev, generator=g))
    kw = dict(src_randn=torch.randn(1, T * 480, 1, device=dev, generator=g), src_rand=torch.zeros(1, 1, device=dev))
    on = native.Decoder("HiFi-GAN", 48000, folded)
    ref = on.forward(*args, **kw).clone()
    native.resblock_bf16x3_set_enabled(False)
    try:
        off = native.Decoder("HiFi-GAN", 48000, folded)
        out_off = off.forward(*args, **kw).clone()
        assert torch.equal(on.forward(*args, **kw), ref)          # the earlier handle keeps its kernels
    finally:
        native.resblock_bf16x3_set_enabled(True)
    again = native.Decoder("HiFi-GAN", 48000, folded).forward(*args, **kw)
    assert torch.equal(again, ref)
    e = rms((out_off - ref).cpu().numpy())
    print(f"K3f off vs on: waveform rms difference {e:.2e} (signal rms {rms(ref.cpu().numpy()):.3f})")
    assert 0.0 < e <= 2e-6, e                                      # (exactly 0 would mean the switch changed nothing)


@pytest.mark.parametrize("n_rows,k,m,mode,k_parts", [
    (1599, 768, 2304, "f32", 1), (1599, 768, 768, "parts", 3), (1599, 768, 3072, "gelu_planes", 1),
    (1599, 3072, 768, "parts", 3), (149, 768, 768, "parts", 6), (1, 768, 2304, "f32", 1), (300, 256, 256, "parts", 2),
    (4797, 768, 3072, "gelu_planes", 1), (257, 3072, 768, "f32", 1), (130, 64, 256, "parts", 1), (128, 32, 128, "f32", 1),
])
def test_linear_bf16x3_presplit_matches_float64(native, dev, n_rows, k, m, mode, k_parts):
    """linbf.hip (K12): HuBERT's transformer projections (attention q/k/v + out, feed-forward; `transformers` HubertEncoderLayer behind
    pipeline.py:450) with both operands pre-split into three bf16: plain, GELU -> planes, and split-K partial sums meeting in the
    fused bias + residual + LayerNorm pass.  Against float64; gate and error level as the K11 test (2e-5 of the largest value,
    relative RMS at torch's fp32 level).  Ragged row counts (1599 = 6.2 tiles of 256), a single row, three stacked utterances."""
    g = torch.Generator().manual_seed(n_rows + k + m)
    x = torch.randn(n_rows, k, generator=g)
    w = torch.randn(m, k, generator=g) * k ** -0.5
    b = torch.randn(m, generator=g)
    xd = x.to(dev)
    xs = native.split_rows_bf16x3(xd)
    back = xs[:, :n_rows].float().sum(0)                       # bf16 planes: exact in fp32, and their fp32 sum is exact (24 bits)
    assert torch.equal(back, xd), "the three planes do not sum to the fp32 input"
    a = native.gemm_bf16x3_pack_weight(w, dev)
    ref = x.double() @ w.double().t() + b.double()
    lib = F.linear(xd, w.to(dev), b.to(dev)).cpu()
    rel = lambda t, r: ((t.double() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
    if mode == "f32":
        got = native.linear_bf16x3_presplit(xs, a, b.to(dev), n_rows, m, "f32", 1).cpu()
        assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
        assert rel(got, ref) <= max(1.5 * rel(lib, ref), 4e-7), (rel(got, ref), rel(lib, ref))
    elif mode == "gelu_planes":
        ys = native.linear_bf16x3_presplit(xs, a, b.to(dev), n_rows, m, "gelu_planes", 1)
        got = ys[:, :n_rows].float().sum(0).cpu()
        refg = F.gelu(ref)
        assert (got.double() - refg).abs().max().item() <= 2e-5 * refg.abs().max().item()
        assert rel(got, refg) <= max(1.5 * rel(F.gelu(lib), refg), 3e-7)
    else:
        parts = native.linear_bf16x3_presplit(xs, a, None, n_rows, m, "parts", k_parts)
        res = torch.randn(n_rows, m, generator=g)
        gamma, beta = torch.randn(m, generator=g), torch.randn(m, generator=g)
        y, ys = native.bias_residual_layernorm_bf16x3(parts, b.to(dev), res.to(dev), gamma.to(dev), beta.to(dev), 1e-5)
        refl = F.layer_norm(ref + res.double(), (m,), gamma.double(), beta.double(), 1e-5)
        libl = F.layer_norm(lib.to(dev) + res.to(dev), (m,), gamma.to(dev), beta.to(dev), 1e-5).cpu()
        assert (y.cpu().double() - refl).abs().max().item() <= 2e-5 * refl.abs().max().item()
        assert rel(y.cpu(), refl) <= max(1.5 * rel(libl, refl), 3e-7)
        assert torch.equal(ys[:, :n_rows].float().sum(0), y)    # the planes are the fp32 output, split exactly
        plain = parts.sum(0).cpu() + b
        assert (plain.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("frames_in,c,taps,stride,m,mode", [
    (95999, 512, 3, 2, 512, "gelu_planes"), (2999, 512, 2, 2, 512, "gelu_f32"), (11999, 512, 3, 2, 512, "gelu_planes"),
    (131, 64, 3, 2, 128, "f32"), (3, 512, 3, 2, 512, "gelu_f32"), (257, 128, 2, 1, 256, "gelu_planes"), (1000, 32, 5, 3, 128, "f32"),
])
def test_conv1d_frames_bf16x3_matches_float64(native, dev, frames_in, c, taps, stride, m, mode):
    """K12 as a strided conv over TIME-MAJOR frames (rvc_conv1d_frames_bf16x3: HuBERT's feature-extractor layers 1-6,
    `transformers` HubertNoLayerNormConvLayer behind pipeline.py:450): the window of output frame t is the contiguous run of
    taps x channels values starting at frame t * stride.  Against float64 F.conv1d on the channel-major tensor; the error level of
    torch's fp32 conv on the same operands.  The 30 s clip's first and sixth layers, a clip of one output frame, odd shapes."""
    g = torch.Generator().manual_seed(frames_in + c + taps)
    x = torch.randn(frames_in, c, generator=g)                       # [frame][channel]
    w = torch.randn(m, c, taps, generator=g) * (c * taps) ** -0.5
    b = torch.randn(m, generator=g)
    xd = x.to(dev)
    xs = native.split_rows_bf16x3(xd)
    a = native.gemm_bf16x3_pack_weight(w.permute(0, 2, 1).reshape(m, -1).contiguous(), dev)
    ref = conv1d_f64(x.t()[None], w, b, stride=stride)[0].t()                                # [frames_out][m], float64 on the host
    lib = F.conv1d(xd.t()[None], w.to(dev), b.to(dev), stride=stride)[0].t().cpu()
    if mode != "f32":
        ref, lib = F.gelu(ref), F.gelu(lib)
    y, n_out = native.conv1d_frames_bf16x3(xs, frames_in, a, b.to(dev), m, taps, stride, mode)
    assert n_out == ref.shape[0] == (frames_in - taps) // stride + 1
    got = (y[:, :n_out].float().sum(0) if mode == "gelu_planes" else y).cpu()
    assert got.shape == ref.shape
    rel = lambda t, r: ((t.double() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
    err = (got.double() - ref).abs().max().item()
    print(f"conv over frames [{frames_in} x {c}] k{taps} s{stride} -> {m} ({mode}): max abs err {err:.2e}, rel rms {rel(got, ref):.2e} (torch fp32 {rel(lib, ref):.2e})")
    assert err <= 2e-5 * ref.abs().max().item()
    assert rel(got, ref) <= max(1.5 * rel(lib, ref), 6e-7)      # (one output frame = 512 values: the ratio is noisy there)
    y2, _ = native.conv1d_frames_bf16x3(xs, frames_in, a, b.to(dev), m, taps, stride, mode)
    assert torch.equal(y2[:, :n_out] if mode == "gelu_planes" else y2, y[:, :n_out] if mode == "gelu_planes" else y)


@pytest.mark.parametrize("frames,d,groups,taps", [(1499, 768, 16, 128), (49, 768, 16, 128), (1, 768, 16, 128), (300, 1024, 16, 128), (257, 96, 2, 5)])
def test_posconv_gelu_bf16x3_matches_float64(native, dev, frames, d, groups, taps):
    """K14 (posconv.hip): HuBERT's positional conv embedding -- Conv1d(D, D, 128, padding 64, groups 16) -> drop the last frame -> GELU
    (transformers' HubertPositionalConvEmbedding behind pipeline.py:450) over time-major frames.  Against float64 F.conv1d; the
    error level of torch's fp32 conv.  The 30 s clip, the golden clip's 49 frames, one frame, HuBERT-large's 64 channels per group,
    a 5-tap toy with two groups."""
    g = torch.Generator().manual_seed(frames + d)
    cg = d // groups
    x = torch.randn(frames, d, generator=g)
    w = torch.randn(d, cg, taps, generator=g) * (cg * taps) ** -0.5
    b = torch.randn(d, generator=g)
    pad = taps // 2
    ref = F.gelu(F.conv1d(x.t().double()[None], w.double(), b.double(), padding=pad, groups=groups)[0, :, :frames]).t()
    xd = x.to(dev)
    lib = F.gelu(F.conv1d(xd.t()[None], w.to(dev), b.to(dev), padding=pad, groups=groups)[0, :, :frames]).t().cpu()
    a = native.posconv_bf16x3_pack_weight(w, groups, dev)
    got = native.posconv_gelu_bf16x3(xd, a, b.to(dev), groups, taps, pad)
    again = native.posconv_gelu_bf16x3(xd, a, b.to(dev), groups, taps, pad)
    assert torch.equal(got, again)
    got = got.cpu()
    rel = lambda t, r: ((t.double() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
    err = (got.double() - ref).abs().max().item()
    print(f"positional conv [{frames} x {d}] groups {groups} taps {taps}: max abs err {err:.2e}, rel rms {rel(got, ref):.2e} (torch fp32 {rel(lib, ref):.2e})")
    assert err <= 2e-5 * max(1.0, ref.abs().max().item())
    # one fp32 accumulator chain over K = taps x 48 = 6144 products (sqrt(K) 2^-24 = 4.7e-6 is the random-walk level; the library's
    # GEMM splits K and lands at 5e-7)
    assert rel(got, ref) <= max(2.0 * rel(lib, ref), 1.2e-6)


@pytest.mark.parametrize("n_samples", [480000, 16000, 410, 47999])
def test_hubert_conv0_frames_matches_float64(native, dev, n_samples):
    """K13 (hubert_front.hip): Conv1d(1, 512, 10, stride 5, no bias) -> GroupNorm(512, 512) -> GELU (transformers'
    HubertGroupNormConvLayer behind pipeline.py:450) written as time-major bf16x3 planes.  Against the float64 graph; the error level
    of torch's fp32 graph on the same input.  30 s, 1 s, a clip of 81 frames, an odd length."""
    g = torch.Generator().manual_seed(n_samples)
    wav = torch.randn(n_samples, generator=g) * 0.3
    w = torch.randn(512, 1, 10, generator=g) * 0.4
    gamma, beta = torch.randn(512, generator=g), torch.randn(512, generator=g)
    ref = F.gelu(F.group_norm(F.conv1d(wav.double()[None, None], w.double(), stride=5), 512, gamma.double(), beta.double(), 1e-5))[0].t()
    wd = wav.to(dev)
    lib = F.gelu(F.group_norm(F.conv1d(wd[None, None], w.to(dev), stride=5), 512, gamma.to(dev), beta.to(dev), 1e-5))[0].t().cpu()
    ys, frames = native.hubert_conv0_frames_bf16x3(wd, w.to(dev), gamma.to(dev), beta.to(dev), 1e-5, stride=5)
    assert frames == ref.shape[0] and ys.shape[1] % 128 == 0 and ys.shape[1] >= frames
    got = ys[:, :frames].float().sum(0).cpu()
    err, err_lib = (got.double() - ref).abs().max().item(), (lib.double() - ref).abs().max().item()
    print(f"HuBERT layer 0 on {n_samples} samples: max abs err vs float64 {err:.2e} (torch fp32 graph: {err_lib:.2e})")
    assert err <= max(2.0 * err_lib, 2e-6 * ref.abs().max().item())
    ys2, _ = native.hubert_conv0_frames_bf16x3(wd, w.to(dev), gamma.to(dev), beta.to(dev), 1e-5, stride=5)
    assert torch.equal(ys2[:, :frames], ys[:, :frames])


@pytest.mark.parametrize("n_rows,k,m,act,with_res", [
    (1599, 768, 2304, "none", False), (1599, 768, 768, "none", True), (1599, 768, 3072, "gelu", False),
    (1599, 3072, 768, "none", True), (149, 768, 768, "gelu", True), (1, 512, 768, "none", False), (130, 16, 128, "none", False),
])
def test_linear_bf16x3_matches_float64(native, dev, n_rows, k, m, act, with_res):
    """gemmbf.hip, linear mode: the HuBERT projections (transformers' HubertModel behind pipeline.py:450) as exact bf16x3
    splits on the bf16 matrix cores.  Against float64; the relative RMS error must stay at the level torch's fp32 GEMM
    (hipBLASLt) leaves on the same operands (<= 2x + 1e-7).  Shapes: the four projection shapes at the cfg-2 frame count, a short clip,
    one row, the smallest legal k."""
    g = torch.Generator().manual_seed(n_rows + k + m)
    x = torch.randn(n_rows, k, generator=g)
    w = torch.randn(m, k, generator=g) / k ** 0.5
    b = torch.randn(m, generator=g)
    res = torch.randn(n_rows, m, generator=g) if with_res else None
    ref = F.linear(x.double(), w.double(), b.double())
    if act == "gelu":
        ref = F.gelu(ref)
    if with_res:
        ref = ref + res.double()
    a = native.gemm_bf16x3_pack_weight(w, dev)
    got = native.linear_bf16x3(x.to(dev), a, b.to(dev), m, act=act, res=res.to(dev) if with_res else None).cpu()
    lib = F.linear(x.to(dev), w.to(dev), b.to(dev))
    lib = (F.gelu(lib) if act == "gelu" else lib).cpu()
    if with_res:
        lib = lib + res
    rel = lambda t: ((t.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"linear {n_rows} x {k} -> {m} ({act}{', +res' if with_res else ''}): relative RMS error vs float64: bf16x3 {rel(got):.2e}, "
          f"torch fp32 {rel(lib):.2e}; max abs {(got.double() - ref).abs().max().item():.2e}")
    assert (got.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert rel(got) <= 2.0 * rel(lib) + 1e-7          # both at the fp32 rounding level (1-3e-7)


@pytest.mark.parametrize("c_in,c_out,k,stride,length,batch,act", [
    (512, 512, 3, 2, 5119, 1, "gelu"), (512, 512, 2, 2, 640, 2, "gelu"), (64, 128, 3, 1, 300, 1, "none"), (512, 512, 3, 2, 4, 1, "none"),
    (1, 512, 10, 5, 16000, 1, "none"), (1, 512, 10, 5, 4003, 1, "none"),      # HuBERT's first layer: one input channel, 10 taps, stride 5
    (32, 128, 1, 1, 1, 1, "none"),                                               # l_in = 1 with several channels (not the one-channel path)
])
def test_conv1d_bf16x3_matches_float64(native, dev, c_in, c_out, k, stride, length, batch, act):
    """gemmbf.hip, conv mode: HuBERT's feature-extractor convs (Conv1d(512, 512, k in {3, 2}, stride 2, no padding) + GELU)."""
    g = torch.Generator().manual_seed(c_in + c_out + k + length)
    x = torch.randn(batch, c_in, length, generator=g)
    w = torch.randn(c_out, c_in, k, generator=g) / (c_in * k) ** 0.5
    ref = conv1d_f64(x, w, None, stride=stride)
    if act == "gelu":
        ref = F.gelu(ref)
    a = native.gemm_bf16x3_pack_weight(w, dev)
    got = native.conv1d_bf16x3(x.to(dev), a, None, c_out, k, stride=stride, act=act).cpu()
    assert got.shape == ref.shape
    lib = F.conv1d(x.to(dev), w.to(dev), None, stride=stride)
    lib = (F.gelu(lib) if act == "gelu" else lib).cpu()
    rel = lambda t: ((t.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"conv1d {c_in}->{c_out} k {k} s {stride} L {length}: relative RMS error vs float64: bf16x3 {rel(got):.2e}, torch fp32 {rel(lib):.2e}")
    assert (got.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert rel(got) <= 1e-6          # 1536 sequentially accumulated terms: the level of conv.hip's direct fp32 form (4-10e-7)


def test_conv1d_winograd_f43_groups_still_pass():
    """7- and 11-tap layers default to F(4,4) groups; the F(4,3) form of the same kernel (RVC_WINO_R4=0, read once per
    process) stays covered by re-running the Winograd test above in a child process."""
    import os
    import subprocess
    import sys
    if os.environ.get("RVC_WINO_R4") == "0":
        pytest.skip("already the child")
    import __graft_entry__ as G
    if not os.path.isfile(G.LIB_ABLATE):
        pytest.skip("ablation library not built (the F(4,3) form of the 7- / 11-tap layers exists only there)")
    env = dict(os.environ, RVC_WINO_R4="0", RVC_AMD_LIB=G.LIB_ABLATE)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", "test_conv1d_winograd_matches_float64",
                        os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "12 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("c_in,c_out,h,w,batch", [
    (32, 32, 50, 64, 1), (64, 64, 37, 32, 2), (128, 128, 101, 16, 1), (256, 256, 51, 8, 1), (512, 512, 26, 4, 1),
    (256, 512, 101, 4, 1), (512, 256, 13, 8, 1), (16, 16, 9, 128, 1), (32, 16, 7, 128, 1), (16, 3, 11, 128, 1),
    (64, 32, 33, 64, 1), (16, 16, 1250, 128, 1), (32, 32, 700, 64, 1), (128, 64, 95, 32, 1), (256, 128, 47, 16, 2),
    (512, 512, 5, 4, 1), (16, 32, 3, 4, 1),
])
def test_conv2d_bf16x3_matches_float64(native, dev, c_in, c_out, h, w, batch):
    """K10b (conv2dbf.hip), the 3x3 conv of RMVPE's ConvBlockRes (RMVPE.py:13-60) as exact bf16x3 products: every level's
    (channels, row length) pair of the U-Net and the decoder's 2 C -> C first convs, heights that are not a multiple of the tile's
    rows, the 16- and 3-channel outputs (padded to 32 rows), the K-split deep levels, maps with more tiles than the chip has CUs
    (a workgroup walks several: the tap ring and the staging run on across tiles), maps smaller than one tile -- vs float64."""
    g = torch.Generator().manual_seed(c_in * 100 + c_out + h + w)
    x = torch.randn(batch, c_in, h, w, generator=g)
    wt = torch.randn(c_out, c_in, 3, 3, generator=g) / (c_in * 9) ** 0.5
    b = torch.randn(c_out, generator=g)
    res = torch.randn(batch, c_out, h, w, generator=g)
    assert native.conv2d_bf16x3_supported(c_in, c_out, h, w)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1)) + res.double()
    u = native.conv2d_bf16x3_pack_weight(wt, dev)
    got = native.conv2d_bf16x3_forward(x.to(dev), u, b.to(dev), c_out, relu=True, res=res.to(dev)).cpu()
    assert got.shape == ref.shape
    err = (got.double() - ref).abs().max().item()
    f32 = (F.relu(F.conv2d(x, wt, b, padding=1)) + res).double()
    assert err <= 4e-6 and err <= 4 * max((f32 - ref).abs().max().item(), 1e-7), err   # fp32-level, not "2e-5"
    plain_ref = F.conv2d(x.double(), wt.double(), None, padding=1)
    plain = native.conv2d_bf16x3_forward(x.to(dev), u, None, c_out).cpu()
    assert (plain.double() - plain_ref).abs().max().item() <= 4e-6
    again = native.conv2d_bf16x3_forward(x.to(dev), u, None, c_out).cpu()
    assert torch.equal(plain, again)                # split-K partials are summed in a fixed order


def test_conv2d_bf16x3_random_shape_sweep(native, dev):
    """K10b on 40 seeded random shapes -- every supported (output-channel class, row length) pair with ragged heights from 1 row up,
    batches of 1-3, maps on either side of "more tiles than CUs" (the persistent walk with XCD-contiguous ranges) and of the K-split
    threshold, with and without bias / ReLU / skip path -- against F.conv2d in float64."""
    rng = np.random.default_rng(20260)
    n_split = n_persist = 0
    for it in range(40):
        c_out = int(rng.choice([3, 16, 32, 64, 128, 256]))
        c_in = int(rng.choice([16, 32, 64, 128, 256]))
        max_w = 128 if c_out <= 32 else (128 if c_out <= 64 else 64)
        w = int(rng.choice([x for x in (4, 8, 16, 32, 64, 128) if x <= max_w]))
        batch = int(rng.integers(1, 4))
        h = int(rng.integers(1, max(2, min(3000, 600_000 // (w * max(c_in, c_out) * batch)))))
        if it % 8 == 0:
            c_in, c_out, w, batch, h = 16, 16, 128, 1, int(rng.integers(600, 1400))      # > 256 tiles: several per workgroup
        g = torch.Generator().manual_seed(1000 + it)
        x = torch.randn(batch, c_in, h, w, generator=g)
        wt = torch.randn(c_out, c_in, 3, 3, generator=g) / (c_in * 9) ** 0.5
        use_b, use_relu, use_res = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        b = torch.randn(c_out, generator=g) if use_b else None
        res = torch.randn(batch, c_out, h, w, generator=g) if use_res else None
        assert native.conv2d_bf16x3_supported(c_in, c_out, h, w), (c_in, c_out, h, w)
        ref = F.conv2d(x.double(), wt.double(), b.double() if use_b else None, padding=1)
        if use_relu: ref = F.relu(ref)
        if use_res: ref = ref + res.double()
        u = native.conv2d_bf16x3_pack_weight(wt, dev)
        got = native.conv2d_bf16x3_forward(x.to(dev), u, b.to(dev) if use_b else None, c_out, relu=use_relu,
                                           res=res.to(dev) if use_res else None).cpu()
        err = (got.double() - ref).abs().max().item()
        f32 = F.conv2d(x, wt, b, padding=1)                 # what fp32 arithmetic itself loses on this shape (grows with 9 C_in)
        if use_relu: f32 = F.relu(f32)
        if use_res: f32 = f32 + res
        # six fp32 accumulations per multiply-add instead of one: up to ~sqrt(6) x the rounding walk of a plain fp32 conv; K10's own gate is 2e-5
        assert err <= min(1.5e-5, max(4e-6, 6 * (f32.double() - ref).abs().max().item())), (it, c_in, c_out, h, w, batch, use_b, use_relu, use_res, err)
        need = ctypes.c_size_t()
        assert native._lib.rvc_conv2d_bf16x3_workspace_bytes(batch, c_in, c_out, h, w, ctypes.byref(need)) == 0
        n_split += need.value > 0
        n_persist += c_out <= 32 and batch * ((h * w + 255) // 256) > 256
    assert n_split >= 3 and n_persist >= 3, (n_split, n_persist)       # the sweep reached both regimes


@pytest.mark.parametrize("c_in,c_out,h,w,ks,batch", [
    (32, 32, 50, 64, 3, 1), (64, 64, 37, 32, 3, 2), (128, 128, 101, 16, 3, 1), (256, 256, 51, 8, 3, 1), (512, 512, 26, 4, 3, 1),
    (256, 512, 101, 4, 3, 1), (512, 256, 13, 8, 3, 1), (16, 16, 9, 128, 3, 1), (32, 16, 7, 128, 3, 1), (16, 3, 11, 128, 3, 1),
    (16, 32, 20, 64, 1, 1), (256, 512, 17, 4, 1, 1), (64, 32, 33, 64, 3, 1),
])
def test_conv2d_matches_float64(native, dev, c_in, c_out, h, w, ks, batch):
    """K10 (conv2d.hip), the conv of RMVPE's ConvBlockRes (RMVPE.py:13-60) with folded BatchNorm bias, ReLU and the skip path:
    every level's (channels, row length) pair of the U-Net, heights that are not a multiple of the block's rows, the 16- and
    3-channel outputs (padded to 32 rows), the K-split deep levels, 1x1 shortcuts -- against F.conv2d in float64."""
    g = torch.Generator().manual_seed(c_in * 100 + c_out + h + w)
    x = torch.randn(batch, c_in, h, w, generator=g)
    wt = torch.randn(c_out, c_in, ks, ks, generator=g) / (c_in * ks * ks) ** 0.5
    b = torch.randn(c_out, generator=g)
    res = torch.randn(batch, c_out, h, w, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=ks // 2)) + res.double()
    wp = native.conv2d_pack_weight(wt, dev)
    got = native.conv2d_forward(x.to(dev), wp, b.to(dev), c_out, ks, relu=True, res=res.to(dev)).cpu()
    assert got.shape == ref.shape
    err = (got.double() - ref).abs().max().item()
    assert err <= 2e-5, err
    plain = native.conv2d_forward(x.to(dev), wp, None, c_out, ks).cpu()
    assert (plain.double() - F.conv2d(x.double(), wt.double(), None, padding=ks // 2)).abs().max().item() <= 2e-5
    again = native.conv2d_forward(x.to(dev), wp, None, c_out, ks).cpu()
    assert torch.equal(plain, again)                # split-K partials are summed in a fixed order


# ---- K1 kNN ------------------------------------------------------------------------------------------
def _knn_case(native, dev, n_rows, n_q, seed):
    from oracle import rvc_oracle as O
    from rvc_amd.lib import synthetic as S
    big = S.synth_index(n_rows, seed=seed)
    rng = np.random.default_rng(seed + 100)
    q = (big[rng.integers(0, n_rows, n_q)] + rng.standard_normal((n_q, 768)).astype(np.float32) * 0.03).astype(np.float32)
    index = torch.from_numpy(big).to(dev)
    norms = native.knn_index_build(index)      # aux blob: row norms first, then statistics and the fp16 copy
    assert torch.allclose(norms[:4 * n_rows].view(torch.float32).cpu(), torch.from_numpy((big.astype(np.float64) ** 2).sum(1)).float(), rtol=1e-5)
    d2, ids = native.knn_search(index, norms, torch.from_numpy(q).to(dev))
    d2, ids = d2.cpu().numpy(), ids.cpu().numpy()
    d2_ref, ids_ref = O.knn_search(big, q, 8, np.float64)
    return big, q, d2, ids, d2_ref, ids_ref


@pytest.mark.parametrize("n_rows,n_q", [(4096, 64), (10007, 129), (100, 5), (50000, 300)])
def test_knn_ids_and_distances(native, dev, n_rows, n_q):
    big, q, d2, ids, d2_ref, ids_ref = _knn_case(native, dev, n_rows, n_q, seed=3)
    assert (np.diff(d2, axis=1) >= 0).all()  # ascending
    # ids bit-exact except documented near-ties.  ||x||^2 - 2 q.x + ||q||^2 (faiss' own BLAS formulation) rounds at
    # the granularity of the NORMS, not of d2: two candidates whose true distances differ by less than
    # 8 ulp_fp32(||q||^2 + ||x||^2) are a tie for this algorithm and may swap places.
    mism = ids != ids_ref
    if mism.any():
        qi, ki = np.nonzero(mism)
        true_d = ((q[qi].astype(np.float64) - big[ids[qi, ki]].astype(np.float64)) ** 2).sum(1)
        scale = (q[qi].astype(np.float64) ** 2).sum(1) + (big[ids[qi, ki]].astype(np.float64) ** 2).sum(1)
        assert np.all(np.abs(true_d - d2_ref[qi, ki]) <= 8 * 1.1920929e-07 * scale), "id mismatch that is not a near-tie"
    assert mism.mean() <= 0.01
    # ||x||^2 - 2q.x + ||q||^2 in fp32 (faiss' own form) cancels at the scale of the norms (~190 + 190 here): the absolute
    # error is a few ulp of that sum whatever d2 is
    assert np.allclose(d2, d2_ref, rtol=2e-4, atol=1e-3)


@pytest.mark.parametrize("dim,n_rows,n_q", [(256, 3000, 20), (256, 3000, 200), (768, 777, 33), (768, 5, 3), (768, 40, 64)])
def test_knn_edge_shapes_and_ties(native, dev, dim, n_rows, n_q):
    """v1 models (256-dim features), row counts that are not tile multiples, fewer rows than k, and exact duplicates:
    equal distances keep the lower row id first (faiss' IndexFlat order), missing neighbours come back as id -1."""
    g = torch.Generator().manual_seed(dim + n_rows + n_q)
    index = torch.randn(n_rows, dim, generator=g) * 0.4
    if n_rows >= 40:
        index[n_rows // 2] = index[3]            # exact duplicates of rows 3 and 7 further down
        index[n_rows - 1] = index[7]
    q = index[torch.randint(0, n_rows, (n_q,), generator=g)] + 0.02 * torch.randn(n_q, dim, generator=g)
    if n_rows >= 40:
        q[0], q[1] = index[3], index[7]          # distance 0 to both copies
    ix = index.to(dev)
    d2, ids = native.knn_search(ix, native.knn_index_norms(ix), q.to(dev))
    d2, ids = d2.cpu(), ids.cpu()
    # explicit sum of squared differences in float64: duplicates get EXACTLY equal distances (cdist's GEMM form does not)
    d_ref = torch.stack([((qi.double()[None, :] - index.double()) ** 2).sum(1) for qi in q])
    k_eff = min(8, n_rows)
    full = torch.argsort(d_ref, dim=1, stable=True)
    order = full[:, :k_eff]
    d_sorted = torch.gather(d_ref, 1, order)
    assert torch.allclose(d2[:, :k_eff].double(), d_sorted, rtol=2e-4, atol=2e-3)
    if n_rows < 8:
        assert (ids[:, n_rows:] == -1).all() and torch.isinf(d2[:, n_rows:]).all()
        assert (torch.sort(ids[:, :n_rows], 1).values == torch.arange(n_rows)).all()
    else:
        # ids equal wherever the float64 distances are separated by more than fp32 resolution of the norms
        d9 = torch.gather(d_ref, 1, full[:, :min(9, n_rows)])
        gaps = d9[:, 1:] - d9[:, :-1]
        gaps = torch.where(gaps == 0, torch.full_like(gaps, 1.0), gaps)      # exact ties are decided by the id rule
        gap_ok = gaps.min(1).values > 5e-3
        assert gap_ok.float().mean() > 0.5
        assert (ids[gap_ok] == order[gap_ok]).all()
    if n_rows >= 40:
        assert ids[0, 0] == 3 and ids[0, 1] == n_rows // 2 and ids[1, 0] == 7 and ids[1, 1] == n_rows - 1


def test_knn_golden_and_blend(native, dev):
    from rvc_amd.lib import synthetic as S
    g = load_golden("knn")
    big = S.synth_index(int(g["index_rows"]), seed=int(g["index_seed"]))
    index = torch.from_numpy(big).to(dev)
    norms = native.knn_index_norms(index)
    q = torch.from_numpy(g["q"]).to(dev)
    d2, ids = native.knn_search(index, norms, q)
    assert np.array_equal(ids.cpu().numpy(), g["ids"])
    blended = native.knn_blend(index, q, d2, ids, 0.75).cpu().numpy()
    assert np.abs(blended - g["blended"][0]).max() <= 1e-4


def _both_regimes(native, index, q):
    aux = native.knn_index_build(index)
    out = []
    for mode in (1, 2):       # 1: exact fp32 GEMM / streaming regimes, 2: fp16-screened regime
        native.knn_set_mode(mode)
        try:
            d2, ids = native.knn_search(index, aux, q)
            torch.cuda.synchronize()
        finally:
            native.knn_set_mode(0)
        out.append((d2.cpu(), ids.cpu()))
    return out


@pytest.mark.parametrize("n_rows,n_q,dim", [(100_000, 1599, 768), (40_000, 599, 768), (20_000, 300, 256), (16_500, 65, 768),
                                            (4_096, 149, 768), (70_001, 257, 768)])
def test_knn_screened_regime_equals_exact_regime(native, dev, n_rows, n_q, dim):
    """The fp16 matrix-core screening pass only proposes candidates; ids AND distances must be bit-identical to the exact
    fp32 regime's (pipeline.py:497-499 semantics), at the BASELINE cfg-2 shape, a v1 (256-dim) index, ragged tile counts."""
    from rvc_amd.lib import synthetic as S
    rng = np.random.default_rng(n_rows + n_q)
    big = S.synth_index(n_rows, dim=dim, seed=1)
    q = (big[rng.integers(0, n_rows, n_q)] + rng.standard_normal((n_q, dim)).astype(np.float32) * 0.03).astype(np.float32)
    q[::3] = rng.standard_normal((len(q[::3]), dim)).astype(np.float32)      # queries far from every cluster, too
    index, qd = torch.from_numpy(big).to(dev), torch.from_numpy(q).to(dev)
    (d_exact, i_exact), (d_scr, i_scr) = _both_regimes(native, index, qd)
    assert torch.equal(i_exact, i_scr)
    assert torch.equal(d_exact, d_scr)
    # and both are the true neighbours: float64 direct-difference distances of the returned rows, ascending, and no other
    # row of a 64-query sample is closer than the 8th
    sel = rng.integers(0, n_q, 64)
    x64, q64 = big.astype(np.float64), q[sel].astype(np.float64)
    d_all = (q64 ** 2).sum(1)[:, None] - 2 * q64 @ x64.T + (x64 ** 2).sum(1)[None, :]
    d8 = np.sort(d_all, axis=1)[:, 7]
    got = np.take_along_axis(d_all, i_scr.numpy()[sel], axis=1)
    assert np.all(got[:, 7] <= d8 * (1 + 1e-6) + 1e-6)
    assert np.allclose(d_scr.numpy()[sel], got, rtol=1e-5, atol=1e-5)


def test_knn_screened_regime_survives_hostile_data(native, dev):
    """Data built to defeat the screening pass -- thousands of exact duplicates of the nearest row (every candidate list
    overflows), magnitudes fp16 cannot hold, an all-zero index -- must still give the exact regime's answer (the
    overflowing queries are answered by the in-launch exact scan)."""
    g = torch.Generator().manual_seed(9)
    n, dim = 20_000, 768
    index = torch.randn(n, dim, generator=g) * 0.3
    index[5000:9000] = index[17]                                   # 4000 copies: > capacity of a candidate list
    q = index[torch.randint(0, n, (130,), generator=g)] + 0.01 * torch.randn(130, dim, generator=g)
    q[0] = index[17]
    (d_exact, i_exact), (d_scr, i_scr) = _both_regimes(native, index.to(dev), q.to(dev))
    assert torch.equal(i_exact, i_scr) and torch.equal(d_exact, d_scr)
    assert i_scr[0].tolist() == [17] + list(range(5000, 5007))     # ties -> lower id first
    big = torch.randn(n, dim, generator=g) * 3e4                   # |x| up to ~1.2e5: beyond fp16
    (d_exact, i_exact), (d_scr, i_scr) = _both_regimes(native, big.to(dev), (big[:100] * 1.001).to(dev))
    assert torch.equal(i_exact, i_scr) and torch.equal(d_exact, d_scr) and (i_scr[:, 0] == torch.arange(100)).all()
    zero = torch.zeros(n, dim)
    (d_exact, i_exact), (d_scr, i_scr) = _both_regimes(native, zero.to(dev), q.to(dev))
    assert torch.equal(i_exact, i_scr) and torch.equal(d_exact, d_scr)
    assert (i_scr == torch.arange(8)[None, :]).all()


def test_knn_screened_regime_fp16_subnormal_rows(native, dev):
    """The screening bound measures the fp16 rounding residuals ||x - x~|| with scalar conversions that keep subnormals; it
    is only a bound if the matrix cores do not flush fp16 subnormal operands to zero.  Rows (and queries) whose elements sit
    in the fp16 subnormal range (|x| < 6.1e-5, down to below its smallest subnormal 6e-8) must give the exact regime's
    answer bit for bit; so must a mixed index where only a sub-cluster is that small."""
    g = torch.Generator().manual_seed(31)
    n, dim = 20_000, 768
    tiny = torch.randn(n, dim, generator=g) * 2e-5                  # almost every element subnormal in fp16
    tiny[::7] *= 1e-3                                               # ... and some below the smallest subnormal
    q = tiny[torch.randint(0, n, (140,), generator=g)] + 2e-6 * torch.randn(140, dim, generator=g)
    (d_exact, i_exact), (d_scr, i_scr) = _both_regimes(native, tiny.to(dev), q.to(dev))
    assert torch.equal(i_exact, i_scr) and torch.equal(d_exact, d_scr)
    mixed = torch.randn(n, dim, generator=g) * 0.3
    mixed[3000:6000] = torch.randn(3000, dim, generator=g) * 3e-5   # a cluster at the origin, subnormal in fp16
    q = torch.cat([mixed[3000:3070] * 1.01, mixed[torch.randint(0, n, (70,), generator=g)] * 0.99])
    (d_exact, i_exact), (d_scr, i_scr) = _both_regimes(native, mixed.to(dev), q.to(dev))
    assert torch.equal(i_exact, i_scr) and torch.equal(d_exact, d_scr)
    assert (i_scr[:70, 0] == torch.arange(3000, 3070)).all()


@pytest.mark.parametrize("n_q", [96, 1599])
def test_knn_screened_equals_exact_at_two_million_rows(native, dev, n_q):
    """BASELINE cfg 5's index size (2 000 000 x 768, 6.1 GB): screened == exact regime, ids and distances bit-identical, at a
    short clip's query count and at the full 30 s utterance's 1599."""
    import bench
    index = bench.synth_index_device(torch, 2_000_000, dev, seed=0)
    g = torch.Generator(device=dev).manual_seed(5)
    pick = torch.randint(0, index.shape[0], (n_q,), device=dev, generator=g)
    q = index[pick] + 0.03 * torch.randn(n_q, 768, device=dev, generator=g)
    q[::3] = torch.randn(len(q[::3]), 768, device=dev, generator=g)
    (d_exact, i_exact), (d_scr, i_scr) = _both_regimes(native, index, q)
    assert torch.equal(i_exact, i_scr)
    assert torch.equal(d_exact, d_scr)
    del index


# ---- K4 log-mel --------------------------------------------------------------------------------------
def test_logmel_golden(native, dev):
    g = load_golden("logmel")
    mel, t = native.logmel_rmvpe(torch.from_numpy(g["audio"]).to(dev), pad_to=32)
    assert t == 101 and mel.shape == (1, 128, 128)
    got = mel.cpu().numpy()
    assert np.abs(got[:, :, :101] - g["mel"]).max() <= 2e-3  # log domain; DFT-by-GEMM vs FFT
    ref_pad = F.pad(torch.from_numpy(g["mel"]), (0, 27), mode="reflect").numpy()
    assert np.abs(got - ref_pad).max() <= 2e-3


def test_logmel_long_batch(native, dev):
    from oracle import rvc_oracle as O
    from rvc_amd.lib import synthetic as S
    a = np.stack([S.synth_audio(80000, seed=s) for s in (1, 2)]).astype(np.float32)
    mel, t = native.logmel_rmvpe(torch.from_numpy(a).to(dev), pad_to=32)
    ref = O.logmel_rmvpe(torch.from_numpy(a)).numpy()
    assert t == ref.shape[-1] == 501
    assert np.abs(mel.cpu().numpy()[:, :, :t] - ref).max() <= 2e-3


# ---- K2/K3 decoder -----------------------------------------------------------------------------------
def _decoder_inputs(ref_inputs, T, batch=1):
    feats, f0c, f0f = ref_inputs
    f0 = torch.from_numpy(f0f[:T]).float().unsqueeze(0).repeat(batch, 1)
    if batch > 1:
        f0[1] = torch.roll(f0[1], 7) * 1.3
    return f0


@pytest.mark.parametrize("tag,sr,voc", [("nsf48", 48000, "HiFi-GAN"), ("nsf40", 40000, "HiFi-GAN"),
                                        ("nsf32", 32000, "HiFi-GAN"), ("mrf48", 48000, "MRF HiFi-GAN"),
                                        ("refine48", 48000, "RefineGAN"), ("refine40", 40000, "RefineGAN")])
def test_decoder_matches_oracle(native, dev, ref_inputs, tag, sr, voc):
    from oracle import rvc_oracle as O
    from rvc_amd.lib import synthetic as S
    from rvc_amd.lib.algorithm.weights import fold_weight_norm
    T, batch = 64, 2
    cpt = S.make_synth_checkpoint(sr, voc, seed=0)
    w = O.fold_weight_norm(cpt["weight"])
    rates, ksizes = cpt["config"][12], cpt["config"][14]
    upp = int(np.prod(rates))
    gen = torch.Generator().manual_seed(7)
    z = torch.randn(batch, 192, T, generator=gen)
    g = torch.randn(batch, 256, 1, generator=gen)
    f0 = _decoder_inputs(ref_inputs, T, batch)
    dim = 9 if voc.startswith("MRF") else 1
    src_rand = torch.rand(batch, dim, generator=gen)
    src_randn = torch.randn(batch, T * upp, dim, generator=gen)
    refine = voc == "RefineGAN"
    adain = []
    if refine:
        length, ch = T, 512
        for r in rates:
            length, ch = length * r, ch // 2
            adain += [torch.randn(batch, ch, length, generator=gen) for _ in range(6)]
    outs = []
    for b in range(batch):
        if refine:
            noise = O.ListNoise([src_rand[b:b + 1].clone(), src_randn[b:b + 1]] + [a[b:b + 1] for a in adain])
            o = O.decoder_refine(w, z[b:b + 1], f0[b:b + 1], g[b:b + 1], rates, sr, noise)
        elif dim == 1:
            noise = O.ListNoise([torch.zeros(1, 1, 1), src_randn[b:b + 1]])
            o = O.decoder_nsf(w, z[b:b + 1], f0[b:b + 1], g[b:b + 1], rates, ksizes, sr, noise)
        else:
            noise = O.ListNoise([src_rand[b:b + 1].clone(), src_randn[b:b + 1]])
            o = O.decoder_mrf(w, z[b:b + 1], f0[b:b + 1], g[b:b + 1], rates, ksizes, sr, noise)
        outs.append(o)
    ref = torch.cat(outs, 0).numpy()

    folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
    dec = native.Decoder(voc, sr, folded, upsample_rates=rates, upsample_kernel_sizes=ksizes)
    assert dec.upp == upp
    adain_flat = torch.cat([a.reshape(-1) for a in adain]).to(dev) if refine else None
    out = dec.forward(z.to(dev), f0.to(dev), g[:, :, 0].to(dev), src_randn=src_randn.to(dev),
                      src_rand=src_rand.to(dev), adain_randn=adain_flat).cpu().numpy()
    assert out.shape == ref.shape
    err = rms(out - ref)
    assert rms(ref) > 0.02
    assert err <= 5e-5, (tag, err)  # gate is 1e-3 (north_star); fp32 MFMA vs torch-CPU fp32 sits ~1e-6


def test_decoder_full_synth_golden(native, dev, ref_inputs):
    """Decoder fed with the golden z of the reference run reproduces the reference waveform."""
    from oracle import rvc_oracle as O
    from rvc_amd.lib import synthetic as S
    from rvc_amd.lib.algorithm.weights import fold_weight_norm
    g = load_golden("synth_nsf48")
    feats, f0c, f0f = ref_inputs
    T = int(g["T"])
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
    folded_all = fold_weight_norm(cpt["weight"])
    folded = {k[4:]: v for k, v in folded_all.items() if k.startswith("dec.")}
    dec = native.Decoder("HiFi-GAN", 48000, folded)
    # replay the reference's draws: randn_like(m_p), rand(1,1,1), randn_like(sine)
    torch.manual_seed(int(g["seed"]))
    torch.randn(1, 192, T)
    torch.rand(1, 1, 1)
    src_randn = torch.randn(1, T * 480, 1)
    z = torch.from_numpy(g["z"]).unsqueeze(0)
    gvec = folded_all["emb_g.weight"][int(g["sid"])].view(1, 256)
    f0 = torch.from_numpy(f0f[:T]).float().unsqueeze(0)
    out = dec.forward(z.to(dev), f0.to(dev), gvec.to(dev), src_randn=src_randn.to(dev)).cpu().numpy()[0, 0]
    assert rms(out - g["o"]) <= 5e-5, rms(out - g["o"])


@pytest.mark.parametrize("co", ["gemmbf", "winobf2", "winobf", "knn_screen", "attention_bf", "resblock_bf", "resblock_bf1", "convbf1", "upsbf", "linear_presplit", "posconv", "conv2dbf"])
@pytest.mark.parametrize("k,c", [(11, 128), (3, 64), (7, 32)])
def test_fp32_winograd_next_to_the_bf16_gemm_is_bit_exact(native, dev, k, c, co):
    """Regression test of profiles/r03_mfma_cohabitation.txt / r04_mfma_cohabitation.txt: while one thread launches a kernel that
    issues bf16 / fp16 matrix instructions in a loop on its own stream -- gemmbf.hip, winobf2.hip, the kNN screening pass, HuBERT's
    bf16x3 attention -- the
    bf16x3 attention, the fused ResBlock pair (K3f), HuBERT's pre-split projections (K12) -- the
    fp32 Winograd kernel on another stream must return bit-identical results every time.  It does because every such kernel asks
    for a CU's whole LDS (common.h: LDS_WHOLE_CU) and so never shares one; next to gemmbf's first form (two 60 KiB blocks per CU)
    300 of 300 runs of this loop came back wrong by up to 2.2."""
    import threading
    g = torch.Generator().manual_seed(k * 1000 + c)
    a = native.gemm_bf16x3_pack_weight(torch.randn(512, 512, 3, generator=g) * 0.03, dev)
    xg = torch.randn(1, 512, 51000, generator=g).to(dev)
    if co == "winobf2":
        ub = native.conv1d_winobf_pack_weight(torch.randn(128, 128, 11, generator=g) * 0.03, dev)
        xb = torch.randn(1, 128, 100000, generator=g).to(dev)
        bb = torch.zeros(128, device=dev)
    if co == "winobf":       # K3x: the 64-row form (c_out % 128 != 0)
        ub = native.conv1d_winobf_pack_weight(torch.randn(64, 64, 11, generator=g) * 0.03, dev)
        xb = torch.randn(1, 64, 200000, generator=g).to(dev)
        bb = torch.zeros(64, device=dev)
    if co == "convbf1":      # K3d
        ud = native.conv1d_bf16w_pack_weight(torch.randn(256, 256, 7, generator=g) * 0.03, dev)
        xd1 = torch.randn(1, 256, 40000, generator=g).to(dev)
        yd1 = torch.empty_like(xd1)
    if co == "upsbf":        # K3u
        pk = native.upsample_bf16x3_pack_weight(torch.randn(256, 128, 20, generator=g) * 0.03, torch.randn(128, 1, 8, generator=g) * 0.1,
                                                torch.zeros(128), 10, 4, dev)
        xu = torch.randn(1, 256, 8000, generator=g).to(dev)
        hu = torch.randn(1, 8000 * 40, generator=g).to(dev)
    if co == "conv2dbf":     # K10b
        u2 = native.conv2d_bf16x3_pack_weight(torch.randn(64, 64, 3, 3, generator=g) * 0.04, dev)
        x2 = torch.randn(1, 64, 752, 32, generator=g).to(dev)
        y2 = torch.empty_like(x2)
    if co == "posconv":      # K14
        pw = native.posconv_bf16x3_pack_weight(torch.randn(768, 48, 128, generator=g) * 0.02, 16, dev)
        pb = torch.zeros(768, device=dev)
        pxp = torch.randn(1599, 768, generator=g).to(dev)
    if co == "attention_bf":
        qkv = (torch.randn(1, 1599, 3 * 12 * 64, generator=g) * 1.5).to(dev)
    if co in ("resblock_bf", "resblock_bf1"):
        up = native.resblock_bf16x3_pack_weight(torch.randn(32, 32, 7, generator=g) * 0.03, torch.randn(32, 32, 7, generator=g) * 0.03, dev,
                                                bf16_taps=co == "resblock_bf1")
        xp = torch.randn(1, 32, 400000, generator=g).to(dev)
        yp = torch.empty_like(xp)
    if co == "linear_presplit":
        al = native.gemm_bf16x3_pack_weight(torch.randn(3072, 768, generator=g) * 0.03, dev)
        xl = native.split_rows_bf16x3(torch.randn(1599, 768, generator=g).to(dev))
    if co == "knn_screen":
        index = torch.randn(50000, 768, generator=g).to(dev)
        norms = native.knn_index_norms(index)
        q = torch.randn(600, 768, generator=g).to(dev)
    L = 60000 if c >= 64 else 400000
    x = torch.randn(1, c, L, generator=g).to(dev)
    res = torch.randn(1, c, L, generator=g).to(dev)
    bias = torch.randn(c, generator=g).to(dev)
    u = native.conv1d_wino_pack_weight(torch.randn(c, c, k, generator=g) * 0.03, dev)
    ref = native.conv1d_wino_forward(x, u, bias, c, k, 1, 0.1, res=res).clone()
    torch.cuda.synchronize()
    state = {"stop": False, "bad": 0}

    def co_runner():
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            while not state["stop"]:
                if co == "gemmbf":
                    native.conv1d_bf16x3(xg, a, None, 512, 3, stride=2, act="gelu")
                elif co == "winobf2":
                    native.conv1d_winobf_forward(xb, ub, bb, 128, 11, 1, 0.1)
                elif co == "attention_bf":
                    native.attention_qkv(qkv, 12, 0.125)
                elif co == "winobf":
                    native.conv1d_winobf_forward(xb, ub, bb, 64, 11, 1, 0.1)
                elif co == "convbf1":
                    native.conv1d_bf16w_forward(xd1, ud, None, 7, 3, 0.1, out=yd1)
                elif co == "upsbf":
                    native.upsample_bf16x3_forward(xu, hu, pk, 128, 10, 20, 5, 4, 2)
                elif co == "posconv":
                    native.posconv_gelu_bf16x3(pxp, pw, pb, 16, 128, 64)
                elif co == "conv2dbf":
                    native.conv2d_bf16x3_forward(x2, u2, None, 64, relu=True, out=y2)
                elif co in ("resblock_bf", "resblock_bf1"):
                    native.resblock_bf16x3_forward(xp, up, None, None, 7, 3, 0.1, out=yp, bf16_taps=co == "resblock_bf1")
                elif co == "linear_presplit":
                    native.linear_bf16x3_presplit(xl, al, None, 1599, 3072, "gelu_planes")
                else:
                    native.knn_search(index, norms, q)
                st.synchronize()

    def victim():
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            for _ in range(100):
                out = native.conv1d_wino_forward(x, u, bias, c, k, 1, 0.1, res=res)
                st.synchronize()
                state["bad"] += int((out != ref).any().item())
        state["stop"] = True

    th = [threading.Thread(target=co_runner), threading.Thread(target=victim)]
    for t in th: t.start()
    for t in th: t.join()
    assert state["bad"] == 0, f"{state['bad']} of 100 Winograd launches changed next to the bf16 GEMM kernel"


def test_decoder_forwards_on_two_streams_are_bit_exact(native, dev):
    """Two host threads run the same decoder handle on their own streams (what convert_batch does): every output must equal
    its one-at-a-time reference BIT FOR BIT.  This is the regression test of a hardware interaction found in round 3: a
    workgroup issuing bf16 matrix instructions (gemmbf.hip) that shares a CU with a workgroup of the fp32 Winograd kernel
    (wino.hip) corrupts the latter's results, so no default path may put such kernels next to the vocoder."""
    import threading
    from rvc_amd.lib import synthetic as S
    from rvc_amd.lib.algorithm.weights import fold_weight_norm
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
    folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
    dec = native.Decoder("HiFi-GAN", 48000, folded)
    Ts = (500, 800)
    ins = []
    for i, T in enumerate(Ts):
        g = torch.Generator(device=dev).manual_seed(i)
        ins.append((torch.randn(1, 192, T, device=dev, generator=g), torch.full((1, T), 220.0, device=dev),
                    torch.randn(1, 256, device=dev, generator=g), torch.zeros(1, T * 480, 1, device=dev), torch.zeros(1, 1, device=dev)))
    refs = [dec.forward(z, f0, gv, src_randn=nz, src_rand=rnd).clone() for z, f0, gv, nz, rnd in ins]
    torch.cuda.synchronize()
    bad = [0, 0]

    def worker(i):
        st = torch.cuda.Stream(device=dev)
        z, f0, gv, nz, rnd = ins[i]
        with torch.cuda.stream(st):
            for _ in range(25):
                out = dec.forward(z, f0, gv, src_randn=nz, src_rand=rnd)
                st.synchronize()
                bad[i] += int((out != refs[i]).any().item())

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert bad == [0, 0], f"decoder outputs changed under concurrency in {bad} of 25 runs per thread"


@pytest.mark.parametrize("voc", ["HiFi-GAN", "MRF HiFi-GAN"])
def test_decoder_branches_on_side_streams_are_bit_exact(native, dev, voc):
    """rvc_decoder_set_branch_parallel: the ResBlock branches of a short stage on 1 or 2 side streams of the handle
    (hifigan_nsf.py:195-203 sums them; they depend on the stage's input only) must give the one-stream waveform BIT FOR BIT --
    same kernels, same order of the additions -- from one thread and from two threads on their own streams."""
    import threading
    from rvc_amd.lib import synthetic as S
    from rvc_amd.lib.algorithm.weights import fold_weight_norm
    cpt = S.make_synth_checkpoint(48000, voc, seed=0)
    folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
    dec = native.Decoder(voc, 48000, folded)
    dim = 9 if voc.startswith("MRF") else 1
    ins = []
    for i, T in enumerate((301, 1000)):       # 301: every stage short; 1000: the last two stages stay on one stream
        g = torch.Generator(device=dev).manual_seed(i)
        ins.append((torch.randn(1, 192, T, device=dev, generator=g), torch.full((1, T), 220.0, device=dev),
                    torch.randn(1, 256, device=dev, generator=g), torch.randn(1, T * 480, dim, device=dev, generator=g),
                    torch.rand(1, dim, device=dev, generator=g)))
    refs = [dec.forward(z, f0, gv, src_randn=nz, src_rand=rnd).clone() for z, f0, gv, nz, rnd in ins]
    for n_side in (1, 2, -1):
        dec.set_branch_parallel(n_side)
        for (z, f0, gv, nz, rnd), ref in zip(ins, refs):
            for _ in range(3):
                assert torch.equal(dec.forward(z, f0, gv, src_randn=nz, src_rand=rnd), ref), f"{n_side} side stream(s), T = {z.shape[2]}"
    bad = [0, 0]

    def worker(i):
        st = torch.cuda.Stream(device=dev)
        z, f0, gv, nz, rnd = ins[i]
        with torch.cuda.stream(st):
            for _ in range(15):
                out = dec.forward(z, f0, gv, src_randn=nz, src_rand=rnd)
                st.synchronize()
                bad[i] += int((out != refs[i]).any().item())

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    dec.set_branch_parallel(0)
    assert bad == [0, 0], f"outputs changed with the branches on side streams in {bad} of 15 runs per thread"


@pytest.mark.parametrize("b,c,length", [(1, 512, 95999), (2, 64, 1003), (1, 8, 4)])
def test_rownorm_gelu_matches_float64(native, dev, b, c, length):
    """rvc_rownorm_gelu_f32 vs float64 GroupNorm(num_groups = channels) + exact GELU (HuBERT's first layer, pipeline.py:450)."""
    g = torch.Generator().manual_seed(b * 7 + c + length)
    x = torch.randn(b, c, length, generator=g) * 3.0 + 0.7
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    ref = F.gelu(F.group_norm(x.double(), c, gamma.double(), beta.double(), 1e-5))
    got = native.rownorm_gelu_(x.to(dev).clone(), gamma.to(dev), beta.to(dev), 1e-5).cpu()
    lib = F.gelu(F.group_norm(x.to(dev), c, gamma.to(dev), beta.to(dev), 1e-5)).cpu()
    err, err_lib = (got.double() - ref).abs().max().item(), (lib.double() - ref).abs().max().item()
    print(f"rownorm + gelu [{b}, {c}, {length}]: max abs error vs float64 {err:.2e} (torch fp32: {err_lib:.2e})")
    assert err <= 2e-6 * max(1.0, ref.abs().max().item())


# ---- K5 BiGRU ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("multi_cu", [False, True])
@pytest.mark.parametrize("batch,steps", [(1, 96), (2, 333), (1, 3232)])
def test_bigru_matches_torch_gru(native, dev, batch, steps, multi_cu):
    torch.manual_seed(5)
    gru = torch.nn.GRU(384, 256, num_layers=1, batch_first=True, bidirectional=True).eval()
    x = torch.randn(batch, steps, 384)
    with torch.no_grad():
        ref = gru(x)[0]
    sd = gru.state_dict()
    wih = torch.cat([sd["weight_ih_l0"], sd["weight_ih_l0_reverse"]], 0)
    bih = torch.cat([sd["bias_ih_l0"], sd["bias_ih_l0_reverse"]], 0)
    gi = F.linear(x, wih, bih).view(batch, steps, 2, 768)
    whh_t = torch.stack([sd["weight_hh_l0"].t(), sd["weight_hh_l0_reverse"].t()], 0).contiguous()
    bhh = torch.stack([sd["bias_hh_l0"], sd["bias_hh_l0_reverse"]], 0).contiguous()
    out = native.bigru_forward(gi.to(dev), whh_t.to(dev), bhh.to(dev), multi_cu=multi_cu).cpu()
    assert torch.isfinite(out).all()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 2e-5


def _gru_case(batch, steps, seed=5):
    torch.manual_seed(seed)
    gru = torch.nn.GRU(384, 256, num_layers=1, batch_first=True, bidirectional=True).eval()
    x = torch.randn(batch, steps, 384)
    with torch.no_grad():
        ref = gru(x)[0]
    sd = gru.state_dict()
    wih = torch.cat([sd["weight_ih_l0"], sd["weight_ih_l0_reverse"]], 0)
    bih = torch.cat([sd["bias_ih_l0"], sd["bias_ih_l0_reverse"]], 0)
    gi = F.linear(x, wih, bih).view(batch, steps, 2, 768)
    whh_t = torch.stack([sd["weight_hh_l0"].t(), sd["weight_hh_l0_reverse"].t()], 0).contiguous()
    bhh = torch.stack([sd["bias_hh_l0"], sd["bias_hh_l0_reverse"]], 0).contiguous()
    return gi, whh_t, bhh, ref


def test_bigru_while_another_stream_saturates_the_cus(native, dev):
    """The 8-workgroup exchange kernel launched while a second stream keeps every CU busy with conv tiles (what two
    utterances in flight do, RMVPE.py:515-536 under VoiceConverter.convert_batch): late-starting workgroups must
    either rendezvous in time or be recomputed by the single-workgroup kernel -- the result is the GRU's either way."""
    gi, whh_t, bhh, ref = _gru_case(1, 3232)
    gi, whh_t, bhh = gi.to(dev), whh_t.to(dev), bhh.to(dev)
    C, L = 128, 383_760
    x = torch.randn(1, C, L, device=dev)
    y = torch.empty_like(x)
    w = native.conv1d_pack_weight(torch.randn(C, C, 11) * 0.02, dev)
    bias = torch.zeros(C, device=dev)
    hog = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for rep in range(3):
        with torch.cuda.stream(hog):
            for _ in range(12):                       # ~13 ms of back-to-back 3000-tile launches
                native.conv1d_forward_into(x, w, bias, C, 11, 1, 0.1, out=y)
        out = native.bigru_forward(gi, whh_t, bhh, multi_cu=True)
        redone = native.bigru_redone(1, dev)
        torch.cuda.synchronize()
        err = (out.cpu() - ref).abs().max().item()
        print(f"bigru under load, rep {rep}: max err {err:.2e}, sequences recomputed: {redone}")
        assert torch.isfinite(out).all() and err <= 2e-5


def test_bigru_timeout_is_recomputed_not_poisoned(native, dev):
    """Force every rendezvous wait to give up after one poll (rvc_bigru_set_spin_limit): the in-call single-workgroup
    recompute must restore the exact result, for every batch item and direction, and report itself."""
    gi, whh_t, bhh, ref = _gru_case(2, 333)
    native.bigru_set_spin_limit(1)
    try:
        out = native.bigru_forward(gi.to(dev), whh_t.to(dev), bhh.to(dev), multi_cu=True)
        redone = native.bigru_redone(2, dev)
    finally:
        native.bigru_set_spin_limit(0)
    assert redone >= 1, "a 1-poll bound should have timed out somewhere"
    assert (out.cpu() - ref).abs().max().item() <= 2e-5
    out = native.bigru_forward(gi.to(dev), whh_t.to(dev), bhh.to(dev), multi_cu=True)
    assert native.bigru_redone(2, dev) == 0 and (out.cpu() - ref).abs().max().item() <= 2e-5


# ---- K7 attention ------------------------------------------------------------------------------------
@pytest.mark.parametrize("batch,frames,heads,hd", [(1, 1, 12, 64), (1, 31, 2, 64), (2, 97, 12, 64), (1, 1599, 12, 64),
                                                   (1, 64, 3, 64), (1, 200, 2, 96), (2, 333, 2, 96),
                                                   (1, 256, 1, 64), (1, 257, 5, 64), (3, 500, 4, 64), (1, 4799, 12, 64)])
def test_attention_matches_float64_softmax(native, dev, batch, frames, heads, hd):
    """softmax(q k^T / sqrt(d)) v against a float64 evaluation of the same formula (what transformers' HubertAttention
    computes, modeling_hubert.py eager path), on the fused-projection layout [B, T, 3, H, d]; key splits included.  Head dim 64
    runs K7b (attention_bf_kernel: bf16x3 products, 8-wave workgroups of 256 queries -- whole / partial / single workgroups per
    head, 1-8 key splits, a 45 s utterance's 4799 frames), head dim 96 K7."""
    torch.manual_seed(frames)
    qkv = torch.randn(batch, frames, 3 * heads * hd) * 1.5
    got = native.attention_qkv(qkv.to(dev), heads, hd ** -0.5).cpu()
    v = qkv.double().view(batch, frames, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(v[0] @ v[1].transpose(-1, -2) * hd ** -0.5, -1) @ v[2]).transpose(1, 2).reshape(batch, frames, -1)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    # tolerance: fp32 dot products of 64 terms and a 2^x hardware exp (1 ulp) under a row sum -> a few 1e-6 relative
    assert (got.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("batch,frames,heads,hd", [(1, 5, 2, 96), (1, 64, 2, 96), (2, 171, 2, 96), (1, 3198, 2, 96), (1, 100, 3, 64)])
def test_relative_attention_matches_oracle(native, dev, batch, frames, heads, hd):
    """TextEncoder attention with window-10 relative-position terms against the oracle's restatement of
    attentions.py:101-180 (pad/reshape formulation, float64)."""
    from oracle import rvc_oracle as O
    torch.manual_seed(frames + hd)
    c = heads * hd
    x = torch.randn(batch, c, frames)
    w = {"a.emb_rel_k": torch.randn(1, 21, hd) * hd ** -0.5, "a.emb_rel_v": torch.randn(1, 21, hd) * hd ** -0.5,
         "a.conv_o.weight": torch.eye(c).unsqueeze(-1), "a.conv_o.bias": torch.zeros(c)}
    for n in "qkv":
        w[f"a.conv_{n}.weight"] = torch.randn(c, c, 1) * c ** -0.5
        w[f"a.conv_{n}.bias"] = torch.randn(c) * 0.1
    ref = O._rel_attention(x.double(), {k: v.double() for k, v in w.items()}, "a", n_heads=heads)
    qkv = F.linear(x.transpose(1, 2), torch.cat([w[f"a.conv_{n}.weight"][:, :, 0] for n in "qkv"], 0),
                   torch.cat([w[f"a.conv_{n}.bias"] for n in "qkv"], 0)).contiguous()
    got = native.attention_qkv(qkv.to(dev), heads, hd ** -0.5, w["a.emb_rel_k"][0].contiguous().to(dev),
                               w["a.emb_rel_v"][0].contiguous().to(dev)).cpu().transpose(1, 2)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert (got.double() - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())


# ---- K8 conv epilogue --------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 16, 64, 128), (2, 3, 7, 4), (1, 256, 101, 4), (1, 1, 4)])
def test_bias_relu_add_bit_exact(native, dev, shape):
    torch.manual_seed(len(shape))
    x, res, bias = torch.randn(*shape), torch.randn(*shape), torch.randn(shape[1])
    bshape = [1, -1] + [1] * (len(shape) - 2)
    want = torch.relu(x + bias.view(bshape)) + res
    got = native.bias_relu_add_(x.clone().to(dev), bias.to(dev), res.to(dev)).cpu()
    assert torch.equal(got, want)                      # same three fp32 operations in the same order
    assert torch.equal(native.bias_relu_add_(x.clone().to(dev), bias.to(dev)).cpu(), torch.relu(x + bias.view(bshape)))
    assert torch.equal(native.bias_relu_add_(x.clone().to(dev), None, res.to(dev), relu=False).cpu(), x + res)


# ---- K6 filtfilt -------------------------------------------------------------------------------------
# 476 / 477: one chunk exactly / a second chunk of one sample; 4572 / 5000: the first chunk whose state sum is truncated at 4096 terms
@pytest.mark.parametrize("n", [4000, 480_000, 19, 257, 476, 477, 4572, 5000, 100_003])
def test_filtfilt_matches_scipy(native, dev, n):
    from scipy import signal
    from rvc_amd.lib import synthetic as S
    b, a = signal.butter(N=5, Wn=48, btype="high", fs=16000)
    x = S.synth_audio(n, seed=n % 97)
    ref = signal.filtfilt(b, a, x)
    y = native.filtfilt_order5(torch.from_numpy(x).to(dev), b, a).cpu().numpy()
    assert y.shape == ref.shape
    # the direct-form recurrence is ill-conditioned: the reference's own lfilter restarted 8192 samples earlier moves by
    # 2e-8 (csrc/filtfilt.hip); short inputs (single chunk, exact start state) reproduce SciPy's op sequence exactly
    assert np.abs(y - ref).max() <= (2e-7 if n > 512 - 36 else 0.0), np.abs(y - ref).max()   # single chunk: bit-exact
    if n == 4000:
        g = load_golden("filtfilt")
        yg = native.filtfilt_order5(torch.from_numpy(g["x"]).to(dev), g["bh"], g["ah"]).cpu().numpy()
        assert np.abs(yg - g["y"]).max() <= 2e-7


# ---- BASELINE cfg 4 / cfg 5 shapes ---------------------------------------------------------------------
@pytest.mark.parametrize("voc", ["MRF HiFi-GAN", "HiFi-GAN"])
def test_cfg4_bf16_weight_storage(native, dev, ref_inputs, voc):
    """cfg 4: vocoder with bf16 weights.  The folded weights are rounded to bf16 (what a bf16 copy holds); the handle keeps
    the ResBlock / MRF-layer conv weights as bf16 in HBM (weight_storage="bf16": conv_mfma_kernel<..., WB16>) and must
    reproduce, to fp32 accumulation noise, (a) the oracle's fp32 math on the SAME bf16-valued weights and
    (b) the fp32-storage handle fed with the same values -- bit for bit, since widening bf16 -> fp32 is exact."""
    from oracle import rvc_oracle as O
    from rvc_amd.lib import synthetic as S
    T = 48
    cpt = S.make_synth_checkpoint(48000, voc, seed=0)
    w = {k: (v.float().bfloat16().float() if k.startswith("dec.") else v) for k, v in O.fold_weight_norm(cpt["weight"]).items()}
    rates, ksizes = cpt["config"][12], cpt["config"][14]
    gen = torch.Generator().manual_seed(3)
    z, g = torch.randn(1, 192, T, generator=gen), torch.randn(1, 256, 1, generator=gen)
    f0 = _decoder_inputs(ref_inputs, T)
    mrf = voc.startswith("MRF")
    dim = 9 if mrf else 1
    src_rand, src_randn = torch.rand(1, dim, generator=gen), torch.randn(1, T * 480, dim, generator=gen)
    if mrf:
        ref = O.decoder_mrf(w, z, f0, g, rates, ksizes, 48000, O.ListNoise([src_rand.clone(), src_randn])).numpy()
    else:
        ref = O.decoder_nsf(w, z, f0, g, rates, ksizes, 48000, O.ListNoise([torch.zeros(1, 1, 1), src_randn])).numpy()
    folded = {k[4:]: v for k, v in w.items() if k.startswith("dec.")}
    outs = {}
    for storage in ("bf16", "f32"):
        dec = native.Decoder(voc, 48000, folded, weight_storage=storage)
        outs[storage] = dec.forward(z.to(dev), f0.to(dev), g[:, :, 0].to(dev), src_randn=src_randn.to(dev),
                                    src_rand=src_rand.to(dev)).cpu().numpy()
    assert rms(outs["bf16"] - ref) <= 5e-5, rms(outs["bf16"] - ref)
    # same values, same fp32 arithmetic; the C = 32 stage takes the unfused conv pair instead of the fused layer kernel
    assert rms(outs["bf16"] - outs["f32"]) <= 2e-6, rms(outs["bf16"] - outs["f32"])
    with pytest.raises(native.NativeError):
        native.Decoder("RefineGAN", 48000, {}, weight_storage="bf16")


def test_cfg5_two_million_row_index(native, dev):
    """cfg 5: 2 M x 768 index (6.1 GB) searched brute-force in HBM.  Checked against a float64 torch search on the device
    for a sample of queries, and against planted neighbours."""
    n_rows, n_q = 2_000_000, 96
    g = torch.Generator(device=dev).manual_seed(0)
    centres = torch.randn(4096, 768, device=dev, generator=g) * 0.35
    index = torch.empty(n_rows, 768, device=dev)
    for s in range(0, n_rows, 250_000):
        which = torch.randint(0, 4096, (250_000,), device=dev, generator=g)
        index[s:s + 250_000] = centres[which] + 0.05 * torch.randn(250_000, 768, device=dev, generator=g)
    planted = torch.randint(0, n_rows, (n_q,), device=dev, generator=g)
    q = index[planted] + 0.01 * torch.randn(n_q, 768, device=dev, generator=g)
    norms = native.knn_index_norms(index)
    d2, ids = native.knn_search(index, norms, q)
    assert (ids[:, 0] == planted).all()                      # the planted row is the nearest neighbour
    assert (d2[:, 1:] >= d2[:, :-1]).all()
    # exact float64 reference for the first 16 queries
    q64 = q[:16].double()
    best_d = torch.full((16, 8), float("inf"), dtype=torch.float64, device=dev)
    best_i = torch.zeros((16, 8), dtype=torch.int64, device=dev)
    for s in range(0, n_rows, 250_000):
        x = index[s:s + 250_000].double()
        d = (q64 * q64).sum(1)[:, None] - 2 * q64 @ x.T + (x * x).sum(1)[None, :]
        cd = torch.cat([best_d, d], 1)
        ci = torch.cat([best_i, torch.arange(s, s + x.shape[0], device=dev).expand(16, -1)], 1)
        o = torch.argsort(cd, dim=1, stable=True)[:, :8]
        best_d, best_i = torch.gather(cd, 1, o), torch.gather(ci, 1, o)
    agree = (ids[:16] == best_i).float().mean().item()
    assert agree >= 0.97, agree                              # near-ties may swap (see test_knn_ids_and_distances)
    assert torch.allclose(d2[:16].double(), best_d, rtol=1e-3, atol=1e-3)
