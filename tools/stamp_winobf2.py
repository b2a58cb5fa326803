#!/usr/bin/env python3
"""Per-phase cycle breakdown of winobf2_conv_kernel<11,128> (K=3 in the environment: <3,128>) (ablation build, RVC_W2_DBG=128: wave 0 and the loader wave stamp
s_memtime at every barrier).  C = 128, K = 11, 383 760 columns, d from the environment (D, default 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
os.environ.setdefault("RVC_AMD_LIB", os.path.join(ROOT, "codename-rvc-fork-3_amd", "rvc_amd", "_lib", "librvc_amd_ablate.so"))
os.environ["RVC_W2_DBG"] = "128"
import numpy as np, torch
from rvc_amd import _native
dev = "cuda:0"
C, K, L, D = int(os.environ.get("C", 128)), int(os.environ.get("K", 11)), int(os.environ.get("L", 383760)), int(os.environ.get("D", 1))
x = torch.randn(1, C, L, device=dev); res = torch.randn(1, C, L, device=dev); bias = torch.zeros(C, device=dev)
u = _native.conv1d_winobf_pack_weight(torch.randn(C, C, K) * 0.03, dev)
stamps = torch.zeros(1, C, L, device=dev)          # the kernel writes [block][2][32] uint64 into it
for _ in range(3):
    _native.conv1d_winobf_forward(x, u, bias, C, K, D, 0.1, res=res, acc=stamps)
stamps.zero_()
_native.conv1d_winobf_forward(x, u, bias, C, K, D, 0.1, res=res, acc=stamps)
torch.cuda.synchronize()
raw = stamps.view(-1).view(torch.int64).cpu().numpy()
n_chunks = C // 16
n_blocks = -(-(-(-L // (4 * D)) // (64 // D)) // 8) * 8
s = raw[: n_blocks * 64].reshape(n_blocks, 2, 32).astype(np.float64)
w0, ld = s[:, 0], s[:, 1]
ok = w0[:, 0] > 0
w0, ld = w0[ok], ld[ok]
print(f"{ok.sum()} blocks; cycles (s_memtime ticks), mean over blocks; d = {D}")
def m(a): return float(np.mean(a))
print(f"  start -> P1 (chunk 0 staged by the loader)      {m(w0[:, 1] - w0[:, 0]):8.0f}")
print(f"  P1 -> P2 (transform of chunk 0 / staging chunk 1) {m(w0[:, 2] - w0[:, 1]):8.0f}   (loader arrives {m(ld[:, 2] - w0[:, 1]):.0f} after P1)")
ph = [w0[:, 3 + c] - (w0[:, 2] if c == 0 else w0[:, 2 + c]) for c in range(n_chunks)]
print("  phases (wave 0, own work until it reaches the barrier): " + " ".join(f"{m(p):.0f}" for p in ph))
# barrier release time ~ next stamp's predecessor: use the loader's idle = (compute arrival) - (loader arrival)
idle = [w0[:, 3 + c] - ld[:, 3 + c] for c in range(n_chunks)]
print("  loader waits for wave 0 at each chunk barrier:           " + " ".join(f"{m(i):.0f}" for i in idle))
print(f"  K loop total (P2 -> wave 0 done with the last chunk)      {m(w0[:, 2 + n_chunks] - w0[:, 2]):8.0f}")
e = w0[:, 20:30]
print(f"  last chunk done -> epilogue start                         {m(e[:, 0] - w0[:, 2 + n_chunks]):8.0f}")
le = ld[:, 20:30]
print(f"  residual loads issued                                     {m(e[:, 1] - e[:, 0]):8.0f}   (loader {m(le[:, 1] - e[:, 0]):.0f} after wave 0's epilogue start)")
print(f"  pass 0: accumulators written to LDS                       {m(e[:, 2] - e[:, 1]):8.0f}   (loader at the barrier {m(le[:, 2] - e[:, 0]):.0f})")
print(f"  pass 0: barrier                                           {m(e[:, 3] - e[:, 2]):8.0f}")
print(f"  pass 0: reduced into registers                            {m(e[:, 4] - e[:, 3]):8.0f}")
print(f"  pass 1: written                                           {m(e[:, 5] - e[:, 4]):8.0f}")
print(f"  pass 1: barrier                                           {m(e[:, 6] - e[:, 5]):8.0f}")
print(f"  pass 1: reduced into registers                            {m(e[:, 7] - e[:, 6]):8.0f}")
print(f"  residual added, stores issued                             {m(e[:, 9] - e[:, 7]):8.0f}")
print(f"  whole block (first to last stamp)                         {m(e[:, 9] - w0[:, 0]):8.0f}")
