#!/usr/bin/env python3
"""A/B on ONE box (boxes differ by several per cent): child processes alternate between two settings, each timing the winobf2 conv
at C / L / K / D from the environment (default 128 / 383 760 / 11 / 1), with residual.
  AB=<n>        the ablation library with RVC_W2_DBG=0 against RVC_W2_DBG=<n> (default 64)
  AB_LIB=<.so>  the product library against another build of it (e.g. the previous commit's)
  AB_ENV=A=B    the ablation library without and with that environment setting (e.g. RVC_WBF_V3=1)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
    import torch
    from rvc_amd import _native
    dev = "cuda:0"
    C, L, K, D = int(os.environ.get("C", 128)), int(os.environ.get("L", 383760)), int(os.environ.get("K", 11)), int(os.environ.get("D", 1))
    x = torch.randn(1, C, L, device=dev); r = torch.randn(1, C, L, device=dev); b = torch.zeros(C, device=dev); y = torch.empty_like(x)
    u = _native.conv1d_winobf_pack_weight(torch.randn(C, C, K) * 0.03, dev)
    f = lambda: _native.conv1d_winobf_forward(x, u, b, C, K, D, 0.1, res=r, out=y)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for _ in range(7):
        e0.record()
        for _ in range(12): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 12 * 1e3)
    print(f"{sorted(out)[3]:.1f}")
    sys.exit(0)
libdir = os.path.join(ROOT, "codename-rvc-fork-3_amd", "rvc_amd", "_lib")
if os.environ.get("AB_LIB"):
    names = ("product", os.environ["AB_LIB"])
    envs = (dict(os.environ), dict(os.environ, RVC_AMD_LIB=os.path.join(libdir, os.environ["AB_LIB"])))
elif os.environ.get("AB_ENV"):
    k_, v_ = os.environ["AB_ENV"].split("=", 1)
    env = dict(os.environ, RVC_AMD_LIB=os.path.join(libdir, "librvc_amd_ablate.so"))
    names = ("default", os.environ["AB_ENV"])
    envs = (env, dict(env, **dict(kv.split("=", 1) for kv in os.environ["AB_ENV"].split(","))))
else:
    ab = os.environ.get("AB", "64")
    env = dict(os.environ, RVC_AMD_LIB=os.path.join(libdir, "librvc_amd_ablate.so"))
    names = ("RVC_W2_DBG=0", "RVC_W2_DBG=" + ab)
    envs = (dict(env, RVC_W2_DBG="0"), dict(env, RVC_W2_DBG=ab))
for k in os.environ.get("KS", os.environ.get("K", "11")).split(","):
    res = ([], [])
    for rep in range(4):
        for i in (0, 1):
            o = subprocess.run([sys.executable, __file__, "child"], env=dict(envs[i], K=k), capture_output=True, text=True)
            res[i].append(float(o.stdout.strip().splitlines()[-1]))
    print(f"C={os.environ.get('C', 128)} K={k} d={os.environ.get('D', 1)}: {names[0]} {res[0]} us | {names[1]} {res[1]} us", flush=True)
