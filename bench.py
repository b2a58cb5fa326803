#!/usr/bin/env python3
"""bench.py -- end-to-end voice-conversion throughput of the MI355X path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4,5}] [--inflight M]

One "step" = one pass of the hot path over one synthetic utterance: `Pipeline.pipeline` from the 16 kHz input array to
the float32 waveform at the model's rate.  `--config` picks the BASELINE.json configuration (default 2, the one the
metric is quoted on); config 3 is config 2's utterance as a 512-utterance batch: `--steps` defaults to 512 / N per rank, so
`python bench.py --config 3 --gpus 8` is BASELINE cfg 3 literally:
  1  10 s clip, HuBERT-base + v2 40k NSF-HiFi-GAN, index_rate 0
  2  30 s clip, HuBERT-base + NSF-HiFi-GAN 48k, 100 000 x 768 index, index_rate 0.75
  4  30 s clip, MRF-HiFi-GAN 48k with bf16 weights, 100 000 x 768 index, index_rate 0.75
  5  30 s clip, RefineGAN 48k, 2 000 000 x 768 index (6.1 GB, brute-force L2 in HBM), noise drawn on the device
Weights are seeded random-init (no network), inputs synthetic.

N > 1: one rank per GPU.  Started either by `python -m torch.distributed.run ... bench.py --gpus N ...` (RANK / WORLD_SIZE
already in the environment) or directly as `python bench.py --gpus N ...`, in which case this process -- before it touches
the GPU -- starts the N ranks itself (rvc_amd.infer.distributed.spawn_ranks; the reference's multi-GPU precedent,
rvc/train/extract/extract.py:141-152, starts its per-device workers the same way) and relays rank 0's line.  A world size
that differs from --gpus, or fewer visible GPUs than --gpus, is an error, never a silent 1-GPU run.  Every rank converts
its own utterances (utterance i -> rank i mod N); the index is built on rank 0 and replicated with ONE RCCL broadcast
through the C ABI (rvc_index_broadcast) and verified by a device-side checksum; the steady state has no collective
("scaling": "weak").  On each GPU `--inflight` utterances (default 3) are in flight at a time, each on its own HIP stream.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel symbol (the 11-tap ResBlock convs of vocoder stages 0-2: Winograd F(4,4) on the bf16 matrix
                cores with fp32 operands split exactly into three bf16), timed live with HIP events on the launch stream
  roofline_knn  the L2 top-8 search at this config's (queries x rows): the fp16-screened regime, `frac` = its fp16 MFMA fraction
                (what bounds it); physical HBM bytes and SURVEY §8d's formula figure ride along as hbm_frac_physical / frac_8d
  roofline_knn_stream  the same search for ONE 32-query tile (knn_direct_kernel: one pass over the fp32 rows) -- the HBM-bound
                regime, `frac` = N x 3072 B / time / 8 TB/s, `traffic` from profiles/pmc_knn_stream_<N>.json
  host_io       the same K steps with host NumPy in / host float32 out (PCIe inclusive) at the same `inflight`
  cpu_baseline  the oracle (CPU restatement of the reference, oracle/rvc_oracle.py) on the host cores, on the config's own
                utterance and index: warm-up + median of 3 (rank 0, N = 1 only; --cpu-seconds bounds the sample)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = FP32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA
PEAK_HBM_GBS = 8000.0           # HBM3E spec


def pmc_traffic(symbol_key, field="bytes_per_launch"):
    """Average HBM bytes per launch of the roofline kernel symbol at the cfg-2 shape, from the rocprofv3 PMC passes committed
    under profiles/ (they cannot be collected inside bench.py: that needs rocprofv3).  The figure lives in a JSON file next to
    the raw CSVs, written by tools/summarize_pmc.py -- not in this script."""
    path = os.path.join(ROOT, "profiles", f"pmc_{symbol_key}.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        return float(d[field]), f"profiles/pmc_{symbol_key}.json: " + d.get("source", "")
    except (OSError, ValueError, KeyError) as e:
        print(f"bench.py: no PMC traffic figure for '{symbol_key}' ({path}: {type(e).__name__}); `traffic` is null -- "
              "tools/collect_profiles_r05.sh regenerates it", file=sys.stderr)
        return None, None


CONFIGS = {
    1: dict(seconds=10.0, sr=40000, vocoder="HiFi-GAN", index_rows=0, index_rate=0.0, weights="f32",
            name="BASELINE cfg 1: 10 s 16 kHz clip, HuBERT-base + v2 40k NSF-HiFi-GAN, index_rate 0"),
    2: dict(seconds=30.0, sr=48000, vocoder="HiFi-GAN", index_rows=100_000, index_rate=0.75, weights="f32",
            name="BASELINE cfg 2: 30 s 16 kHz clip -> 48 kHz, HuBERT-base + NSF-HiFi-GAN 48k, 100000x768 index, index_rate 0.75"),
    3: dict(seconds=30.0, sr=48000, vocoder="HiFi-GAN", index_rows=100_000, index_rate=0.75, weights="f32", batch=512,
            name="BASELINE cfg 3: 512 x 30 s 48 kHz utterance batch sharded over the ranks (512 / N per GPU), HuBERT-base + "
                 "NSF-HiFi-GAN 48k, 100000x768 index replicated by one RCCL broadcast, index_rate 0.75"),
    4: dict(seconds=30.0, sr=48000, vocoder="MRF HiFi-GAN", index_rows=100_000, index_rate=0.75, weights="bf16",
            name="BASELINE cfg 4: 30 s clip, MRF-HiFi-GAN 48k, vocoder weights stored as bf16 in HBM, 100000x768 index, index_rate 0.75"),
    5: dict(seconds=30.0, sr=48000, vocoder="RefineGAN", index_rows=2_000_000, index_rate=0.75, weights="f32",
            name="BASELINE cfg 5 (one GPU's share): 30 s clip, RefineGAN 48k, 2000000x768 index brute-force L2 in HBM, "
                 "index_rate 0.75, noise drawn on the device"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed utterances per GPU (default 20; config 3: 512 / N)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--seconds", type=float, default=None, help="override the clip length of the config")
    ap.add_argument("--index-rows", type=int, default=None, help="override the index size of the config")
    ap.add_argument("--inflight", type=int, default=3, help="utterances in flight per GPU, each on its own HIP stream (round 6: 3 -- "
                    "25.1-25.7 ms per cfg-2 utterance against 26.0-26.2 with 2 on the same box, profiles/r06_inflight_sweep.txt; rounds 2-5: 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vocoder-side-streams", type=int, default=None,
                    help="A/B: side streams the vocoder spreads a short stage's ResBlock branches over (default 0 = every launch on the utterance's stream; -1 = one per branch)")
    ap.add_argument("--no-rooflines", action="store_true", help="skip the per-kernel roofline legs (timed region only)")
    ap.add_argument("--cpu-seconds", type=float, default=None,
                    help="clip length of the CPU-baseline sample (default: the config's own clip length, i.e. the same "
                         "synthetic input as the GPU run; e.g. 3 for a quick bounded sample)")
    ap.add_argument("--cpu-threads", type=int, default=None, help="torch threads of the CPU baseline (default min(host cpus, 32))")
    ap.add_argument("--control-flow-only", action="store_true",
                    help="CPU test hook: run the launcher / process-group / sharding / broadcast / report path over gloo "
                         "without any kernel (no throughput is measured; the line says so)")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = max(1, CONFIGS[args.config].get("batch", 20 * max(1, args.gpus)) // max(1, args.gpus))
    return args


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ---- roofline legs ----------------------------------------------------------------------------------------------------
def roofline_mix(torch, native, dev, T, rates, weights, k=11):
    """Every launch of the dominant kernel symbol in one utterance's vocoder forward: rvc::winobf2_conv_kernel<11,128,0>
    (winobf2.hip: Winograd F(4,4) on the bf16 matrix cores, fp32 operands split exactly into three bf16, one transform point per
    wave), i.e. the 11-tap ResBlock convs of vocoder stages 0 and 1 (C = 256, 128; the C = 64 stage runs winobf.hip's 64 x 128
    blocks, the C = 32 stage the fp32 kernels), per stage and for each dilation d: conv1 (dilation d) then conv2 (dilation 1,
    + residual; the last one also + running sum, x 1/3) -- 12 launches, the population rocprofv3 --stats averages over for that
    symbol.  `weights` = "bf16" (BASELINE cfg 4): the decoder handle with bf16 weight storage runs THESE layers on K3d
    (rvc::convbf1_kernel<11, C>, convbf1.hip: direct form, one-term taps, three bf16 products per multiply-add) -- the same 12
    launches are timed on that kernel and `executed` counts its products.
    Returns (callable, algorithmic flops per call, launches per call, algorithmic HBM bytes per call, executed flops)."""
    gen = torch.Generator().manual_seed(1)
    stages, flops, executed, alg_bytes = [], 0.0, 0.0, 0.0
    points = 7 * ((k + 3) // 4)     # multiply-adds executed per 4 outputs and (c_in, c_out) pair: 7 points per group of four taps
    per_mac = 6                     # bf16 products per fp32 multiply-add of the split form
    L = T
    for i in range(2):
        C, L = 512 >> (i + 1), L * rates[i]
        x = torch.randn(1, C, L, device=dev)
        w1, w2 = torch.randn(C, C, k, generator=gen) * 0.02, torch.randn(C, C, k, generator=gen) * 0.02
        if weights == "bf16":
            w1, w2 = w1.bfloat16().float(), w2.bfloat16().float()
        pack = native.conv1d_bf16w_pack_weight if weights == "bf16" else native.conv1d_winobf_pack_weight
        st = dict(C=C, x=x, t1=torch.empty_like(x), y=torch.randn(1, C, L, device=dev), acc=torch.randn(1, C, L, device=dev),
                  w1=pack(w1, dev), w2=pack(w2, dev), bias=torch.zeros(C, device=dev))
        stages.append(st)
        flops += 6 * 2.0 * C * C * k * L                          # 2 x MACs of the conv as the reference computes it
        # bf16 storage: the same layers run K3d (convbf1.hip) -- direct form, every multiply-add, THREE bf16 products each
        executed += 6 * 2.0 * C * C * k * L * 3 if weights == "bf16" else 6 * 2.0 * C * C * points * (L / 4.0) * per_mac
        tensor = C * L * 4.0
        alg_bytes += 3 * (2 * tensor) + 2 * (3 * tensor) + 1 * (4 * tensor)   # conv1: r+w; conv2: r+res+w (+acc)
    if weights == "bf16":
        def fwd(x, u, bias, C, k_, d, slope, **kw):
            return native.conv1d_bf16w_forward(x, u, bias, k_, d, slope, **kw)
    else:
        fwd = native.conv1d_winobf_forward

    def run():
        for st in stages:
            C = st["C"]
            for j, d in enumerate((1, 3, 5)):
                fwd(st["x"], st["w1"], st["bias"], C, k, d, 0.1, out=st["t1"])
                if j < 2:
                    fwd(st["t1"], st["w2"], st["bias"], C, k, 1, 0.1, res=st["x"], out=st["y"])
                else:
                    fwd(st["t1"], st["w2"], st["bias"], C, k, 1, 0.1, res=st["x"], acc=st["acc"], out_scale=1 / 3, out=st["y"])
    return run, flops, 12, alg_bytes, executed


def decoder_flops(T, rates, ksizes, c0=512, cin=192, res_k=(3, 7, 11), n_dil=3):
    """Closed form of SURVEY §8d (2 x MACs), NSF / MRF topology."""
    macs = cin * c0 * 7 * T
    length, ch = T, c0
    for i, (u, k) in enumerate(zip(rates, ksizes)):
        co = ch // 2
        lout = length * u
        stride_f0 = int(np.prod(rates[i + 1:])) if i + 1 < len(rates) else 1
        k_nc = 1 if stride_f0 == 1 else stride_f0 * 2 - stride_f0 % 2
        macs += ch * co * k * length + co * k_nc * lout + 2 * n_dil * sum(res_k) * co * co * lout
        length, ch = lout, co
    macs += ch * 7 * length
    return 2.0 * macs


def synth_index_device(torch, n_rows, dev, seed=0, n_centres=512, jitter=0.05, dim=768):
    """rvc_amd.lib.synthetic.synth_index's recipe (cluster centres + jitter) drawn on the device: a 2 M-row index is
    6.1 GB, too slow to draw with NumPy inside a bench run."""
    g = torch.Generator(device=dev).manual_seed(seed)
    centres = torch.randn(n_centres, dim, device=dev, generator=g) * 0.35
    out = torch.empty(n_rows, dim, device=dev)
    for s in range(0, n_rows, 1 << 18):
        e = min(n_rows, s + (1 << 18))
        which = torch.randint(0, n_centres, (e - s,), device=dev, generator=g)
        out[s:e] = centres[which] + jitter * torch.randn(e - s, dim, device=dev, generator=g)
    return out


def main():
    args = parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.seconds is not None:
        cfg["seconds"] = args.seconds
    if args.index_rows is not None:
        cfg["index_rows"] = args.index_rows
    want = max(1, args.gpus)

    # ---- who am I: a rank started by a launcher, or the process that has to start the ranks ----
    if "WORLD_SIZE" not in os.environ and want > 1:
        from rvc_amd.infer import distributed as D   # the parent never opens the GPU driver: no torch.cuda call here
        if not args.control_flow_only:
            n_dev = D.count_gpus_sysfs()              # KFD topology in sysfs; None when it cannot be read (the ranks check)
            if n_dev is not None and n_dev < want:
                print(f"bench.py: --gpus {want} but only {n_dev} GPU(s) are visible; refusing to report a {want}-GPU "
                      "number from fewer devices", file=sys.stderr)
                sys.exit(2)
        extra = {"RVC_DIST_BACKEND": "gloo"} if args.control_flow_only else {}
        sys.exit(D.spawn_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], want, env_extra=extra))

    import torch
    from rvc_amd.infer import distributed as D
    rank, world, local = D.init_process_group()
    if world != want:
        print(f"bench.py: --gpus {want} but WORLD_SIZE is {world}; start one rank per GPU (or drop the launcher and let "
              "bench.py start them)", file=sys.stderr)
        sys.exit(2)
    if args.control_flow_only:
        return control_flow_only(torch, D, args, cfg, rank, world)

    n_dev = torch.cuda.device_count()
    if world > max(n_dev, 1) and os.environ.get("RVC_DIST_BACKEND") != "gloo" or (world > 1 and n_dev == 0):
        if rank == 0:
            print(f"bench.py: --gpus {want} but only {n_dev} GPU(s) are visible; refusing to report a {want}-GPU number from "
                  "fewer devices", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; there is no CPU fallback"
    local = local % n_dev   # ranks > devices only with RVC_DIST_BACKEND=gloo (several ranks share a GPU: control-flow runs)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    gloo_only = os.environ.get("RVC_DIST_BACKEND") == "gloo"
    if world > 1:   # N host processes share the node's cores: keep each rank's CPU thread pool small
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))

    def barrier():
        if world > 1:
            torch.distributed.barrier(**({} if gloo_only else {"device_ids": [local]}))

    from rvc_amd import _native
    from rvc_amd.infer.infer import VoiceConverter
    from rvc_amd.lib import synthetic as S

    # ---- load (untimed): weights, index broadcast ----
    sr = cfg["sr"]
    cpt = S.make_synth_checkpoint(sr, cfg["vocoder"], seed=0)
    vc = VoiceConverter(device=dev)
    if cfg["weights"] == "bf16":
        vc.dec_weight_dtype = "bf16"
    vc.load_checkpoint_dict(cpt)
    vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
    vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
    vc.branch_streams = args.vocoder_side_streams or 0
    index_dev, bcast = None, {}
    if cfg["index_rows"] > 0:
        big = None
        if rank == 0:
            big = S.synth_index(cfg["index_rows"], seed=0) if cfg["index_rows"] <= 200_000 else \
                synth_index_device(torch, cfg["index_rows"], dev, seed=0)
        index_dev = D.broadcast_index(big, dev, force_rccl=not gloo_only)
        bcast = D.last_broadcast_info()
        assert D.checksums_agree(index_dev), "feature index differs across ranks after the broadcast"
        vc.vc.set_index(index_dev)
        del big

    n_in = int(round(cfg["seconds"] * 16000))
    n_total = args.steps + args.warmup
    # utterance i (global) uses rng seed i; this rank converts i = rank, rank + world, ...
    audios_host = [S.synth_audio(n_in, seed=rank + world * j) for j in range(min(n_total, 4))]
    audios = [torch.from_numpy(a).to(dev) for a in audios_host]   # inputs resident in HBM before the timed region

    inflight = max(1, args.inflight)
    kw = dict(index_path="", index_rate=cfg["index_rate"], protect=0.5, sid=0)

    def run_steps(pool, first, count):
        """`count` utterances, `inflight` at a time (VoiceConverter.convert_batch; 1 = strictly one after the other)."""
        return vc.convert_batch([pool[(first + j) % len(pool)] for j in range(count)], inflight=inflight, **kw)

    run_steps(audios, 0, args.warmup * inflight)   # every stream warms up its own workspaces / side stream
    # Two one-off stalls that a short timed region must not contain by accident (seen as 10-50 ms in about one fresh process of three:
    # 28.5 against 26.0 ms per step for the same tree on one box, profiles/r06_bench_cfg2.json vs _again): a device allocation in the
    # middle of the run -- the host threads of the utterances in flight interleave differently every time, and an interleaving that
    # needs one more cached block than warm-up left behind sends torch's allocator to hipMalloc -- and a full Python garbage collection.
    # Warm-up therefore ends with one batch at `inflight + 1` (peak demand above the steady state's: the pool keeps the slack), and the
    # interpreter's long-lived objects are frozen out of the collector's generations.  Both are reported in the line (`timed_region`).
    # (A pool reserved up front -- one big cached segment -- changes nothing: tools/bench_pool_ab.sh, 27-33 allocations either way.)
    vc.convert_batch([audios[j % len(audios)] for j in range(inflight + 1)], inflight=inflight + 1, **kw)
    # ... and the allocator is given time to SETTLE: torch's caching allocator keeps growing its per-stream pools through the first ~50
    # utterances of a process (tools/diag_alloc.py: 238 hipMalloc calls in the first 20 utterances, then 5, 1, 0, 0, 1, 2, 0 ... per 20),
    # i.e. well into a 20-step region that starts after 5 warm-up steps.  Untimed batches of `inflight` utterances run until one of them
    # needs no device allocation (at most 16 batches, ~25 ms per utterance); `timed_region.settle_utterances` says how many it took.
    settle = 0
    for _ in range(16):
        n0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        run_steps(audios, settle, inflight)
        settle += inflight
        if torch.cuda.memory_stats(dev).get("num_device_alloc", 0) == n0:
            break
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    torch.cuda.synchronize()
    mem0, gc0 = torch.cuda.memory_stats(dev), [g["collections"] for g in gc.get_stats()]
    t0 = time.perf_counter()
    outs = run_steps(audios, args.warmup, args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    mem1, gc1 = torch.cuda.memory_stats(dev), [g["collections"] for g in gc.get_stats()]
    timed_region = {"device_allocations": int(mem1.get("num_device_alloc", 0) - mem0.get("num_device_alloc", 0)),
                    "allocator_retries": int(mem1.get("num_alloc_retries", 0) - mem0.get("num_alloc_retries", 0)),
                    "reserved_bytes": int(mem1.get("reserved_bytes.all.current", 0)),
                    "gc_collections": [b - a for a, b in zip(gc0, gc1)], "settle_utterances": settle}
    samples = sum(int(o.shape[0]) for o in outs)
    total_samples, t_max = D.reduce_report(samples, elapsed, dev)
    rank_seconds = D.gather_seconds(elapsed, dev)
    # outside the timed region: EVERY timed waveform must be finite and inside [-1, 1] (a silent NaN utterance must not
    # count as throughput)
    for o in outs:
        assert bool(torch.isfinite(o).all()) and float(o.abs().max()) <= 1.0, "bench produced a non-finite or unnormalised waveform"
    out_len = int(outs[-1].shape[0])
    del outs

    # the boundary as the reference has it: host NumPy in, host float32 NumPy out (PCIe inclusive), same schedule
    run_steps(audios_host, 0, 2 * inflight)   # every stream touches both of its page-locked slots once
    torch.cuda.synchronize()
    t_host = None
    for _ in range(2):                        # best of two passes of K steps: a one-off host stall must not pose as PCIe cost
        t0 = time.perf_counter()
        houts = run_steps(audios_host, args.warmup, args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t_host = dt if t_host is None else min(t_host, dt)
    assert all(isinstance(o, np.ndarray) and o.dtype == np.float32 and np.isfinite(o).all() for o in houts)
    host_io = {"samples_per_s_rank0": round(sum(o.shape[0] for o in houts) / t_host, 1),
               "ms_per_step": round(t_host / args.steps * 1e3, 2), "inflight": inflight,
               "what": "same K steps (best of two passes), 16 kHz float64 NumPy array in host memory -> float32 NumPy waveform in host "
                       "memory (the reference's Pipeline.pipeline boundary, pipeline.py:509-528)"}
    del houts

    if rank != 0:
        barrier()
        D.destroy_native_comm()
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    value = total_samples / t_max
    line = {
        "metric": "48 kHz audio samples/sec end-to-end VC" if sr == 48000 else f"{sr // 1000} kHz audio samples/sec end-to-end VC",
        "value": round(value, 1),
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(t_max / args.steps * 1e3, 2),
        "ms_per_step_ranks": {"min": round(min(rank_seconds) / args.steps * 1e3, 2), "max": round(max(rank_seconds) / args.steps * 1e3, 2),
                              "all": [round(t / args.steps * 1e3, 2) for t in rank_seconds]},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f32 activations x bf16-stored vocoder taps (every ResBlock conv in direct form with ONE-TERM taps -- a bf16-valued tap is its own first split, "
                  "three bf16 products per multiply-add against exact bf16x3 activations: fused pairs up to 128 channels x 7 taps, single convs at 256 channels "
                  "and at 128 x 11 taps; fp32 accumulate everywhere)")
                 if cfg["weights"] == "bf16" else
                 "f32 (vocoder ResBlock convs and upsamplers, HuBERT, RMVPE's U-Net convs: fp32 operands as exact bf16x3 splits on the bf16 matrix cores -- six "
                 "products of order <= 2^-16 per multiply-add, fp32 accumulate; everything else fp32 / fp64)",
        "data": "synthetic (seeded random-init weights, FM-tone utterances, clustered index)",
        "rtf": round(value / sr, 2),
        "headline_is": "inputs resident in HBM, waveform left in HBM (the bench contract's definition of `value`); `host_io` is the "
                       "same run through the reference's host-array boundary and is the figure to quote for a drop-in user",
        "config": {"workload": f"{cfg['name']}; rmvpe, protect 0.5; 1 utterance per step per GPU, {inflight} utterance(s) in "
                               f"flight per GPU on separate HIP streams" + (f"; clip {cfg['seconds']:g} s" if args.seconds else ""),
                   "baseline_config": args.config,
                   "samples_per_step": out_len, "parallelism": f"utterance-sharded x{world}", "inflight": inflight,
                   "input_residency": "16 kHz float64 utterances resident in HBM before the timed region; waveform left in HBM "
                                      "(host_io carries the PCIe-inclusive figure at the same inflight)",
                   "noise": "drawn on the device (torch Philox)",
                   "vocoder_weights": cfg["weights"]},
        "rccl_ranks": bcast.get("n_ranks"),
        "index_broadcast": None if not bcast else {
            "transport": bcast.get("transport"), "bytes": bcast.get("bytes"), "seconds": round(bcast.get("seconds", 0.0), 4),
            "gbps": round(bcast.get("bytes", 0) / max(bcast.get("seconds", 0.0), 1e-9) / 1e9, 2) if world > 1 else None,
            "rccl_version": bcast.get("rccl_version"), "library": bcast.get("library"),
            "verified": "device checksum (rvc_checksum64) equal on every rank"},
        "host_io": host_io,
        "timed_region": timed_region,   # hipMalloc calls / Python collections INSIDE the timed region (both should be zero / young generations only)
        # what the host side of each rank looks like: convert_batch drives `inflight` Python threads per rank; at N = 8 the
        # driver's scaling curve can be read against this (host-bound ranks show as equal GPU idle on every rank)
        "host": {"cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                 "torch_threads": torch.get_num_threads(), "python_threads_per_rank": inflight + 1, "ranks": world},
    }

    if not args.no_rooflines:
        line.update(rooflines(torch, _native, vc, cpt, cfg, dev, n_in, index_dev, audios[0]))

    # ---- CPU baseline: the oracle on the host cores, on the SAME synthetic utterance and index the GPU run used ----
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(torch, S, cfg, cpt, args, index_dev, sr)
    else:
        line["cpu_baseline"] = None

    # tear down first, report last: the JSON line is the LAST thing on stdout (RCCL / the runtime may print on C stdout while
    # communicators go down; distributed._c_stdout_to_stderr keeps that off this stream)
    barrier()
    with D._c_stdout_to_stderr():
        D.destroy_native_comm()
        if world > 1:
            torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


def cpu_baseline(torch, S, cfg, cpt, args, index_dev, sr):
    """oracle.pipeline (the CPU restatement of the reference, pinned to reference-generated fixtures) on the box's host cores.
    Default sample = the config's own utterance (seed 0, the first of the timed ones) and, up to 200 000 rows, its own index:
    one warm-up on a 3 s clip (thread pool, mel basis, allocator), then the median of 3 full runs -- ~2-3 min for cfg 2.
    `--cpu-seconds S` bounds the sample to an S-second clip instead.  Threads: min(host cpus, 32) unless --cpu-threads says
    otherwise -- the restatement's torch ops stop scaling (and start thrashing) far below the 256+ hardware threads of the
    GPU boxes; the line states what was used."""
    from oracle import rvc_oracle as O
    cores = args.cpu_threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    secs = cfg["seconds"] if args.cpu_seconds is None else args.cpu_seconds
    hub_sd, rm_sd = S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0)
    cpu_rows = min(cfg["index_rows"], 200_000)     # cfg 5's 2 M rows x float64 scores do not fit a bounded run
    big_host = index_dev[:cpu_rows].cpu().numpy() if cpu_rows else None
    kw = dict(sid=0, pitch=0, big_npy=big_host, index_rate=cfg["index_rate"], protect=0.5, knn_dtype=np.float32)
    torch.manual_seed(0)
    O.pipeline(hub_sd, rm_sd, cpt, S.synth_audio(48000, seed=0), **kw)          # warm-up
    a = S.synth_audio(int(round(secs * 16000)), seed=0)
    times, n_out = [], 0
    for _ in range(3):
        torch.manual_seed(0)
        t0 = time.perf_counter()
        ref = O.pipeline(hub_sd, rm_sd, cpt, a, **kw)
        times.append(time.perf_counter() - t0)
        n_out = ref.shape[0]
    t_cpu = float(np.median(times))
    return {"value": round(n_out / t_cpu, 1), "unit": "samples/s", "cores": cores, "cpu_model": cpu_model_string(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "cores_policy": "torch.set_num_threads(min(host cpus, 32)) unless --cpu-threads is given",
            "rtf": round(n_out / t_cpu / sr, 3),
            "sample": f"{secs:g} s clip (synth_audio seed 0" + (", the GPU run's first utterance" if args.cpu_seconds is None else "")
                      + f") at this config's settings ({cfg['vocoder']} {sr // 1000}k, {cpu_rows}x768 index"
                      + ("" if cpu_rows == cfg["index_rows"] else f" = the first {cpu_rows} of the GPU run's {cfg['index_rows']} rows")
                      + f", index_rate {cfg['index_rate']}), oracle.pipeline, {cores} torch threads; warm-up on a 3 s clip + median of 3",
            "seconds_median": round(t_cpu, 2), "seconds_all": [round(t, 2) for t in times]}


def rooflines(torch, _native, vc, cpt, cfg, dev, n_in, index_dev, utterance):
    """Per-kernel legs (rank 0, after the timed region): dominant conv symbol, whole vocoder, kNN."""
    res = {}
    sr = cfg["sr"]
    rates, ksizes = cpt["config"][12], cpt["config"][14]
    n_pad = n_in + 32000                              # 1 s reflect pad each side (pipeline.py:581)
    F_ = (n_pad - 400) // 320 + 1                     # HuBERT frames
    T = min(n_pad // 160, 2 * F_)                     # synth frames (pipeline.py:467)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # ---- dominant kernel symbol: the 11-tap ResBlock convs of stages 0-2 ----
    run_mix, mix_flops, mix_launches, mix_alg_bytes, mix_executed = roofline_mix(torch, _native, dev, T, rates, cfg["weights"])
    # the vocoder of this config on synthetic inputs: timed below, and run (untimed) in front of every roofline batch
    z = torch.randn(1, 192, T, device=dev)
    f0 = torch.full((1, T), 220.0, device=dev)
    gv = torch.randn(1, 256, device=dev)
    nz = vc.net_g._draw(None, 1, T)

    def dec():
        return vc.net_g.dec.forward(z, f0, gv, src_randn=nz["src_randn"].contiguous(), src_rand=nz.get("src_rand"),
                                    adain_randn=nz.get("adain_randn"))
    dec()
    for _ in range(2):
        run_mix()
    reps = 1
    t_batches = []
    # MEDIAN of nine batches of the 12 launches, each timed DIRECTLY BEHIND one vocoder forward on the same stream: the kernel runs
    # 4-5 % slower in the state the pipeline leaves the chip in (clock under the preceding launches' load) than in a long run of
    # nothing but itself, and the pipeline is where rocprofv3 --stats averages it (round 5: 321.0 us traced, 306.5 us from 60
    # back-to-back isolated launches, 311-314 us measured this way)
    for _ in range(9):
        torch.cuda.synchronize()
        dec()
        e0.record()
        for _ in range(reps):
            run_mix()
        e1.record()
        torch.cuda.synchronize()
        t_batches.append(e0.elapsed_time(e1) / (reps * mix_launches) * 1e-3)
    t_launch = float(np.median(t_batches))
    flops_launch = mix_flops / mix_launches
    cfg2 = T == 3198 and list(rates[:2]) == [12, 10]
    exe_launch = mix_executed / mix_launches
    traffic, traffic_src = pmc_traffic("winobf2_k11") if cfg2 else (None, None)
    bf16w = cfg["weights"] == "bf16"
    if bf16w:
        traffic, traffic_src = None, None      # (the PMC file is K3y's)
    res["roofline"] = {
        "kernel": ("rvc::convbf1_kernel<11,256> + <11,128> (K3d: direct form, one-term bf16 taps x exact bf16x3 activations): ALL 12 launches per utterance"
                   if bf16w else "rvc::winobf2_conv_kernel<11,128,0>: ALL 12 launches per utterance of this symbol")
                  + " -- the 11-tap ResBlock convs of vocoder "
                  f"stages 0-1 (C=256/128 at {T * rates[0]}/{T * rates[0] * rates[1]} columns) in the decoder's own mix (dilations 1/3/5, "
                  "residual on every second one); per-launch figures are averages over the 12",
        "bound": "mfma", "achieved": round(exe_launch / t_launch / 1e12, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(exe_launch / t_launch / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
        "achieved_is": "bf16 matrix flops the kernel EXECUTES / launch time, against the dense bf16 MFMA peak: the pipe's own occupancy.  "
                       + ("Direct form: every multiply-add of the conv, three bf16 products each (a bf16-valued tap is one term; the activations "
                          "keep their exact three-way split), fp32 accumulate; " if bf16w else
                          "One fp32 multiply-add of the Winograd F(4,4) form (7 * 3 / (4 * 11) = 0.477 of the conv's "
                          "multiply-adds) costs six bf16 products (fp32 operands split exactly into three bf16, fp32 accumulate); ")
                       + "the SURVEY 8d ALGORITHMIC rate (2 x MACs of the 11-tap conv / time) is algorithmic_tflops, "
                       "i.e. algorithmic_vs_fp32_mfma_peak x the 157.3 TF an fp32-matrix-instruction kernel could reach",
        "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(mix_alg_bytes / mix_launches),
        "algorithmic_flops_per_launch": flops_launch, "executed_flops_per_launch": exe_launch,
        "algorithmic_tflops": round(flops_launch / t_launch / 1e12, 2),
        "algorithmic_vs_fp32_mfma_peak": round(flops_launch / t_launch / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
        "avg_launch_ms": round(t_launch * 1e3, 4), "batch_launch_ms": [round(t * 1e3, 4) for t in t_batches],
        "launches_per_utterance": mix_launches,
        "timing": "HIP events around the 12 launches of the mix on the launch stream, directly behind one (untimed) vocoder forward, median "
                  "of 9 such batches; rocprofv3 --stats of a sequential run (--inflight 1) averages the same symbol over the pipeline's "
                  "own launches (profiles/); with two utterances in flight a kernel's traced duration also contains the time it "
                  "shares the chip"}
    del run_mix

    # ---- whole vocoder, timed with events around rvc_decoder_forward ----
    dec()
    e0.record()
    for _ in range(3):
        dec()
    e1.record()
    torch.cuda.synchronize()
    t_dec = e0.elapsed_time(e1) / 3 * 1e-3
    if cfg["vocoder"] == "RefineGAN":
        dflops = 3889.5e9 / 3198 * T   # measured with FlopCounterMode on the reference module
    else:
        dflops = decoder_flops(T, rates, ksizes)
    # SURVEY 8d's conv-as-written FLOPs over the vocoder's time: a RATE, not a roofline fraction -- the Winograd / bf16x3 form executes
    # 0.477-0.5 of those multiply-adds on another pipe, so this rate exceeds the fp32 matrix peak without any work being skipped.
    res["decoder"] = {"vocoder": cfg["vocoder"], "tflops_per_utterance": round(dflops / 1e12, 4), "ms": round(t_dec * 1e3, 2),
                      "algorithmic_tflops": round(dflops / t_dec / 1e12, 2)}
    del z, nz

    # ---- kNN at this config's shape (F_ queries of one utterance x the resident index) ----
    if index_dev is not None:
        idx = vc.vc._preset_index
        n_rows = int(index_dev.shape[0])
        # the queries of one of the timed utterances (HuBERT features of the synthetic clip), not a synthetic stand-in
        vc.vc.debug_taps = {}
        vc.convert_batch([utterance], inflight=1, index_path="", index_rate=cfg["index_rate"], protect=0.5, sid=0)
        q = vc.vc.debug_taps["knn_queries"].clone()
        vc.vc.debug_taps = None
        assert q.shape == (F_, 768), q.shape
        idx.search_device(q)
        reps = 20 if n_rows <= 200_000 else 3
        e0.record()
        for _ in range(reps):
            idx.search_device(q)
        e1.record()
        torch.cuda.synchronize()
        t_knn = e0.elapsed_time(e1) / reps * 1e-3
        ktraffic, ksrc = pmc_traffic(f"knn_{n_rows}", "bytes_per_search") if F_ == 1599 else (None, None)
        res["roofline_knn"] = _native.knn_roofline_report(n_rows, F_, 768, t_knn, PEAK_HBM_GBS, PEAK_FP32_MFMA_TFLOPS,
                                                          PEAK_F16_MFMA_TFLOPS, traffic=ktraffic, traffic_source=ksrc)
        # ---- the STREAMING regime of the same search: one query tile (32 queries: a 0.64 s clip, or one segment's tail) against the
        # resident index -- knn_direct_kernel makes ONE pass over the fp32 rows and is bound by HBM; this is the regime in which
        # north_star's ">= 60 % HBM roofline on the kNN kernel" is a physical statement (pipeline.py:497-507 with a short feats tensor)
        qs = q[:32].contiguous()
        idx.search_device(qs)
        reps = 50 if n_rows <= 200_000 else 8
        e0.record()
        for _ in range(reps):
            idx.search_device(qs)
        e1.record()
        torch.cuda.synchronize()
        t_s = e0.elapsed_time(e1) / reps * 1e-3
        straffic, ssrc = pmc_traffic(f"knn_stream_{n_rows}", "bytes_per_search")
        res["roofline_knn_stream"] = _native.knn_roofline_report(n_rows, 32, 768, t_s, PEAK_HBM_GBS, PEAK_FP32_MFMA_TFLOPS,
                                                                 PEAK_F16_MFMA_TFLOPS, traffic=straffic, traffic_source=ssrc)
    return res


def control_flow_only(torch, D, args, cfg, rank, world):
    """CPU test hook (--control-flow-only): everything around the kernels -- process group, striding, index broadcast,
    checksum, report reduction, the one-line report -- over gloo on host tensors.  No throughput is measured."""
    from rvc_amd.lib import synthetic as S
    big = S.synth_index(256, seed=0) if rank == 0 else None
    idx = D.broadcast_index(big, "cpu")
    assert D.checksums_agree(idx)
    mine = D.shard_indices(args.steps * world, rank, world)
    total, t_max = D.reduce_report(len(mine) * 1000, 1.0 + 0.001 * rank, "cpu")
    secs = D.gather_seconds(1.0 + 0.001 * rank, "cpu")
    info = D.last_broadcast_info()
    if rank == 0:
        print(json.dumps({"metric": "control-flow-only (no kernels, no throughput)", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ranks": torch.distributed.get_world_size() if world > 1 else 1,
                          "rccl_ranks": None,
                          "index_broadcast": {"transport": info.get("transport"), "n_ranks": info.get("n_ranks", 1), "bytes": info.get("bytes"),
                                              "seconds": info.get("seconds"),
                                              "gbps": round(info.get("bytes", 0) / max(info.get("seconds") or 0.0, 1e-9) / 1e9, 3) if world > 1 else None},
                          # the same self-diagnosis fields the measured line carries: per-rank spread of the timed region
                          "ms_per_step_ranks": {"min": round(min(secs) / args.steps * 1e3, 2), "max": round(max(secs) / args.steps * 1e3, 2),
                                                "all": [round(t / args.steps * 1e3, 2) for t in secs]},
                          "total_samples": total, "t_max": t_max, "baseline_config": args.config}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
