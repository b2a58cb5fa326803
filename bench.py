#!/usr/bin/env python3
"""bench.py -- end-to-end voice-conversion throughput of the MI355X path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic utterance: `Pipeline.pipeline` from the 16 kHz input
array to the float32 48 kHz waveform, BASELINE cfg 2 (30 s clip, HuBERT-base + NSF-HiFi-GAN 48k, 100k x 768
feature index, index_rate 0.75, rmvpe, protect 0.5).  Weights are seeded random-init (no network), inputs are
synthetic.  With N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL) every rank
converts its own utterances (utterance i -> rank i mod N); the index is built on rank 0 and replicated with one
broadcast at load; the steady state has no collective ("scaling": "weak").  On each GPU `--inflight` utterances
(default 2) are in flight at a time, each on its own HIP stream (VoiceConverter.convert_batch): the K timed steps
are K utterances either way; `--inflight 1` is the strictly one-after-the-other schedule.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel (the 11-tap 128-channel ResBlock conv of vocoder stage 1, fp32 MFMA implicit
                GEMM), timed live with HIP events on the launch stream
  roofline_knn  the L2 top-8 kernel against HBM bytes per query-tile pass (SURVEY §8d definition) and fp32 MFMA
  cpu_baseline  the oracle (CPU restatement of the reference, oracle/rvc_oracle.py) timed on the host cores on
                a bounded sample of the same workload (rank 0, N = 1 only)
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

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = FP32 vector peak
PEAK_HBM_GBS = 8000.0           # HBM3E spec
# average HBM bytes per launch of the roofline kernel symbol at the cfg-2 shape, from the PMC passes committed under
# profiles/ (cannot be collected inside bench.py: it needs rocprofv3)
PMC_TRAFFIC_BYTES = 641.0e6
PMC_TRAFFIC_SOURCE = ("profiles/r01_pmc_conv.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                      "tools/pmc_conv.py (this same launch mix); FETCH_SIZE calibrated on the same kernel at K=1 with known bytes")


def roofline_mix(native, dev, T, rates, k=11):
    """The launches of conv_mfma_kernel<11,2,2,2,2,4,false> in one utterance's vocoder forward: the 11-tap ResBlock of
    stage 1 (C = 128; the 38k-column stage 0 takes the 128x64-tile symbol), for each dilation d: conv1 (dilation d) then
    conv2 (dilation 1, + residual; the last one also + running sum, x 1/3).
    Returns (callable, flops per call, launches per call, algorithmic HBM bytes per call)."""
    shapes = [(128, T * rates[0] * rates[1])]
    state, flops, alg_bytes = [], 0.0, 0.0
    gen = torch.Generator().manual_seed(1)
    for C, L in shapes:
        x = torch.randn(1, C, L, device=dev)
        t1 = torch.empty_like(x)
        y = torch.randn(1, C, L, device=dev)
        acc = torch.randn(1, C, L, device=dev)
        w1 = native.conv1d_pack_weight(torch.randn(C, C, k, generator=gen) * 0.02, dev)
        w2 = native.conv1d_pack_weight(torch.randn(C, C, k, generator=gen) * 0.02, dev)
        bias = torch.zeros(C, device=dev)
        state.append((C, x, t1, y, acc, w1, w2, bias))
        flops += 6 * 2.0 * C * C * k * L
        tensor = C * L * 4.0
        alg_bytes += 3 * (2 * tensor) + 2 * (3 * tensor) + 1 * (4 * tensor)   # conv1: r+w; conv2: r+res+w (+acc)

    def run():
        for C, x, t1, y, acc, w1, w2, bias in state:
            for j, d in enumerate((1, 3, 5)):
                native.conv1d_forward_into(x, w1, bias, C, k, d, 0.1, out=t1)
                if j < 2:
                    native.conv1d_forward_into(t1, w2, bias, C, k, 1, 0.1, res=x, out=y)
                else:
                    native.conv1d_forward_into(t1, w2, bias, C, k, 1, 0.1, res=x, acc=acc, out_scale=1 / 3, out=y)
    return run, flops, 6, alg_bytes


def decoder_flops(T, rates, ksizes, c0=512, cin=192, res_k=(3, 7, 11), n_dil=3):
    """Closed form of SURVEY §8d (2 x MACs)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=30.0, help="clip length (BASELINE cfg 2 = 30 s)")
    ap.add_argument("--index-rows", type=int, default=100_000)
    ap.add_argument("--inflight", type=int, default=2, help="utterances in flight per GPU, each on its own HIP stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="clip length of the bounded CPU-baseline sample")
    args = ap.parse_args()

    from rvc_amd.infer import distributed as D
    rank, world, local = D.init_process_group()
    assert world == max(1, args.gpus) or world == 1, (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py measures the HIP path; there is no CPU fallback"
    local = local % torch.cuda.device_count()   # ranks > devices only in the gloo control-flow test (RVC_DIST_BACKEND)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:   # N host processes share the node's cores: keep each rank's CPU thread pool small
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))

    from rvc_amd import _native
    from rvc_amd.infer.infer import VoiceConverter
    from rvc_amd.lib import synthetic as S

    # ---- load (untimed): weights, index broadcast ----
    sr = 48000
    cpt = S.make_synth_checkpoint(sr, "HiFi-GAN", seed=0)
    vc = VoiceConverter(device=dev)
    vc.load_checkpoint_dict(cpt)
    vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
    vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
    big = S.synth_index(args.index_rows, seed=0) if rank == 0 else None
    t0 = time.perf_counter()
    index_dev = D.broadcast_index(big, dev)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    assert D.checksums_agree(index_dev), "feature index differs across ranks after the broadcast"
    vc.vc.set_index(index_dev)

    n_in = int(round(args.seconds * 16000))
    n_total = args.steps + args.warmup
    # utterance i (global) uses rng seed i; this rank converts i = rank, rank + world, ...
    audios_host = [S.synth_audio(n_in, seed=rank + world * j) for j in range(min(n_total, 4))]
    audios = [torch.from_numpy(a).to(dev) for a in audios_host]   # inputs resident in HBM before the timed region

    def step(j, host_io=False):
        a = (audios_host if host_io else audios)[j % len(audios)]
        return vc.convert_array(a, index_path="", index_rate=0.75, protect=0.5, sid=0)

    # `--inflight` utterances are on the GPU at a time, each on its own HIP stream (VoiceConverter.convert_batch): the K
    # timed steps are K utterances, interleaved 2 by 2 by default.  --inflight 1 is the strictly sequential schedule.
    inflight = max(1, args.inflight)
    kw = dict(index_path="", index_rate=0.75, protect=0.5, sid=0)
    if inflight == 1:
        for j in range(args.warmup):
            out = step(j)
    else:   # every stream warms up its own workspaces / side stream (W steps per stream)
        vc.convert_batch([audios[j % len(audios)] for j in range(args.warmup * inflight)], inflight=inflight, **kw)
    if world > 1:
        torch.distributed.barrier(**({} if os.environ.get("RVC_DIST_BACKEND") == "gloo" else {"device_ids": [local]}))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = 0
    if inflight == 1:
        for j in range(args.steps):
            out = step(args.warmup + j)
            samples += out.shape[0]
    else:
        outs = vc.convert_batch([audios[(args.warmup + j) % len(audios)] for j in range(args.steps)], inflight=inflight, **kw)
        samples = sum(int(o.shape[0]) for o in outs)
        out = outs[-1]
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier(**({} if os.environ.get("RVC_DIST_BACKEND") == "gloo" else {"device_ids": [local]}))
    elapsed = time.perf_counter() - t0
    total_samples, t_max = D.reduce_report(samples, elapsed, dev)
    # outside the timed region: the last waveform must be finite and inside [-1, 1] (the BiGRU poisons its output with NaN if
    # its workgroups ever fail to rendezvous; a silent NaN utterance must not count as throughput)
    assert bool(torch.isfinite(out).all()) and float(out.abs().max()) <= 1.0, "bench produced a non-finite or unnormalised waveform"

    # PCIe-inclusive variant (host NumPy in, host NumPy out), reported beside `value`, never as `value`
    host_steps = min(args.steps, 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(host_steps):
        step(j, host_io=True)
    torch.cuda.synchronize()
    host_io_rate = host_steps * int(out.shape[0]) / (time.perf_counter() - t0)

    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    # ---- roofline of the dominant kernel symbol: conv_mfma_kernel<11,2,2,2,2,4,false> (stage-1 11-tap ResBlock convs) ----
    rates, ksizes = cpt["config"][12], cpt["config"][14]
    n_pad = n_in + 32000                              # 1 s reflect pad each side (pipeline.py:581)
    T = min(n_pad // 160, 2 * ((n_pad - 400) // 320 + 1))   # synth frames (pipeline.py:467)
    run_mix, mix_flops, mix_launches, mix_alg_bytes = roofline_mix(_native, dev, T, rates)
    for _ in range(2):
        run_mix()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run_mix()
    e1.record()
    torch.cuda.synchronize()
    t_launch = e0.elapsed_time(e1) / (reps * mix_launches) * 1e-3
    flops_launch = mix_flops / mix_launches
    cfg2 = T == 3198 and list(rates[:2]) == [12, 10]
    roofline = {"kernel": "rvc::conv_mfma_kernel<11,2,2,2,2,4,false>: the 6 launches per utterance of the 11-tap ResBlock convs of vocoder "
                          "stage 1 (C=128, 383 760 columns), in the decoder's own mix (dilations 1/3/5, residual on every second one)",
                "bound": "mfma", "achieved": round(flops_launch / t_launch / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(flops_launch / t_launch / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                "traffic": PMC_TRAFFIC_BYTES if cfg2 else None,
                "traffic_source": PMC_TRAFFIC_SOURCE if cfg2 else None,
                "algorithmic_bytes_per_launch": round(mix_alg_bytes / mix_launches),
                "flops_per_launch": flops_launch, "avg_launch_ms": round(t_launch * 1e3, 4), "launches_per_utterance": mix_launches,
                "timing": "HIP events around the isolated launch mix on the launch stream (nothing else running); rocprofv3 "
                          "--stats agrees for a sequential run (profiles/r01_bench_kernel_stats_inflight1.csv); with two "
                          "utterances in flight a kernel's traced duration also contains the time it shares the chip"}
    del run_mix

    # whole vocoder, timed with events around rvc_decoder_forward
    z = torch.randn(1, 192, T, device=dev)
    f0 = torch.full((1, T), 220.0, device=dev)
    gv = torch.randn(1, 256, device=dev)
    nz = torch.randn(1, T * 480, 1, device=dev)
    vc.net_g.dec.forward(z, f0, gv, src_randn=nz)
    e0.record()
    for _ in range(3):
        vc.net_g.dec.forward(z, f0, gv, src_randn=nz)
    e1.record()
    torch.cuda.synchronize()
    t_dec = e0.elapsed_time(e1) / 3 * 1e-3
    dflops = decoder_flops(T, rates, ksizes)

    # kNN kernel
    F_ = (n_in + 32000 - 400) // 320 + 1
    q = index_dev[torch.randint(0, index_dev.shape[0], (F_,), device=dev)] + 0.03 * torch.randn(F_, 768, device=dev)
    idx = vc.vc._preset_index
    idx.search_device(q)
    e0.record()
    for _ in range(5):
        idx.search_device(q)
    e1.record()
    torch.cuda.synchronize()
    t_knn = e0.elapsed_time(e1) / 5 * 1e-3
    passes = -(-F_ // 128)
    knn_bytes = passes * index_dev.shape[0] * 768 * 4.0
    knn_flops = 2.0 * F_ * index_dev.shape[0] * 768
    roofline_knn = {"kernel": "knn_partial_kernel + knn_merge_kernel", "bound": "hbm", "query_tile": 128, "passes": passes,
                    "achieved": round(knn_bytes / t_knn / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(knn_bytes / t_knn / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                    "mfma_tflops": round(knn_flops / t_knn / 1e12, 2),
                    "mfma_frac": round(knn_flops / t_knn / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "avg_launch_ms": round(t_knn * 1e3, 4)}

    # the kernel's HBM-bound regime: <= 32 queries (one MFMA column tile) against a cfg-5-sized index that cannot sit in
    # the 256 MB Infinity Cache; algorithmic bytes = one pass over the index
    roofline_knn_stream = None
    try:
        n_big = 2_000_000
        big_idx = torch.empty(n_big, 768, device=dev)
        for s0 in range(0, n_big, 250_000):
            big_idx[s0:s0 + 250_000].normal_(0, 0.35)
        big_norms = _native.knn_index_norms(big_idx)
        q32 = big_idx[torch.randint(0, n_big, (32,), device=dev)] + 0.03 * torch.randn(32, 768, device=dev)
        for _ in range(2):
            _native.knn_search(big_idx, big_norms, q32)
        e0.record()
        for _ in range(5):
            _native.knn_search(big_idx, big_norms, q32)
        e1.record()
        torch.cuda.synchronize()
        t_s = e0.elapsed_time(e1) / 5 * 1e-3
        sbytes = n_big * 768 * 4.0
        roofline_knn_stream = {"kernel": "knn_direct_kernel<8,3> + knn_merge_kernel, 32 queries x 2 000 000 rows (6.1 GB index), one pass",
                               "bound": "hbm", "achieved": round(sbytes / t_s / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(sbytes / t_s / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                               "avg_search_ms": round(t_s * 1e3, 4)}
        del big_idx, big_norms
    except torch.OutOfMemoryError:
        pass

    # ---- CPU baseline: the oracle on the host cores, bounded sample ----
    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import rvc_oracle as O
        cores = min(os.cpu_count() or 1, 32)   # the restatement's torch ops stop scaling (and thrash) far below 256 threads
        torch.set_num_threads(cores)
        a = S.synth_audio(int(args.cpu_seconds * 16000), seed=0)
        hub_sd, rm_sd = S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0)
        big_host = index_dev.cpu().numpy()
        torch.manual_seed(0)
        t0 = time.perf_counter()
        ref = O.pipeline(hub_sd, rm_sd, cpt, a, sid=0, pitch=0, big_npy=big_host, index_rate=0.75, protect=0.5,
                         knn_dtype=np.float32)
        t_cpu = time.perf_counter() - t0
        cpu_baseline = {"value": round(ref.shape[0] / t_cpu, 1), "unit": "samples/s", "cores": cores, "kind": "port",
                        "sample": f"{args.cpu_seconds:g} s clip, same cfg-2 settings (48k NSF, {args.index_rows}x768 index, "
                                  f"index_rate 0.75), 1 run of oracle.pipeline, torch threads = {cores}",
                        "seconds": round(t_cpu, 2)}

    value = total_samples / t_max
    line = {
        "metric": "48 kHz audio samples/sec end-to-end VC",
        "value": round(value, 1),
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(t_max / args.steps * 1e3, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded random-init weights, FM-tone utterances, clustered index)",
        "rtf": round(value / sr, 2),
        "config": {"workload": f"BASELINE cfg 2: {args.seconds:g} s 16 kHz clip -> 48 kHz, HuBERT-base + NSF-HiFi-GAN 48k, "
                               f"{args.index_rows}x768 index, index_rate 0.75, rmvpe, protect 0.5; 1 utterance per step per GPU, "
                               f"{max(1, args.inflight)} utterance(s) in flight per GPU on separate HIP streams",
                   "samples_per_step": int(out.shape[0]), "parallelism": f"utterance-sharded x{world}",
                   "input_residency": "16 kHz float64 utterances resident in HBM before the timed region; waveform left in HBM",
                   "host_io_samples_per_s_rank0": round(host_io_rate, 1),
                   "index_broadcast_s": round(t_bcast, 4)},
        "roofline": roofline,
        "roofline_knn": roofline_knn,
        "roofline_knn_stream": roofline_knn_stream,
        "decoder": {"tflops_per_utterance": round(dflops / 1e12, 4), "ms": round(t_dec * 1e3, 2),
                    "achieved_tflops": round(dflops / t_dec / 1e12, 2),
                    "frac_of_fp32_peak": round(dflops / t_dec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
        "cpu_baseline": cpu_baseline,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
