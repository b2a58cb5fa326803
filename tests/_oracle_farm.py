"""Host process pool that runs the full-length ORACLE legs of tests/test_fullsize_gpu.py while the GPU legs of the suite execute.

Test infrastructure only (like oracle/ itself): the workers import oracle.rvc_oracle and rvc_amd.lib.synthetic (seeded
state dicts), never the HIP library, and never touch the GPU.  One oracle evaluation per (kind, case): the results land as
.npy files in a session directory and every test that needs them reads them from there, so a 30 s oracle run is paid once
and overlaps the kernel tests instead of serialising in front of its own GPU leg (round 4: 1159 s of a 1200 s limit).

Jobs are plain tuples (picklable); a worker is a spawned Python process with its own torch thread pool."""
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (seconds, target rate, index rows, index_rate, audio seed, noise seed, vocoder, bf16 weight storage)
PEAKED = {
    "cfg2": (30, 48000, 100_000, 0.75, 0, 1234, "HiFi-GAN", False), "cfg1": (10, 40000, 0, 0.0, 0, 1234, "HiFi-GAN", False),
    "45s": (45, 48000, 0, 0.0, 45, 99, "HiFi-GAN", False), "cfg4": (30, 48000, 100_000, 0.75, 0, 1234, "MRF HiFi-GAN", True),
    "cfg5": (30, 48000, 2_000_000, 0.75, 0, 1234, "RefineGAN", False)}
# flat-salience (random-head RMVPE) runs: (seconds, rate, rows, index_rate, audio seed, noise seed, float32 search)
FLAT = {"cfg2": (30, 48000, 100_000, 0.75, 0, 1234, True), "cfg1": (10, 40000, 0, 0.0, 0, 1234, True),
        "45s": (45, 48000, 0, 0.0, 45, 99, False)}
DECODER_CASES = ["nsf", "mrf", "mrf-bf16", "refine"]
DECODER_T = 3198


def _worker_init(threads):
    sys.path[:0] = [REPO, os.path.join(REPO, "codename-rvc-fork-3_amd")]
    import torch
    torch.set_num_threads(threads)


def _save(out_dir, key, arrays):
    d = os.path.join(out_dir, key)
    os.makedirs(d, exist_ok=True)
    for name, a in arrays.items():
        np.save(os.path.join(d, name + ".npy"), np.asarray(a))
    return d


def _pipeline_job(out_dir, key, peaked, spec, index_file, f0_override_file):
    """oracle.pipeline on one full-length input.  spec = the PEAKED / FLAT tuple."""
    import torch
    from oracle import rvc_oracle as O
    from rvc_amd.lib import synthetic as S
    t0 = time.time()
    if peaked:
        secs, sr, rows, rate, aseed, seed, voc, bf16 = spec
        f32 = True
    else:
        (secs, sr, rows, rate, aseed, seed, f32), voc, bf16 = spec, "HiFi-GAN", False
    rm_sd = S.make_rmvpe_state_dict(0, peaked=peaked)
    hub_sd = S.make_hubert_state_dict(1)
    cpt = S.make_synth_checkpoint(sr, voc, seed=0, smooth_pitch=peaked)
    big = None
    if index_file:
        big = np.load(index_file, mmap_mode="r")
    elif rows:
        big = S.synth_index(rows, seed=0)
    audio = S.synth_audio(16000 * secs, seed=aseed)
    kw = {}
    if f32:
        kw["knn_dtype"] = np.float32
    if bf16:
        kw["dec_bf16"] = True
    if f0_override_file:
        kw["f0_override"] = np.load(f0_override_file)
    taps = {}
    t1 = time.time()
    torch.manual_seed(seed)
    want = O.pipeline(hub_sd, rm_sd, cpt, audio.copy(), sid=0, pitch=0, big_npy=big, index_rate=rate, protect=0.5, taps=taps, **kw)
    t_oracle = time.time() - t1
    arrays = {"want": want, "f0_raw": taps["f0_raw"], "opt_ts": np.asarray(taps["opt_ts"], dtype=np.int64),
              "t_oracle": np.float64(t_oracle), "t_job": np.float64(time.time() - t0)}
    if "salience" in taps:
        arrays["salience"] = taps["salience"]
    return _save(out_dir, key, arrays)


class SeededNoise:
    """The oracle's noise interface (rand / randn in draw order) on a seeded CPU generator: the product side replays the
    same generator tensor by tensor, so RefineGAN's 24 AdaIN draws (3.8 GB at T = 3198) never sit in one Python list."""

    def __init__(self, seed):
        import torch
        self.torch = torch
        self.g = torch.Generator().manual_seed(seed)

    def randn(self, *shape):
        return self.torch.randn(*shape, generator=self.g)

    def rand(self, *shape):
        return self.torch.rand(*shape, generator=self.g)


def decoder_inputs(T=DECODER_T):
    """z, g, f0 of test_decoder_T3198_stage_by_stage_vs_oracle and the generator positioned behind them."""
    import torch
    gen = torch.Generator().manual_seed(17)
    z = torch.randn(1, 192, T, generator=gen)
    g = torch.randn(1, 256, 1, generator=gen)
    t = torch.arange(T) / 100.0
    f0 = (180.0 + 40.0 * torch.sin(2 * np.pi * 0.5 * t)).float().unsqueeze(0)
    f0[:, 500:600] = 0.0                                   # an unvoiced stretch: noise-only source, phase carry restarts
    return z, g, f0, gen


def _decoder_job(out_dir, key, case):
    """One vocoder at T = 3198 in the oracle with a tap on the source and after every stage."""
    import torch
    from oracle import rvc_oracle as O
    from rvc_amd.lib import synthetic as S
    t0 = time.time()
    voc = {"nsf": "HiFi-GAN", "mrf": "MRF HiFi-GAN", "refine": "RefineGAN"}[case.split("-")[0]]
    bf16 = case.endswith("bf16")
    cpt = S.make_synth_checkpoint(48000, voc, seed=0)
    w = O.fold_weight_norm(cpt["weight"])
    if bf16:
        w = {k: (v.float().bfloat16().float() if k.startswith("dec.") else v) for k, v in w.items()}
    rates, ksizes = cpt["config"][12], cpt["config"][14]
    upp = int(np.prod(rates))
    z, g, f0, gen = decoder_inputs()
    taps = {}
    with torch.no_grad():
        if voc == "HiFi-GAN":
            src_randn = torch.randn(1, DECODER_T * upp, 1, generator=gen)
            ref = O.decoder_nsf(w, z, f0, g, rates, ksizes, 48000, O.ListNoise([torch.zeros(1, 1, 1), src_randn]), taps=taps)
        elif voc == "MRF HiFi-GAN":
            ref = O.decoder_mrf(w, z, f0, g, rates, ksizes, 48000, SeededNoise(23), taps=taps)
        else:
            ref = O.decoder_refine(w, z, f0, g, rates, 48000, SeededNoise(23), taps=taps)
    arrays = {"ref": ref.numpy(), "har_source": taps["har_source"].numpy(), "t_oracle": np.float64(time.time() - t0)}
    for s in range(len(rates)):
        arrays[f"stage{s}"] = taps[f"stage{s}"].numpy()
    return _save(out_dir, key, arrays)


def draw_index_on_device(n_rows, dev, seed=0, n_centres=512, jitter=0.05, dim=768):
    """synthetic.synth_index's recipe (cluster centres + jitter) drawn on the device, as bench.py does for cfg 5: 2 M rows
    are 6.1 GB, too slow to draw with NumPy inside a test.  (Runs in the pytest process, not in a farm worker.)"""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    centres = torch.randn(n_centres, dim, device=dev, generator=g) * 0.35
    out = torch.empty(n_rows, dim, device=dev)
    for s in range(0, n_rows, 1 << 18):
        e = min(n_rows, s + (1 << 18))
        which = torch.randint(0, n_centres, (e - s,), device=dev, generator=g)
        out[s:e] = centres[which] + jitter * torch.randn(e - s, dim, device=dev, generator=g)
    return out


class Result:
    """Lazy view of one job's directory."""

    def __init__(self, d):
        self.d = d

    def __getitem__(self, name):
        return np.load(os.path.join(self.d, name + ".npy"))

    def __contains__(self, name):
        return os.path.exists(os.path.join(self.d, name + ".npy"))


class Farm:
    def __init__(self, workers=None, threads=None):
        import multiprocessing as mp
        ncpu = os.cpu_count() or 8
        self.threads = threads or max(2, min(32, ncpu // 4))
        self.workers = workers or max(1, min(6, ncpu // self.threads))
        base = "/dev/shm" if os.path.isdir("/dev/shm") and ncpu >= 64 else None   # big box: results stay in memory
        self.dir = tempfile.mkdtemp(prefix="rvc_oracle_farm_", dir=base)
        self.pool = ProcessPoolExecutor(self.workers, mp_context=mp.get_context("spawn"), initializer=_worker_init,
                                        initargs=(self.threads,))
        self.futures = {}
        self.t0 = time.time()

    def submit_pipeline(self, key, peaked, spec, index_file=None, f0_override=None):
        if key in self.futures:
            return
        f0_file = None
        if f0_override is not None:
            f0_file = os.path.join(self.dir, key.replace(":", "_") + "_f0.npy")
            np.save(f0_file, np.asarray(f0_override))
        self.futures[key] = self.pool.submit(_pipeline_job, self.dir, key.replace(":", "_"), peaked, spec, index_file, f0_file)

    def submit_decoder(self, case):
        key = "decoder:" + case
        if key not in self.futures:
            self.futures[key] = self.pool.submit(_decoder_job, self.dir, key.replace(":", "_"), case)

    def submit_all(self, cfg5_index_file=None, which=None):
        """Everything test_fullsize_gpu.py will ask for, longest first."""
        def on(k):
            return which is None or k in which
        if cfg5_index_file and on("peaked:cfg5"):
            self.submit_pipeline("peaked:cfg5", True, PEAKED["cfg5"], index_file=cfg5_index_file)
        for case in ("45s", "cfg2", "cfg4", "cfg1"):
            if on("peaked:" + case):
                self.submit_pipeline("peaked:" + case, True, PEAKED[case])
        for case in ("45s", "cfg2", "cfg1"):
            if on("flat:" + case):
                self.submit_pipeline("flat:" + case, False, FLAT[case])
        for case in DECODER_CASES:
            if on("decoder:" + case):
                self.submit_decoder(case)

    def get(self, key, timeout=1500):
        t0 = time.time()
        d = self.futures[key].result(timeout=timeout)
        waited = time.time() - t0
        if waited > 1.0:
            print(f"[oracle farm] waited {waited:.0f} s for {key} ({time.time() - self.t0:.0f} s after the farm started)")
        return Result(d)

    def close(self):
        """Stop the farm NOW: jobs that are already running (30-45 s oracle pipelines on 32 threads each) are terminated, not
        waited for -- an early-exit session (-x, Ctrl-C, a failed GPU leg) must not keep the host busy, write into the deleted
        directory or hang in the executor's join at interpreter exit."""
        procs = list(getattr(self.pool, "_processes", {}).values())   # (no public handle on the workers of a ProcessPoolExecutor)
        self.pool.shutdown(wait=False, cancel_futures=True)
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()
        shutil.rmtree(self.dir, ignore_errors=True)
