"""Full-length parity on the MI355X: the shapes bench.py times (BASELINE cfg 1 / cfg 2), the multi-segment path, the
vocoder at T = 3198 stage by stage, and the a18 caller -- each against the oracle run on the box's host cores under the
same seed (or against a reference-generated fixture).  Gate: 1e-3 waveform RMS (north_star); integers bit-exact.
The oracle runs take tens of seconds each; they are the price of not extrapolating parity from 3 s clips."""
import os
import time

import numpy as np
import pytest
import torch

from conftest import load_golden, rms

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def S():
    from rvc_amd.lib import synthetic
    return synthetic


@pytest.fixture(scope="module")
def hubert(S):
    from rvc_amd.lib.hubert import HubertModelWithFinalProj
    return HubertModelWithFinalProj(S.make_hubert_state_dict(1), device=DEV)


@pytest.fixture(scope="module")
def sds(S):
    return S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0)


def _converter(S, sr, voc, hubert, config=None):
    from rvc_amd.infer.infer import VoiceConverter
    from rvc_amd.infer.pipeline import Pipeline
    vc = VoiceConverter(device=DEV)
    vc.load_checkpoint_dict(S.make_synth_checkpoint(sr, voc, seed=0))
    vc.hubert_model = hubert
    if config is not None:
        vc.vc = Pipeline(sr, config)
    vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
    return vc


def _coarse_mismatch(vc, audio, oracle_coarse):
    """End-to-end coarse-pitch integers of the product (GPU U-Net -> BiGRU -> decode -> threshold table) vs the oracle's."""
    from oracle import rvc_oracle as O
    a = np.pad(O.highpass(audio), (16000, 16000), mode="reflect")
    f0 = vc.vc.model_rmvpe.infer_from_audio_device(torch.from_numpy(a).float().to(DEV), thred=0.03)
    coarse, _ = vc.vc._postprocess_f0_device(f0, 0)
    got = coarse[:len(oracle_coarse)].cpu().numpy()
    return float((got != oracle_coarse).mean()), int(np.abs(got - oracle_coarse).max())


@pytest.mark.parametrize("cfg", [2, 1])
def test_baseline_config_full_length_vs_oracle(S, hubert, sds, cfg):
    """cfg 2 exactly as bench.py runs it (30 s, 48 k NSF, 100 000-row index, index_rate 0.75) and cfg 1 (10 s, 40 k,
    index_rate 0): whole Pipeline.pipeline vs oracle.pipeline under one seed.  Reference: pipeline.py:509-694."""
    from oracle import rvc_oracle as O
    secs, sr, rows, rate = (30, 48000, 100_000, 0.75) if cfg == 2 else (10, 40000, 0, 0.0)
    cpt = S.make_synth_checkpoint(sr, "HiFi-GAN", seed=0)
    vc = _converter(S, sr, "HiFi-GAN", hubert)
    big = S.synth_index(rows, seed=0) if rows else None
    if rows:
        vc.vc.set_index(big)
    audio = S.synth_audio(16000 * secs, seed=0)
    taps = {}
    t0 = time.time()
    torch.manual_seed(1234)
    want = O.pipeline(sds[0], sds[1], cpt, audio.copy(), sid=0, pitch=0, big_npy=big, index_rate=rate, protect=0.5, taps=taps,
                      knn_dtype=np.float32)
    t_oracle = time.time() - t0
    got = vc.vc.pipeline(hubert, vc.net_g, 0, audio.copy(), 0, "rmvpe", "", rate, True, 3, 1, "v2", 0.5, 128, False, 1, None,
                         noise_seed=1234)
    assert got.dtype == np.float32 and got.shape == want.shape == ((1_439_040,) if cfg == 2 else (399_200,))
    err = rms(got - want)
    mism, worst = _coarse_mismatch(vc, audio, taps["coarse"])
    print(f"cfg {cfg} full length: rms err {err:.3e} (oracle rms {rms(want):.3f}, oracle {t_oracle:.0f} s); coarse f0 bins: "
          f"{100 * mism:.3f} % of {len(taps['coarse'])} frames differ (max |delta| {worst})")
    assert err <= 1e-3, err
    # coarse bins are bit-exact on identical salience (test_rmvpe_matches_reference_golden); end to end they inherit the
    # U-Net's ~1e-3 salience difference through an argmax -> a frame can land in the neighbouring 20-cent peak
    assert mism <= 0.01, mism


def test_multi_segment_matches_reference_golden(S, hubert):
    """pipeline.py:563-577, 614-681 on the product: split points, per-segment HuBERT + synthesis, RNG stream shared by the
    segments, crops, concat -- against the REFERENCE's own output (fixture of make_golden.py::multiseg, tier 1/3/4)."""
    g = load_golden("pipeline_multiseg")

    class Tier:
        x_pad, x_query, x_center, x_max, device = 1, int(g["x_query"]), int(g["x_center"]), int(g["x_max"]), DEV

    vc = _converter(S, 48000, "HiFi-GAN", hubert, config=Tier())
    vc.vc.set_index(S.synth_index(4096, seed=0))
    out = vc.vc.pipeline(hubert, vc.net_g, int(g["sid"]), g["audio"].copy(), 0, "rmvpe", "", float(g["index_rate"]), True, 3, 1,
                         "v2", float(g["protect"]), 128, False, 1, None, noise_seed=int(g["seed"]))
    assert out.dtype == np.float32 and out.shape == g["out"].shape
    err = rms(out - g["out"])
    print(f"multi-segment (3 segments) vs reference: rms err {err:.3e} (ref rms {rms(g['out']):.3f})")
    assert err <= 1e-3, err


def test_45s_two_segments_vs_oracle(S, hubert, sds):
    """> 41 s at the default tier: the host-filtfilt branch, two segments, waveform (not only lengths) vs the oracle."""
    from oracle import rvc_oracle as O
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
    vc = _converter(S, 48000, "HiFi-GAN", hubert)
    audio = S.synth_audio(16000 * 45, seed=45)
    taps = {}
    torch.manual_seed(99)
    want = O.pipeline(sds[0], sds[1], cpt, audio.copy(), sid=0, pitch=0, taps=taps)
    got = vc.vc.pipeline(hubert, vc.net_g, 0, audio.copy(), 0, "rmvpe", "", 0.0, True, 3, 1, "v2", 0.5, 128, False, 1, None,
                         noise_seed=99)
    assert len(taps["opt_ts"]) == 1 and got.shape == want.shape == (int(load_golden("segmentation")["outlen_45"]),)
    err = rms(got - want)
    print(f"45 s, 2 segments: rms err {err:.3e} (oracle rms {rms(want):.3f})")
    assert err <= 1e-3, err


@pytest.mark.parametrize("hint", [1, 2])
def test_decoder_T3198_stage_by_stage_vs_oracle(S, hint):
    """The vocoder at the benchmarked shape (T = 3198 -> 1 535 040 samples) with a tap after every stage, under both tile
    selections of the short first stage (rvc_set_concurrency_hint 1: 128x64 tiles, 2: 128x128).
    Reference: hifigan_nsf.py:173-207."""
    from oracle import rvc_oracle as O
    from rvc_amd import _native
    from rvc_amd.lib.algorithm.weights import fold_weight_norm
    T = 3198
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
    w = O.fold_weight_norm(cpt["weight"])
    rates, ksizes = cpt["config"][12], cpt["config"][14]
    gen = torch.Generator().manual_seed(17)
    z = torch.randn(1, 192, T, generator=gen)
    g = torch.randn(1, 256, 1, generator=gen)
    t = torch.arange(T) / 100.0
    f0 = (180.0 + 40.0 * torch.sin(2 * np.pi * 0.5 * t)).float().unsqueeze(0)
    f0[:, 500:600] = 0.0                                   # an unvoiced stretch: noise-only source, phase carry restarts
    src_randn = torch.randn(1, T * 480, 1, generator=gen)
    taps = {}
    ref = O.decoder_nsf(w, z, f0, g, rates, ksizes, 48000, O.ListNoise([torch.zeros(1, 1, 1), src_randn]), taps=taps).numpy()
    folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
    dec = _native.Decoder("HiFi-GAN", 48000, folded)
    _native.set_concurrency_hint(hint)
    try:
        chans, length = 512, T
        for stage in range(-1, 4):
            if stage >= 0:
                chans, length = chans // 2, length * rates[stage]
            shape = (1, T * 480) if stage < 0 else (1, chans, length)
            tap = torch.zeros(shape, device=DEV)
            dec.set_tap(stage, tap)
            out = dec.forward(z.to(DEV), f0.to(DEV), g[:, :, 0].to(DEV), src_randn=src_randn.to(DEV))
            torch.cuda.synchronize()
            dec.set_tap(stage, None)
            want = taps["har_source"].reshape(1, -1) if stage < 0 else taps[f"stage{stage}"]
            e = rms(tap.cpu().numpy() - want.numpy())
            print(f"hint {hint} stage {stage}: rms err {e:.3e} (oracle rms {rms(want.numpy()):.3f})")
            assert e <= 1e-4 * max(1.0, rms(want.numpy())), (stage, e)
    finally:
        _native.set_concurrency_hint(1)
    err = rms(out.cpu().numpy() - ref)
    print(f"hint {hint} waveform: rms err {err:.3e}")
    assert err <= 5e-5, err


def test_convert_array_caller_vs_oracle(S, hubert, sds, tmp_path):
    """a18 (infer.py:262-311): the peak limit to 0.95 and the index-path normalisation ("trained" -> "added", quotes and
    blanks stripped) as VoiceConverter.convert_array applies them, against the oracle fed with the same front end."""
    from oracle import rvc_oracle as O
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
    vc = _converter(S, 48000, "HiFi-GAN", hubert)
    big = S.synth_index(3000, seed=4)
    np.save(os.path.join(tmp_path, "added_IVF42_Flat.npy"), big)
    asked = '  "' + os.path.join(tmp_path, "trained_IVF42_Flat.npy") + '" \n'     # what a UI text box hands over
    audio = 4.0 * S.synth_audio(30_000, seed=8)                                   # peaks ~1.3: the limiter engages
    assert np.abs(audio).max() > 1.0
    limited = audio / (np.abs(audio).max() / 0.95)                                # infer.py:262-265
    torch.manual_seed(606)
    want = O.pipeline(sds[0], sds[1], cpt, limited.copy(), sid=0, pitch=0, big_npy=big, index_rate=0.6, protect=0.5)
    got = vc.convert_array(audio, index_path=asked, index_rate=0.6, protect=0.5, sid=0, noise_seed=606)
    assert got.shape == want.shape
    err = rms(got - want)
    print(f"convert_array (limiter + index path): rms err {err:.3e}")
    assert err <= 1e-3, err
    no_index = vc.convert_array(audio, index_path="", index_rate=0.6, protect=0.5, sid=0, noise_seed=606)
    assert rms(no_index - got) > 1e-3                                             # the index file really was found and used


def test_rccl_single_rank_broadcast_and_device_checksum():
    """The C-ABI RCCL path on one GPU: bind librccl at run time, ncclCommInitRank with one rank, broadcast in place,
    ncclCommCount == 1; rvc_checksum64 equals the documented formula evaluated in NumPy."""
    from rvc_amd import _native
    from rvc_amd.infer import distributed as D
    comm = _native.Comm(_native.comm_unique_id(), 1, 0)
    info = comm.info()
    print("rccl:", info)
    assert info["n_ranks"] == 1 and info["rank"] == 0 and info["rccl_version"] > 0 and "rccl" in info["library"]
    t = torch.randn(1000, 768, device=DEV)
    before = t.clone()
    comm.broadcast_(t, 0)
    torch.cuda.synchronize()
    assert torch.equal(t, before)
    comm.destroy()
    for n in (1, 3, 1000 * 768, 7_654_321):
        x = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int64, device=DEV).to(torch.int32)
        assert D.tensor_checksum(x) == D.tensor_checksum(x.cpu()), n
    x = torch.arange(10, dtype=torch.uint8, device=DEV)                # 2 words + a 2-byte tail
    assert D.tensor_checksum(x) == D.tensor_checksum(x.cpu())
    # broadcast_index with force_rccl on one rank: the bench's N = 1 path
    idx = D.broadcast_index(np.ones((64, 768), dtype=np.float32), DEV, force_rccl=True)
    assert idx.shape == (64, 768) and D.last_broadcast_info()["n_ranks"] == 1
    D.destroy_native_comm()
