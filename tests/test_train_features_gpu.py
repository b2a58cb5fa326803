"""SURVEY §8f rank 3-4 on the MI355X: the K4b transform (training spectrogram / log-mel) against the reference fixture,
the device resampler against SciPy's polyphase resampler driven with the same FIR, the extraction module against the
oracle, and convert_audio on a 44.1 kHz stereo float WAV."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rms

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_training_spectrogram_and_mel_match_reference_fixture():
    from rvc_amd.train import mel_processing as MP
    g = load_golden("train_features")
    for tag, sr, hop, n_mels in (("48k", 48000, 480, 128), ("40k", 40000, 400, 125)):
        y = torch.from_numpy(g[f"audio_{tag}"]).unsqueeze(0).to(DEV)
        spec = MP.spectrogram_torch(y, 2048, hop, 2048, center=False)[0].cpu().numpy()
        mel = MP.mel_spectrogram_torch(y, 2048, n_mels, sr, hop, 2048, 0.0, None, center=False)[0].cpu().numpy()
        assert spec.shape == g[f"spec_{tag}"].shape and mel.shape == g[f"mel_{tag}"].shape
        # DFT by fp32 GEMM vs torch's FFT: absolute error ~1e-4 on magnitudes up to ~400; log domain 2e-3 like K4
        assert np.abs(spec - g[f"spec_{tag}"]).max() <= 2e-3 * max(1.0, np.abs(g[f"spec_{tag}"]).max() / 100), np.abs(spec - g[f"spec_{tag}"]).max()
        assert np.abs(mel - g[f"mel_{tag}"]).max() <= 2e-3, np.abs(mel - g[f"mel_{tag}"]).max()
        # spec_to_mel_torch on the device spectrogram closes the loop
        mel2 = MP.spec_to_mel_torch(torch.from_numpy(spec).unsqueeze(0).to(DEV), 2048, n_mels, sr, 0.0, None)[0].cpu().numpy()
        assert np.abs(mel2 - g[f"mel_{tag}"]).max() <= 2e-3


def test_training_mel_batch_and_long_clip_vs_oracle():
    from oracle import rvc_oracle as O
    from rvc_amd.train import mel_processing as MP
    rng = np.random.default_rng(1)
    y = torch.from_numpy((0.3 * rng.standard_normal((3, 48000 * 4))).astype(np.float32))      # 3 x 4 s at 48 kHz
    mel = MP.mel_spectrogram_torch(y.to(DEV), 2048, 128, 48000, 480, 2048, 0.0, None)
    ref = O.mel_spectrogram(y, 2048, 128, 48000, 480, 2048)
    assert mel.shape == ref.shape == (3, 128, 400)
    assert (mel.cpu() - ref).abs().max().item() <= 2e-3


@pytest.mark.parametrize("sr_in,sr_out", [(48000, 16000), (44100, 16000), (22050, 16000), (16000, 48000), (32000, 16000)])
def test_device_resampler_equals_scipy_polyphase_with_the_same_filter(sr_in, sr_out):
    import math
    from scipy import signal
    from rvc_amd.lib import audio as A
    rng = np.random.default_rng(sr_in)
    n = 50_001
    t = np.arange(n) / sr_in
    x = 0.5 * np.sin(2 * np.pi * 997.0 * t) + 0.1 * rng.standard_normal(n)
    g = math.gcd(sr_in, sr_out)
    up, down = sr_out // g, sr_in // g
    y = A.resample(x, sr_in, sr_out, device=DEV)
    ref = signal.resample_poly(x, up, down, window=A.resample_filter(up, down) / up)   # SciPy rescales its window by `up`
    assert y.shape == ref.shape == (math.ceil(n * up / down),)
    assert np.abs(y - ref).max() <= 1e-12
    # and it is a good resampler: a 997 Hz tone comes out as a 997 Hz tone, residual below -120 dB
    clean = A.resample(0.5 * np.sin(2 * np.pi * 997.0 * t), sr_in, sr_out, device=DEV)
    tt = np.arange(clean.shape[0]) / sr_out
    mid = slice(2000, -2000)
    assert rms((clean - 0.5 * np.sin(2 * np.pi * 997.0 * tt))[mid]) <= 0.35 * 10 ** (-120 / 20)


def test_extraction_module_matches_oracle(tmp_path):
    """rvc/train/extract/extract.py: f0 / coarse f0 / HuBERT features per file, files strided over the devices."""
    from oracle import rvc_oracle as O
    from rvc_amd.infer.infer import _write_wav
    from rvc_amd.lib import synthetic as S
    from rvc_amd.train.extract import extract as E
    hub_sd, rm_sd = S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0)
    files = []
    for i, n in enumerate((16000, 24000, 20000)):
        wav = os.path.join(tmp_path, f"u{i}.wav")
        _write_wav(wav, S.synth_audio(n, seed=i), 16000)
        files.append((wav, os.path.join(tmp_path, f"u{i}_f0c.npy"), os.path.join(tmp_path, f"u{i}_f0f.npy"),
                      os.path.join(tmp_path, f"u{i}_emb.npy")))
    assert E._stride(list(range(7)), ["a", "b", "c"]) == [[0, 3, 6], [1, 4], [2, 5]]      # files[i::len(devices)]
    E.run_pitch_extraction(files, [DEV], "rmvpe", 160, rmvpe_state_dict=rm_sd)
    E.run_embedding_extraction(files, [DEV], "contentvec", None, hubert_state_dict=hub_sd)
    from rvc_amd.lib.audio import load_audio
    for wav, f0c, f0f, emb in files:
        audio = load_audio(wav, 16000)
        ref_f0 = O.rmvpe_infer_from_audio(audio, rm_sd)
        got_f0, got_c = np.load(f0f), np.load(f0c)
        assert got_f0.shape == ref_f0.shape and got_c.dtype.kind == "i"
        same = np.abs(got_f0 - ref_f0) <= 1e-3 * np.maximum(ref_f0, 1)
        assert same.mean() >= 0.98                                   # arg-max near-ties of the random-weight salience
        assert np.array_equal(got_c, O.extract_coarse_f0(got_f0))    # the quantiser itself is bit-exact
        ref_emb = O.hubert_forward(hub_sd, torch.from_numpy(audio).float().view(1, -1))[0].numpy()
        assert np.load(emb).shape == ref_emb.shape and np.abs(np.load(emb) - ref_emb).max() <= 1e-3


def test_convert_audio_reads_any_wav(tmp_path, monkeypatch):
    """a18 / §8f rank 3: convert_audio on a 44.1 kHz stereo float32 WAV (infer.py:257-260 -> load_audio_infer): folded
    to mono, resampled to 16 kHz on the device, converted; equals convert_array on the same 16 kHz signal."""
    import struct
    import wave
    from rvc_amd.infer.infer import VoiceConverter
    from rvc_amd.lib import audio as A
    from rvc_amd.lib import synthetic as S
    monkeypatch.chdir(tmp_path)
    os.makedirs("rvc/models/embedders/contentvec"); os.makedirs("rvc/models/predictors")
    torch.save(S.make_hubert_state_dict(1), "rvc/models/embedders/contentvec/pytorch_model.bin")
    torch.save(S.make_rmvpe_state_dict(0), "rvc/models/predictors/rmvpe.pt")
    torch.save(S.make_synth_checkpoint(40000, "HiFi-GAN", seed=0, half=True), "model.pth")
    n = 44100
    t = np.arange(n) / 44100.0
    stereo = np.stack([0.3 * np.sin(2 * np.pi * 200 * t), 0.2 * np.sin(2 * np.pi * 310 * t)], 1).astype("<f4")
    fmt = struct.pack("<HHIIHH", 3, 2, 44100, 44100 * 8, 8, 32)
    body = b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", stereo.nbytes) + stereo.tobytes()
    open("in.wav", "wb").write(b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body)
    vc = VoiceConverter(device=DEV)
    vc.convert_audio("in.wav", "out.wav", "model.pth", "", index_rate=0.0)
    with wave.open("out.wav", "rb") as f:
        assert f.getframerate() == 40000
        got = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").astype(np.float64) / 32767.0
    mono16k = A.load_audio_infer("in.wav", 16000, device=DEV)
    assert mono16k.shape == (16000,)
    assert np.abs(mono16k - A.resample(stereo.astype(np.float64).mean(1), 44100, 16000, device=DEV)).max() == 0
    n_pad = 16000 + 32000
    assert got.shape[0] == min(n_pad // 160, 2 * ((n_pad - 400) // 320 + 1)) * 400 - 2 * 40000
