import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "codename-rvc-fork-3_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rms(x):
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt(np.mean(x * x)))


# ---- tie-aware f0 comparison (flat-salience synthetic RMVPE): shared by test_fullsize_gpu.py and test_pipeline_gpu.py --------
SAL_TIE = 2.5e-4   # salience near-tie bound: ~4x the largest GPU-vs-CPU salience difference measured (6e-5, tools/diag_rmvpe.py)
F0_NOISE = 2e-5    # relative f0 difference that identical arg-max bins produce (measured ~1e-6: the 9-bin weighted mean
                   # moves with the salience's 1e-5-level differences)


def f0_tie_report(f0_p, sal_p, f0_o, sal_o):
    """Frames where the product's RMVPE contour differs from the oracle's by more than fp noise, each CERTIFIED as a
    near-tie of the salience arg-max (RMVPE.py:459-512 picks the arg-max bin, then averages +-4 bins around it): the bin
    the product chose must be within SAL_TIE of the oracle's maximum IN THE ORACLE'S OWN salience, and the two saliences
    must agree to SAL_TIE everywhere on that frame.  The synthetic (random-weight) RMVPE has a flat, noise-like
    salience -- median top-2 gap 2e-3, minimum ~1e-6 over 3200 frames -- so two fp32 evaluations of the same network
    legitimately pick different bins on a frame now and then.  Returns the differing frame indices (all certified)."""
    n = min(len(f0_p), len(f0_o))
    f0_p, f0_o, sal_p, sal_o = f0_p[:n], f0_o[:n], sal_p[:n], sal_o[:n]
    assert np.abs(sal_p - sal_o).max() <= SAL_TIE, np.abs(sal_p - sal_o).max()
    differ = np.nonzero(np.abs(f0_p - f0_o) > F0_NOISE * np.maximum(f0_o, 1.0))[0]
    for t in differ:
        bp, bo = int(sal_p[t].argmax()), int(sal_o[t].argmax())
        if bp == bo:          # same bin, so the voicing decision differs: max salience within SAL_TIE of the 0.03 threshold
            assert abs(sal_o[t].max() - 0.03) <= SAL_TIE, (t, f0_p[t], f0_o[t], sal_o[t].max())
            continue
        assert sal_o[t, bo] - sal_o[t, bp] <= SAL_TIE, \
            f"frame {t}: product bin {bp} vs oracle bin {bo} is not a salience near-tie ({sal_o[t, bo] - sal_o[t, bp]:.2e})"
    return differ



def vs_oracle(vc, hubert, S, cpt, audio, *, seed, sid=0, pitch=0, big=None, index_rate=0.0, protect=0.5, gate=1e-3,
              volume_envelope=1, f0_autotune=False, f0_autotune_strength=1, oracle_kw=None, label="", version="v2", reference=None):
    """Product vs oracle (flat-salience synthetic RMVPE) under one seed, TIE-AWARE in f0 like test_fullsize_gpu.py: two correct
    fp32 evaluations of the random-weight RMVPE can pick different salience bins on a frame now and then (the GPU contour itself
    moves by ~1e-7 from run to run: DESIGN.md section 2), and one such frame moves the NSF source's phase for the rest of the clip
    (seen once as 1.1e-2 on the 2 s 40 k sweep case that otherwise agrees to 2.5e-6).  When the plain comparison misses the gate,
    every frame on which the two raw contours differ must be a CERTIFIED near-tie (f0_tie_report: salience arg-max, or the 0.03
    voicing threshold; coarse-pitch flips: a .5 rounding boundary) -- and there must be one, or the miss is a real error -- and the
    oracle is re-run on the product's contour, so that everything downstream is still compared at the gate.
    `reference`: the REFERENCE's own output for this input (a tests/golden fixture).  The product is then compared with it directly;
    only on a miss is the oracle evaluated -- it must reproduce the reference's output (2e-5: the pin) -- and the treatment above
    applied.  Returns (product output, what it was finally compared with, rms error)."""
    import torch
    from oracle import rvc_oracle as O
    okw = dict(sid=sid, pitch=pitch, big_npy=big, index_rate=index_rate, protect=protect, volume_envelope=volume_envelope,
               f0_autotune=f0_autotune, f0_autotune_strength=f0_autotune_strength, **(oracle_kw or {}))
    hub_sd, rm_sd = S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0)
    taps_o = {}

    def oracle(**extra):
        torch.manual_seed(seed)
        return O.pipeline(hub_sd, rm_sd, cpt, np.array(audio, copy=True), **okw, **extra)

    want = oracle(taps=taps_o) if reference is None else np.asarray(reference)
    vc.vc.debug_taps = {}
    try:
        got = vc.vc.pipeline(hubert, vc.net_g, sid, np.array(audio, copy=True), pitch, "rmvpe", "", index_rate, True, 3, volume_envelope, version,
                             protect, 128, f0_autotune, f0_autotune_strength, None, noise_seed=seed)
        f0_p = vc.vc.debug_taps["f0_raw"].cpu().numpy()
        sal_p = vc.vc.debug_taps["salience"].cpu().numpy()
    finally:
        vc.vc.debug_taps = None
    assert got.shape == want.shape and got.dtype == np.float32
    err = rms(got - want)
    if err > gate:
        if reference is not None:
            pinned = oracle(taps=taps_o)
            assert rms(pinned - want) <= 2e-5, f"{label}: the oracle does not reproduce the reference's output ({rms(pinned - want):.2e})"
        f0_o, sal_o = taps_o["f0_raw"], taps_o["salience"]
        differ = f0_tie_report(f0_p, sal_p, f0_o, sal_o)            # asserts that every differing frame is a certified near-tie
        n_f = min(len(f0_p), len(f0_o))
        c_p = O.f0_to_coarse(f0_p[:n_f].astype(np.float64), pitch, f0_autotune, f0_autotune_strength)[0]
        c_o = O.f0_to_coarse(f0_o[:n_f].astype(np.float64), pitch, f0_autotune, f0_autotune_strength)[0]
        flips = np.nonzero(c_p != c_o)[0]
        assert len(differ) + len(flips) > 0, f"{label}: rms err {err:.3e} with identical f0 contours"
        assert len(differ) + len(flips) <= max(2, n_f // 100), (len(differ), len(flips), n_f)
        want = oracle(f0_override=f0_p)
        print(f"{label}: plain rms err {err:.3e} explained by {len(differ)} certified salience near-tie frame(s) / {len(flips)} coarse-pitch "
              f"flip(s) of {n_f}; the oracle follows the product's contour")
        err = rms(got - want)
    return got, want, err


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def ref_inputs():
    """The reference's own realistic Synthesizer inputs (logs/reference/ref_*.npy, train.py:839-842)."""
    return (np.load(os.path.join(GOLDEN, "ref_feats.npy")), np.load(os.path.join(GOLDEN, "ref_f0c.npy")),
            np.load(os.path.join(GOLDEN, "ref_f0f.npy")))


# ------------------------------------------------------------------------------------------------------------------
# The full-length oracle legs run in a host process pool (tests/_oracle_farm.py) that starts with the session and works
# underneath the kernel / pipeline tests; tests/test_fullsize_gpu.py runs last and reads the results.
# ------------------------------------------------------------------------------------------------------------------

_FARM = None


def _farm_keys(items):
    """The farm jobs the selected test_fullsize_gpu.py items will ask for."""
    keys = set()
    for it in items:
        if "test_fullsize_gpu" not in it.nodeid:
            continue
        name = it.name
        case = name[name.index("[") + 1:-1] if "[" in name else ""
        if name.startswith("test_full_length_plain_peaked_rmvpe"):
            keys.add("peaked:" + case)
        elif name.startswith("test_baseline_config_full_length_vs_oracle"):
            keys.add("flat:cfg" + case)
        elif name.startswith("test_45s_two_segments_vs_oracle"):
            keys.add("flat:45s")
        elif name.startswith("test_decoder_T3198_stage_by_stage_vs_oracle"):
            keys.add("decoder:" + case.replace("-hint1", "").replace("-hint2", ""))
    return keys


def pytest_collection_modifyitems(config, items):
    late = [it for it in items if "test_fullsize_gpu" in it.nodeid]
    if late:
        items[:] = [it for it in items if "test_fullsize_gpu" not in it.nodeid] + late


def pytest_collection_finish(session):
    global _FARM
    keys = _farm_keys(session.items)
    if not keys or session.config.option.collectonly:
        return
    import torch
    if not torch.cuda.is_available():
        return
    from _oracle_farm import Farm, draw_index_on_device
    _FARM = Farm()
    # the pytest process computes float64 / oracle references of its own on the host: keep it to a quarter of the cores while
    # the farm's workers (6 x 32 threads) are busy, instead of torch's default of one thread per core on top of them
    torch.set_num_threads(max(8, min(64, (os.cpu_count() or 8) // 4)))
    index_file = None
    if "peaked:cfg5" in keys:        # cfg 5's 2 M rows are drawn on the device as bench.py draws them; the worker maps the host copy
        index_file = os.path.join(_FARM.dir, "cfg5_index.npy")
        np.save(index_file, draw_index_on_device(2_000_000, "cuda:0").cpu().numpy())
        torch.cuda.empty_cache()
    _FARM.cfg5_index_file = index_file
    _FARM.submit_all(index_file, which=keys)
    print(f"\n[oracle farm] {len(_FARM.futures)} oracle jobs on {_FARM.workers} workers x {_FARM.threads} threads, results in {_FARM.dir}")


def pytest_sessionfinish(session, exitstatus):
    global _FARM
    if _FARM is not None:
        _FARM.close()
        _FARM = None


@pytest.fixture(scope="session")
def oracle_farm():
    if _FARM is None:
        pytest.skip("oracle farm not started (no GPU)")
    return _FARM
