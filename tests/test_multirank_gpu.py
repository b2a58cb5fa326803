"""SURVEY 8(e) "Verification" on the one GPU a test box has: a 2-rank job (two processes sharing cuda:0, process group over
gloo -- RCCL refuses two ranks on one device) must produce, per utterance, what the 1-rank job produces, and every rank
must hold the root's index bytes.  Precedent for the sharding: rvc/train/extract/extract.py:141-152, 194-207
(one worker per device, files[i::n]).  The RCCL transport itself needs >= 2 GPUs and is the driver's to run."""
import os
import sys

import numpy as np
import pytest

from conftest import rms

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(REPO, "tests", "_ranks_worker.py")
N_UTT = 8      # both tests: the 2-rank job takes four each, the 4-rank job two each; ONE 1-rank job serves as the expected value of both


def _run(world, out_dir, n_utt=N_UTT):
    from rvc_amd.infer.distributed import spawn_ranks
    env = {"OUT_DIR": str(out_dir), "N_UTT": str(n_utt), "UTT_SECONDS": "3.0"}
    if world > 1:
        env["RVC_DIST_BACKEND"] = "gloo"
    rc = spawn_ranks([sys.executable, WORKER], world, env_extra=env, timeout=900)
    assert rc == 0, f"{world}-rank job exited with {rc}"
    return [np.load(os.path.join(out_dir, f"rank{r}_of{world}.npz")) for r in range(world)]


@pytest.fixture(scope="module")
def one_rank(tmp_path_factory):
    (one,) = _run(1, tmp_path_factory.mktemp("one_rank"))
    return one


def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path, one_rank):
    one = one_rank
    two = _run(2, tmp_path)
    # every rank holds the root's bytes: device-side checksums equal each other and the 1-rank job's
    assert all(bool(r["agree"]) for r in two)
    assert np.array_equal(two[0]["checksum"], two[1]["checksum"]) and np.array_equal(two[0]["checksum"], one["checksum"])
    assert "gloo" in str(two[0]["transport"])
    # utterance i went to rank i mod 2, and each waveform equals the 1-rank run's (hipBLASLt GEMMs are not bit-reproducible
    # between processes: ~1e-6 RMS, tools/diag_determinism.py; gate 1e-5)
    worst = 0.0
    for i in range(N_UTT):
        owner, other = two[i % 2], two[1 - i % 2]
        assert f"utt{i}" in owner.files and f"utt{i}" not in other.files
        a, b = owner[f"utt{i}"], one[f"utt{i}"]
        assert a.shape == b.shape and a.dtype == np.float32
        worst = max(worst, rms(a - b))
    print(f"2 ranks on one GPU vs 1 rank: worst per-utterance waveform rms difference {worst:.2e} (gate 1e-5)")
    assert worst <= 1e-5
    # report reduction: SUM of samples, MAX of seconds
    n_total = sum(one[f"utt{i}"].shape[0] for i in range(N_UTT))
    assert int(two[0]["total"]) == int(two[1]["total"]) == int(one["total"]) == n_total
    assert float(two[0]["t_max"]) == 2.0 and float(one["t_max"]) == 1.0


def test_four_ranks_strided_batch_equals_one_rank(tmp_path, one_rank):
    """BASELINE cfg 3's sharding at the scale one GPU allows: a batch of 8 utterances over 4 ranks (utterance i -> rank i mod 4,
    two per rank -- cfg 3 is 512 over 8, 64 per rank, the same striding), four processes sharing cuda:0 over gloo.  Every
    utterance must come out of exactly one rank and equal the 1-rank job's; all four ranks hold the root's index bytes."""
    n = N_UTT
    one = one_rank
    four = _run(4, tmp_path, n)
    assert all(bool(r["agree"]) for r in four)
    assert all(np.array_equal(r["checksum"], one["checksum"]) for r in four)
    worst = 0.0
    for i in range(n):
        holders = [r for r in range(4) if f"utt{i}" in four[r].files]
        assert holders == [i % 4], (i, holders)
        a, b = four[i % 4][f"utt{i}"], one[f"utt{i}"]
        assert a.shape == b.shape and a.dtype == np.float32
        worst = max(worst, rms(a - b))
    print(f"4 ranks on one GPU, 8 utterances vs 1 rank: worst per-utterance waveform rms difference {worst:.2e} (gate 1e-5)")
    assert worst <= 1e-5
    n_total = sum(one[f"utt{i}"].shape[0] for i in range(n))
    assert all(int(r["total"]) == n_total for r in four) and float(four[0]["t_max"]) == 4.0


def test_bench_control_flow_two_ranks_on_the_gpu_box():
    """bench.py's own launcher on the GPU box (verdict round 5, item 9: keep the N-rank path warm where one GPU is all there is):
    `--gpus 2 --control-flow-only` starts two ranks, runs the process group / striding / broadcast / checksum / report
    reduction over gloo and prints ONE line carrying the per-rank spread and the broadcast's rate."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--control-flow-only"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    (line,) = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["index_broadcast"]["n_ranks"] == 2
    assert len(line["ms_per_step_ranks"]["all"]) == 2 and line["index_broadcast"]["gbps"] is not None
