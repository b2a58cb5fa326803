"""bench.py's own multi-rank launcher, driven on CPU: `python bench.py --gpus N` must start N ranks itself (the
reference's extract.py:141-152 starts its per-device workers the same way), and must never report an N-GPU line from
fewer ranks or devices.  --control-flow-only swaps the kernels out and runs the process group over gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_self_launches_n_ranks(world):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--steps", "3", "--warmup", "1", "--control-flow-only"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0's line only
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["ranks"] == world          # the group RCCL/gloo reports == --gpus
    assert line["index_broadcast"]["n_ranks"] == world
    assert line["total_samples"] == 3 * world * 1000                   # every rank converted its i mod N share
    assert abs(line["t_max"] - (1.0 + 0.001 * (world - 1))) < 1e-9     # MAX over ranks
    # the fields that make a first real N-GPU run self-diagnosing: per-rank spread of the timed region, the broadcast's rate
    spread = line["ms_per_step_ranks"]
    assert len(spread["all"]) == world and spread["min"] <= spread["max"] and abs(spread["max"] - line["t_max"] / 3 * 1e3) < 0.01
    assert line["index_broadcast"]["bytes"] == 256 * 768 * 4 and line["index_broadcast"]["gbps"] is not None


def test_bench_refuses_more_gpus_than_visible():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_world_size_mismatch():
    """A launcher that exported WORLD_SIZE=1 while the command says --gpus 2 (round 1 silently printed n_gpus: 1)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--control-flow-only"],
                       env=_clean_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE is 1" in r.stderr


def test_spawn_ranks_stops_the_job_when_a_rank_dies(tmp_path):
    sys.path[:0] = [REPO, os.path.join(REPO, "codename-rvc-fork-3_amd")]
    from rvc_amd.infer.distributed import spawn_ranks
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK']); assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
                      "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "if r == 1: sys.exit(7)\n"
                      "time.sleep(60)\n")
    import time
    t0 = time.time()
    rc = spawn_ranks([sys.executable, str(script)], 3)
    assert rc == 7 and time.time() - t0 < 30      # ranks 0 and 2 were terminated, not waited for


def test_checksum_formula_host():
    """tensor_checksum on host tensors: the documented formula of rvc_checksum64 (the GPU test compares the kernel with it)."""
    sys.path[:0] = [REPO, os.path.join(REPO, "codename-rvc-fork-3_amd")]
    from rvc_amd.infer.distributed import tensor_checksum
    t = torch.arange(7, dtype=torch.int32)
    s1, s2 = tensor_checksum(t)
    assert s1 == 21 and s2 == sum((i + 1) * i for i in range(7))
    big = torch.full((1 << 20,), -1, dtype=torch.int32)                # words 0xFFFFFFFF: the sums wrap mod 2^64
    s1, s2 = tensor_checksum(big)
    n, w = 1 << 20, 0xFFFFFFFF
    assert s1 == (n * w) % (1 << 64) and s2 == (w * n * (n + 1) // 2) % (1 << 64)
    assert tensor_checksum(torch.tensor([1.0, 2.0])) != tensor_checksum(torch.tensor([2.0, 1.0]))   # order matters
