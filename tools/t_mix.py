import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch, bench
from rvc_amd import _native
dev = "cuda:0"
for T, rates in ((1198, [10, 10, 2, 2]), (3198, [12, 10, 2, 2])):
    run, fl, n, ab, ex = bench.roofline_mix(torch, _native, dev, T, rates, "f32")
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(T, rates, e0.elapsed_time(e1) / 5 / n * 1e3, "us per launch")
