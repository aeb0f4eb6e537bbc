#!/usr/bin/env python3
"""Whole-op timings (HIP events, median of 5): K3 linear-attention eta at the C4/8 shard size, fused and unfused
(EGS_LINATTN_UNFUSED=1: projection kernel + scan), and K4 softmax eta at 4096 x 512 tokens, head_dim 64 / 128."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
torch.manual_seed(0)
out = {}
def med(fn):
    fn(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(25):
        ev0.record(); fn(); ev1.record(); ev1.synchronize(); ts.append(ev0.elapsed_time(ev1))
    return [min(ts), sorted(ts)[12]]
B, L, D, dqk = 512, 4096, 128, 128
x = torch.randn((B, L, D), device="cuda") * 0.5
w = torch.randn((2 * dqk, D), device="cuda") / D ** 0.5
b = torch.randn((2 * dqk,), device="cuda") * 0.1
for heads in (1, 2):
    for tag, env in (("fused", None), ("unfused", "1")):
        if env: os.environ["EGS_LINATTN_UNFUSED"] = env
        out[f"linattn_ms_h{heads}_{tag}"] = med(lambda: ops.linattn_eta(x, w, b, dqk, heads))
        if env: del os.environ["EGS_LINATTN_UNFUSED"]
del x
B, L, D = 4096, 512, 128
x = torch.randn((B, L, D), device="cuda") * 0.5
for hd, heads in ((64, 2), (128, 1)):
    dqk = hd * heads
    w = torch.randn((2 * dqk, D), device="cuda") / D ** 0.5
    b = torch.randn((2 * dqk,), device="cuda") * 0.1
    out[f"softmax_ms_hd{hd}"] = med(lambda: ops.softmax_eta(x, w, b, dqk, heads))
print(json.dumps(out, indent=1))
