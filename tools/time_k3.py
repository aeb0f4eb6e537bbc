#!/usr/bin/env python3
"""K3 (linear-attention eta) whole-op timing at the C4/8 shard size (512 x 4096 tokens, D = 128, d_qk = 128)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
torch.manual_seed(0)
B, L, D, dqk = 512, 4096, 128, 128
x = torch.randn((B, L, D), device="cuda") * 0.5
w = torch.randn((2 * dqk, D), device="cuda") / D ** 0.5
b = torch.randn((2 * dqk,), device="cuda") * 0.1
out = {}
for heads in (1, 2):
    for tag, env in (("two_wg", None), ("one_wg", "1")):
        if env: os.environ["EGS_LINATTN_ONE_WG"] = env
        e = ops.linattn_eta(x, w, b, dqk, heads)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            ev0.record(); e = ops.linattn_eta(x, w, b, dqk, heads); ev1.record(); ev1.synchronize()
            ts.append(ev0.elapsed_time(ev1))
        out[f"ms_h{heads}_{tag}"] = sorted(ts)[2]
        if env: del os.environ["EGS_LINATTN_ONE_WG"]
print(json.dumps(out, indent=1))
