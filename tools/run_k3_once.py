#!/usr/bin/env python3
"""Three launches of the fused linear-attention eta (K3) at the C4/8 shard shape and at the MQAR batch shape -- the command
tools/prof_k3.sh profiles."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
torch.manual_seed(0)
for (B, L, D, dqk) in ((512, 4096, 128, 128), (4096, 512, 128, 64)):
    x = torch.randn((B, L, D), device="cuda") * 0.5
    w = torch.randn((2 * dqk, D), device="cuda") / D ** 0.5
    b = torch.randn((2 * dqk,), device="cuda") * 0.1
    for _ in range(3):
        e = ops.linattn_eta(x, w, b, dqk, 1)
    torch.cuda.synchronize()
    del x, e
