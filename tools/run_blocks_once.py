#!/usr/bin/env python3
"""Three launches each of the two block-forward kernels at their benchmark shapes (flash forward 4096 x 512, chunked
linear attention 512 x 4096; hd = dv = 128) -- the command tools/prof_blocks.sh profiles."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
g = torch.Generator(device="cuda"); g.manual_seed(1)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("both", "flash"):
    B, L, H, hd, dv = 4096, 512, 1, 128, 128
    qk = torch.randn((B, L, 2, H, hd), generator=g, device="cuda") / hd ** 0.25
    v = torch.randn((B, L, H * dv), generator=g, device="cuda")
    for _ in range(3):
        y = ops.softmax_attention_fwd(qk[:, :, 0], qk[:, :, 1], v)
    torch.cuda.synchronize(); del qk, v, y
if which in ("both", "lin"):
    B, L, d, C = 512, 4096, 128, 128
    qk = torch.randn((B, L, 2, 1, d), generator=g, device="cuda") / d ** 0.5
    v = torch.randn((B, L, C), generator=g, device="cuda")
    for _ in range(3):
        y = ops.linear_recurrence_fwd(qk[:, :, 0], qk[:, :, 1], v, phi=True, normalise=True)
    torch.cuda.synchronize()
