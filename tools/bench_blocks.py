#!/usr/bin/env python3
"""Timing of the block-forward kernels (linear recurrence) at the reference's shapes; prints JSON lines."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd")
ops = pkg.ops


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    cases = [
        # name, B, L, G, d, cpg, cpd, decay
        ("lin-attn MQAR 64x512 D=128 d_qk=128 H=1", 64, 512, 1, 128, 128, 128, False),
        ("lin-attn C4/8 512x4096 D=128 d_qk=128 H=1", 512, 4096, 1, 128, 128, 128, False),
        ("lin-attn 4096x512 D=128 d_qk=128 H=1", 4096, 512, 1, 128, 128, 128, False),
        ("mamba2 C3-like 1024x2048 D=128 N=16 4 heads", 1024, 2048, 1, 16, 128, 32, True),
        ("mamba2 512x1024 D=256 N=64 8 heads", 512, 1024, 1, 64, 256, 32, True),
    ]
    sm_cases = [
        # name, B, L, H, hd, dv
        ("softmax-attn MQAR 64x512 D=128 d_qk=128 H=1", 64, 512, 1, 128, 128),
        ("softmax-attn 4096x512 D=128 d_qk=128 H=1", 4096, 512, 1, 128, 128),
        ("softmax-attn 64x1024 D=512 H=8 hd=64", 64, 1024, 8, 64, 64),
        ("softmax-attn 64x1024 D=512 H=2 hd=256", 64, 1024, 2, 256, 256),
    ]
    if len(sys.argv) > 1:
        cases = [c for c in cases if sys.argv[1] in c[0]]
        sm_cases = [c for c in sm_cases if sys.argv[1] in c[0]]
    for name, B, L, H, hd, dv in sm_cases:
        qk = torch.randn(B, L, 2, H, hd, device="cuda")
        v = torch.randn(B, L, H * dv, device="cuda")
        ms = timeit(lambda: ops.softmax_attention_fwd(qk[:, :, 0], qk[:, :, 1], v))
        flops = B * H * L * (L + 1) / 2 * (hd + dv) * 2
        print(json.dumps({"case": name, "ms": ms, "tflops_causal": flops / ms / 1e9}), flush=True)
    for name, B, L, G, d, cpg, cpd, decay in cases:
        C = G * cpg
        q = torch.randn(B, L, G, d, device="cuda") / d ** 0.5
        k = torch.randn(B, L, G, d, device="cuda") / d ** 0.5
        v = torch.randn(B, L, C, device="cuda")
        a = torch.rand(B, L, C // cpd, device="cuda") if decay else None
        s = torch.rand(B, L, C // cpd, device="cuda") if decay else None
        ms = timeit(lambda: ops.linear_recurrence_fwd(q, k, v, decay=a, vscale=s, phi=not decay,
                                                      normalise=not decay, cols_per_decay=cpd))
        flops = B * L * d * C * (6 if decay else 4)
        byts = B * L * (2 * G * d + 2 * C) * 4
        print(json.dumps({"case": name, "ms": ms, "tflops": flops / ms / 1e9, "algorithmic_gbs": byts / ms / 1e6}),
              flush=True)


if __name__ == "__main__":
    main()
