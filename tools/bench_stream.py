#!/usr/bin/env python3
"""Times the streaming kernels (K2 Mamba-2 lambda_t, K3 linear-attention eta, norm-attention eta, histograms) at the
BASELINE.json config sizes on one GPU and prints achieved algorithmic HBM GB/s (SURVEY.md 8(d) byte counts)."""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "task-level-insights-from-eigenvalues-across-sequence-models_amd"


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        fn()
        ev[r + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="k2,norm,k3,k4,hist")
    ap.add_argument("--scale", type=float, default=1.0, help="batch scale factor")
    args = ap.parse_args()
    ops = importlib.import_module(PKG).ops
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(16)
    out = {}
    which = args.which.split(",")
    if "k2" in which:          # C3: Mamba d_state=16, L=2048, B=1024, D=128, nheads=4
        B, L, D, nh = int(1024 * args.scale), 2048, 128, 4
        x = torch.randn((B, L, D), generator=g, device=dev)
        w = (torch.rand((nh, D), generator=g, device=dev) * 2 - 1) / np.sqrt(D)
        dtb = torch.randn(nh, generator=g, device=dev)
        alog = torch.log(torch.rand(nh, generator=g, device=dev) * 15 + 1)
        ms = timeit(lambda: ops.mamba2_lambda(x, w, None, dtb, alog))
        byts = B * L * (4 * D + 4 * nh)
        out["k2_mamba2"] = {"ms": ms, "GBs": byts / ms / 1e6, "bytes": byts, "tokens_per_s": B * L / ms * 1e3}
        del x
    if "norm" in which:
        B, L, D, H = int(1024 * args.scale), 2048, 128, 4
        x = torch.randn((B, L, D), generator=g, device=dev)
        w = (torch.rand((H, D), generator=g, device=dev) * 2 - 1) / np.sqrt(D)
        ms = timeit(lambda: ops.normattn_eta(x, w, None, None, "softplus", H))
        byts = B * L * (4 * D + 8 * H)
        out["norm_attn"] = {"ms": ms, "GBs": byts / ms / 1e6, "bytes": byts}
        del x
    if "k3" in which:          # C4 per GPU: lin-attn D=128, L=4096, B=4096/8, H=1, d_qk=128
        B, L, D, dqk, H = int(512 * args.scale), 4096, 128, 128, 1
        x = torch.randn((B, L, D), generator=g, device=dev)
        w = (torch.rand((2 * dqk, D), generator=g, device=dev) * 2 - 1) / np.sqrt(D)
        b = (torch.rand((2 * dqk,), generator=g, device=dev) * 2 - 1) / np.sqrt(D)
        ms = timeit(lambda: ops.linattn_eta(x, w, b, dqk, H), reps=3, warm=1)
        byts = B * L * (4 * D + 8 * H)
        flops = B * L * 2.0 * D * 2 * dqk
        out["k3_linattn"] = {"ms": ms, "GBs": byts / ms / 1e6, "bytes": byts, "proj_TFLOPs": flops / ms / 1e9,
                             "eta_per_s": B * (L - 1) * H / ms * 1e3}
        del x
    if "k4" in which:          # softmax attention at the reference's MQAR scale: B=64, L=512, D=128, d_qk=128, H=1
        B, L, D, dqk, H = 64, 512, 128, 128, 1
        x = torch.randn((B, L, D), generator=g, device=dev)
        w = (torch.rand((2 * dqk, D), generator=g, device=dev) * 2 - 1) / np.sqrt(D)
        b = (torch.rand((2 * dqk,), generator=g, device=dev) * 2 - 1) / np.sqrt(D)
        ms = timeit(lambda: ops.softmax_eta(x, w, b, dqk, H), reps=3, warm=1)
        flops = 2.0 * B * H * (L * L / 2) * (dqk // H) * 2       # two passes over the causal half
        out["k4_softmax"] = {"ms": ms, "score_TFLOPs": flops / ms / 1e9, "eta_per_s": B * (L - 1) * H / ms * 1e3}
        del x
    if "hist" in which:
        B, T, H, Ly = 4096, 4095, 1, 2
        v = torch.rand((B, T, H, Ly), generator=g, device=dev, dtype=torch.float64) * 2
        ms = timeit(lambda: ops.threshold_hist(v, [0.1, 0.5, 0.9, 1.0, 10, 100], "value"))
        byts = v.numel() * 8
        out["hist_f64"] = {"ms": ms, "GBs": byts / ms / 1e6, "bytes": byts}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
