#!/usr/bin/env python3
"""Times egs_hseqr_batched (the n = 64 QR kernel alone) on Hessenberg forms of the benchmark's matrices.
   NMAT=65536 python tools/time_hseqr.py   (EGS_HQR_LDS_PAD=bytes lowers the workgroups per CU)"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
nmat = int(os.environ.get("NMAT", 65536))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(64)
a = torch.empty((nmat, 64, 64), dtype=torch.float64, device=dev)
for m0 in range(0, nmat, 16384):
    cnt = min(16384, nmat - m0)
    a[m0:m0 + cnt] = bench.gen_known_spectrum(cnt, 64, gen, dev)[0]
h = ops.hessenberg_batched(a)
ops.hseqr_batched(h); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record()
for r in range(3):
    w, info = ops.hseqr_batched(h); ev[r + 1].record()
torch.cuda.synchronize()
ms = [ev[r].elapsed_time(ev[r + 1]) for r in range(3)]
print(json.dumps({"nmat": nmat, "pad": os.environ.get("EGS_HQR_LDS_PAD"), "ms": float(np.median(ms)), "bad": int((info != 0).sum().item())}))
