#!/usr/bin/env python3
"""Does the register-resident Hessenberg kernel run beside the n = 64 QR kernel on the same CUs?
Times, on the benchmark's matrices: the fused kernel; the two kernels back to back on one stream; the two kernels on two
streams, chunk c+1's reduction queued behind chunk c's so that it overlaps chunk c's QR sweep.
   NMAT=131072 CHUNK=16384 python tools/overlap_probe.py      (EGS_LIB_PATH selects the library)"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd")
ops = pkg.ops
nmat = int(os.environ.get("NMAT", 131072))
chunk = int(os.environ.get("CHUNK", 16384))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(64)
a = torch.empty((nmat, 64, 64), dtype=torch.float64, device=dev)
for m0 in range(0, nmat, 16384):
    cnt = min(16384, nmat - m0)
    a[m0:m0 + cnt] = bench.gen_known_spectrum(cnt, 64, gen, dev)[0]
lib = pkg._capi.lib()
h = torch.empty_like(a)
w = torch.empty((nmat, 64), dtype=torch.complex128, device=dev)
info = torch.empty((nmat,), dtype=torch.int32, device=dev)
nch = (nmat + chunk - 1) // chunk
queues = torch.zeros((nch, 32), dtype=torch.int64, device=dev)      # one work-queue counter (256 B) per chunk


def hess(c, stream):
    lo, cnt = c * chunk, min(chunk, nmat - c * chunk)
    pkg._capi.check(lib.egs_hessenberg_batched(a[lo].data_ptr(), cnt, 64, 64, 4096, h[lo].data_ptr(), stream.cuda_stream), "hess")


def qr(c, stream):
    lo, cnt = c * chunk, min(chunk, nmat - c * chunk)
    pkg._capi.check(lib.egs_hseqr_batched(h[lo].data_ptr(), cnt, 64, w[lo].data_ptr(), info[lo].data_ptr(),
                                    queues[c].data_ptr(), stream.cuda_stream), "hseqr")


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return float(np.median(out))


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fused():
    ops.eig_general_batched(a)


def serial():
    for c in range(nch):
        hess(c, s0); qr(c, s0)


def only_qr():
    for c in range(nch):
        qr(c, s0)


def only_hess():
    for c in range(nch):
        hess(c, s0)


def overlapped():
    s1.wait_stream(s0); s2.wait_stream(s0)
    for c in range(nch):
        hess(c, s1)
        ev = torch.cuda.Event(); ev.record(s1)
        s2.wait_event(ev)
        qr(c, s2)
    s0.wait_stream(s1); s0.wait_stream(s2)


res = {"nmat": nmat, "chunk": chunk, "lib": os.environ.get("EGS_LIB_PATH", "tree")}
res["fused_ms"] = timed(fused)
res["serial_ms"] = timed(serial)
res["only_hess_ms"] = timed(only_hess)
res["only_qr_ms"] = timed(only_qr)
res["overlapped_ms"] = timed(overlapped)
w_ov = w.clone()
serial(); torch.cuda.synchronize()
res["overlap_equals_serial"] = bool(torch.equal(torch.view_as_real(w_ov), torch.view_as_real(w)))
res["unconverged"] = int((info != 0).sum().item())
print(json.dumps(res))
