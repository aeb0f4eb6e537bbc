#!/usr/bin/env python3
"""K4 timing: softmax eta (projection + one-pass row statistics + finalize) on the matrix cores vs the VALU tile kernel
(EGS_SOFTMAX_VALU=1); prints JSON lines."""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    shapes = ((4096, 512, 128, 64, 1), (4096, 512, 128, 128, 1), (64, 1024, 512, 512, 8), (512, 2048, 128, 128, 4))
    if len(sys.argv) > 1:
        shapes = shapes[int(sys.argv[1]):int(sys.argv[1]) + 1]
    for B, L, D, dqk, H in shapes:
        x = torch.randn(B, L, D, device="cuda")
        w = torch.randn(2 * dqk, D, device="cuda") / D ** 0.5
        b = torch.randn(2 * dqk, device="cuda") * 0.1
        ms = timeit(lambda: ops.softmax_eta(x, w, b, dqk, H))
        flops = B * H * L * (L + 1) / 2 * (dqk // H) * 2 + B * L * 2.0 * D * 2 * dqk   # one pass of causal score tiles + projection
        print(json.dumps({"B": B, "L": L, "D": D, "d_qk": dqk, "H": H, "ms": ms, "tflops": flops / ms / 1e9,
                          "valu": bool(os.environ.get("EGS_SOFTMAX_VALU"))}), flush=True)


if __name__ == "__main__":
    main()
