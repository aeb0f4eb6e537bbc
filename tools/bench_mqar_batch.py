#!/usr/bin/env python3
"""north_star headline check: a 4096-sequence MQAR analysis batch (L = 512, hidden 128, d_state = d_qk = 64) through the
GPU path versus the reference-style CPU path (the oracle's quadratic (B, L, L, H) NumPy restatement of
analysis/eval_eig.py:43-135), the latter timed on a 64-sequence sample -- the reference's own analysis batch size,
configs/iclr2026/analysis_configs/mqar_analysis_config.yaml:1 -- and scaled linearly in B (labelled as such)."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops

def main():
    B, L, D, dqk, H, layers = 4096, 512, 128, 64, 1, 2
    rng = np.random.default_rng(1919)
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    w = (rng.uniform(-1, 1, (2 * dqk + D, D)) / np.sqrt(D)).astype(np.float32)
    b = (rng.uniform(-1, 1, 2 * dqk + D) / np.sqrt(D)).astype(np.float32)
    xd, wd, bd = torch.from_numpy(x).cuda(), torch.from_numpy(w[:2 * dqk]).cuda(), torch.from_numpy(b[:2 * dqk]).cuda()
    out = {"batch": B, "seq_len": L, "hidden": D, "d_state": dqk, "heads": H}
    for name, gpu_fn, cpu_fn in (("lin-attention", ops.linattn_eta, O.eta_linear), ("sm-attention", ops.softmax_eta, O.eta_softmax)):
        gpu_fn(xd, wd, bd, dqk, H); torch.cuda.synchronize()
        t0 = time.perf_counter()
        eta = gpu_fn(xd, wd, bd, dqk, H)
        pct = ops.threshold_hist(eta.unsqueeze(-1), O.THRESHOLDS_RADIUS, "value")
        eta_h, pct_h = eta.cpu().numpy(), pct.cpu().numpy()              # incl. the D2H copy the artefacts need
        t_gpu = time.perf_counter() - t0
        nb = 64
        t0 = time.perf_counter()
        ref = cpu_fn(x[:nb], w, b, dqk, H, D)
        refp = O.threshold_analysis(ref, O.THRESHOLDS_RADIUS, 1, H, nb)
        t_cpu = time.perf_counter() - t0
        err = float(np.max(np.abs(eta_h[:nb] / ref[..., 0] - 1)))
        out[name] = {"gpu_s_full_batch": t_gpu, "cpu_s_sample": t_cpu, "cpu_sample_sequences": nb,
                     "cpu_s_extrapolated_full_batch": t_cpu * B / nb, "speedup_extrapolated": t_cpu * B / nb / t_gpu,
                     "max_rel_err_vs_oracle_on_sample": err, "hist_max_abs_diff_pct_on_sample": float(np.max(np.abs(pct_h[:, :nb].reshape(refp[..., 0].shape) - refp[..., 0])))}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
