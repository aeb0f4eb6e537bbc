#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ from the reference's OWN functions.

Runs only in the dev container (needs /root/reference).  It loads
``/root/reference/analysis/eval_eig.py`` *by file path, unmodified*, with inert ``sys.modules`` stubs
for the third-party modules that are not installed here (jax, orbax, causal_conv1d, wandb, and the
reference's ``models`` package, none of which the called functions need) and with ``jax.numpy``
aliased to NumPy -- so the JAX-side math (``discrete_DPLR``, ``get_eigvals_ssm``) is evaluated in
float64/complex128 rather than JAX's float32 default (SURVEY.md 8(c) caveat i).  The only behavioural
shim is that ``torch.ones(..., device='cuda')`` inside ``get_eig_att_linear`` (eval_eig.py:109-110)
is redirected to the CPU, because no GPU is visible in the dev container.

Nothing from the reference is copied: the committed artefacts are .npz files of seeded inputs and the
outputs the reference's functions returned for them.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference/analysis/eval_eig.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    sys.dont_write_bytecode = True
    jnp = _stub("jax.numpy")
    jnp.__dict__.update({k: getattr(np, k) for k in dir(np) if not k.startswith("__")})
    jnp_linalg = _stub("jax.numpy.linalg", eigh=np.linalg.eigh, inv=np.linalg.inv,
                       matrix_power=np.linalg.matrix_power)
    jnp.linalg = jnp_linalg
    jax_random = _stub("jax.random")
    _stub("jax", numpy=jnp, random=jax_random)
    _stub("orbax")
    sys.modules["orbax"].checkpoint = _stub("orbax.checkpoint")
    _stub("causal_conv1d", causal_conv1d_fn=None)
    _stub("wandb")
    _stub("models", Mamba=None, Transformer=None, BatchClassificationModel=None,
          init_S4=None, init_S5=None, init_LRU=None)
    spec = importlib.util.spec_from_file_location("ref_eval_eig", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class TorchOnCpu:
        """Forwards to torch; drops device='cuda' from torch.ones (no GPU in the dev container)."""

        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def ones(*a, device=None, **kw):
            return torch.ones(*a, **kw)

    mod.torch = TorchOnCpu()
    return mod


def linear(rng, out_f, in_f, scale=1.0):
    lin = torch.nn.Linear(in_f, out_f, bias=True)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy((rng.standard_normal((out_f, in_f)) * scale / np.sqrt(in_f)).astype(np.float32)))
        lin.bias.copy_(torch.from_numpy((rng.standard_normal(out_f) * 0.1).astype(np.float32)))
    return lin


def npy(t):
    return t.detach().cpu().numpy()


def main():
    ref = load_reference()
    scratch = tempfile.mkdtemp()
    os.chdir(scratch)

    # ---------------- attention families ----------------
    rng = np.random.default_rng(20251004)
    B, L, D, d_qk, H = 3, 12, 16, 8, 2
    att = {}
    for case, xscale, wscale in (("a", 1.0, 1.0), ("b", 1.5, 2.0), ("c", 3.0, 4.0)):
        x = torch.from_numpy((rng.standard_normal((B, L, D)) * xscale).astype(np.float32))
        wqkv = linear(rng, 2 * d_qk + D, D, wscale)
        layer = SimpleNamespace(attention=SimpleNamespace(Wqkv=wqkv, head_dim=d_qk // H))
        with torch.no_grad():
            sm = ref.get_eig_att_softmax(x, layer, d_qk, H, D)
            li = ref.get_eig_att_linear(x, layer, d_qk, H, D)
        att.update({f"x_{case}": npy(x), f"w_{case}": npy(wqkv.weight), f"b_{case}": npy(wqkv.bias),
                    f"softmax_{case}": sm, f"linear_{case}": li})
    np.savez(os.path.join(OUT, "attention.npz"), B=B, L=L, D=D, d_qk=d_qk, H=H, **att)

    # ---------------- norm attention ----------------
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.standard_normal((B, L, D)).astype(np.float32) * 2.0)
    wv = linear(rng, D + 2 * d_qk + H, D, 3.0)
    offset = torch.tensor([8.0, 14.0])
    norm = {"x": npy(x), "w": npy(wv.weight), "b": npy(wv.bias), "offset": npy(offset)}
    for fn in ("exp", "elu", "softplus", "sigmoid"):
        for use_off in (False, True):
            layer = SimpleNamespace(attention=SimpleNamespace(Wvqkn=wv, inner_attn=SimpleNamespace(offset=offset)))
            cfg = {"norm_fn": fn, "approx_fn": "none", "offset": use_off}
            with torch.no_grad():
                norm[f"eta_{fn}_{int(use_off)}"] = ref.get_eig_att_norm(x, layer, d_qk, H, D, cfg)
    np.savez(os.path.join(OUT, "norm_attention.npz"), B=B, L=L, D=D, d_qk=d_qk, H=H, **norm)

    # ---------------- mamba2 ----------------
    rng = np.random.default_rng(16)
    d_inner, ngroups, d_state, nheads = 32, 1, 4, 4
    x = torch.from_numpy(rng.standard_normal((B, L, D)).astype(np.float32))
    in_proj = linear(rng, d_inner + 2 * ngroups * d_state + nheads, D, 4.0)
    dt = np.exp(rng.uniform(np.log(1e-3), np.log(1e-1), nheads))
    dt_bias = torch.from_numpy((dt + np.log(-np.expm1(-dt))).astype(np.float32))
    a_log = torch.from_numpy(np.log(rng.uniform(1, 16, nheads)).astype(np.float32))
    a_lti = torch.from_numpy(rng.uniform(-8, -2, nheads).astype(np.float32))
    beta = torch.ones(nheads)
    m = SimpleNamespace(in_proj=in_proj, A_log=a_log, dt_bias=dt_bias, d_inner=d_inner, ngroups=ngroups,
                        d_state=d_state, nheads=nheads, A=a_lti, beta=beta)
    layer = SimpleNamespace(mamba=m)
    with torch.no_grad():
        lam = ref.get_eig_mamba2(x, layer)
        lam_lti = ref.get_eig_mamba2_LTI(x, layer)
    np.savez(os.path.join(OUT, "mamba2.npz"), x=npy(x), w=npy(in_proj.weight), b=npy(in_proj.bias),
             dt_bias=npy(dt_bias), A_log=npy(a_log), A=npy(a_lti), beta=npy(beta),
             d_inner=d_inner, ngroups=ngroups, d_state=d_state, nheads=nheads,
             lam=lam, lam_lti=lam_lti)

    # ---------------- SSM families ----------------
    rng = np.random.default_rng(256)
    N, Hf, SEQ = 8, 3, 16
    s4_layers, s5_layers, lru_layers = [], [], []
    for _ in range(2):
        s4_layers.append({
            "log_step": rng.uniform(np.log(1e-3), np.log(1e-1), (1, Hf)),
            "Lambda_re": -np.abs(rng.standard_normal((N, Hf))) * 0.5 + 0.1,   # some entries > -1e-4: clip quirk Q11
            "Lambda_im": rng.standard_normal((N, Hf)) * 3.0,
            "P": (rng.standard_normal((N, Hf)) + 1j * rng.standard_normal((N, Hf))) * 0.7,
            "B": rng.standard_normal((N, Hf)) + 1j * rng.standard_normal((N, Hf)),
            "C": rng.standard_normal((N, Hf, 2)),
        })
        s5_layers.append({
            "Lambda_re": -np.abs(rng.standard_normal(N)) * 0.5,
            "Lambda_im": rng.standard_normal(N) * 3.0,
            "log_step": rng.uniform(np.log(1e-3), np.log(1e-1), (N, 1)),
        })
        u1, u2 = rng.uniform(size=N), rng.uniform(size=N)
        lru_layers.append({
            "nu_log": np.log(-0.5 * np.log(u1 * (0.99 ** 2 - 0.9 ** 2) + 0.9 ** 2)),
            "theta_log": np.log(6.28 * u2),
        })
    ssm = {}
    for i in range(2):
        for k, v in s4_layers[i].items():
            ssm[f"s4_{i}_{k}"] = v
        for k, v in s5_layers[i].items():
            ssm[f"s5_{i}_{k}"] = v
        for k, v in lru_layers[i].items():
            ssm[f"lru_{i}_{k}"] = v
        ssm[f"s4_{i}_eig"] = np.asarray(ref.get_eigvals_ssm("s4", s4_layers, i, 1, SEQ))
        ssm[f"s5_{i}_eig"] = np.asarray(ref.get_eigvals_ssm("s5", s5_layers, i, 1, SEQ))
        ssm[f"lru_{i}_eig"] = np.asarray(ref.get_eigvals_ssm("lru", lru_layers, i, 1, SEQ))
        lay = s4_layers[i]
        step = np.exp(lay["log_step"][0, 1])
        Lam = np.clip(lay["Lambda_re"][:, 1], None, -1e-4) + 1j * lay["Lambda_im"][:, 1]
        Cm = lay["C"][:, 1, :]
        Ab, Bb, Cb = ref.discrete_DPLR(Lam, lay["P"][:, 1], lay["P"][:, 1], lay["B"][:, 1],
                                       Cm[:, 0] + 1j * Cm[:, 1], step, SEQ)
        ssm[f"s4_{i}_Ab"], ssm[f"s4_{i}_Bb"], ssm[f"s4_{i}_Cb"] = np.asarray(Ab), np.asarray(Bb), np.asarray(Cb)
    ssm["unknown_returns_none"] = np.array(ref.get_eigvals_ssm("s6", s4_layers, 0, 1, SEQ) is None)
    np.savez(os.path.join(OUT, "ssm.npz"), N=N, H=Hf, SEQ=SEQ, **ssm)

    # ---------------- thresholds + report ----------------
    rng = np.random.default_rng(99)
    Bt, T, Ht, Lt = 8, 20, 2, 3
    vals = np.abs(rng.standard_normal((Bt, T, Ht, Lt))) * rng.choice([0.05, 0.7, 1.0, 30.0, 500.0], size=(Bt, T, Ht, Lt))
    edges = [0.1, 0.5, 0.9, 1.0, 10.0, 100.0, 0.0]
    vals[0, : len(edges), 0, 0] = edges            # values exactly on the edges (double counting, Q12)
    vals[1, 0, 1, 1] = -0.3                        # negative: no bin
    vals[1, 1, 1, 1] = np.nan                      # NaN: no bin
    vals[1, 2, 1, 1] = np.inf
    thr_r = np.array([0.1, 0.5, 0.9, 1.0, 10, 100])
    thr_p = np.array([1, 10, 45, 90, 180])
    pct = ref.threshold_analysis(vals, thr_r, Lt, Ht, Bt)
    vals2 = vals * 0.9
    pct_init = ref.threshold_analysis(vals2, thr_r, Lt, Ht, Bt)
    pct_phase = ref.threshold_analysis(0 * vals, thr_p, Lt, Ht, Bt)     # Q13
    ref.create_file_percentage(thr_r, pct, pct_init, pct.mean(axis=1), pct_init.mean(axis=1),
                               pct.std(axis=1), pct_init.std(axis=1))
    report = open("percentage_file.txt").read()

    cplx = (rng.standard_normal((16, 3)) + 1j * rng.standard_normal((16, 3))) * 0.8
    cplx[0, 0] = 1.0 + 0j
    cplx[1, 0] = -1.0 + 0j         # phase exactly 180
    cplx[2, 0] = 0.0 + 0.5j        # phase exactly 90, radius exactly 0.5
    rad = np.sqrt(np.power(cplx.real, 2) + np.power(cplx.imag, 2))
    ph = np.arctan2(cplx.imag, cplx.real) * 180 / np.pi
    pct_s = ref.threshold_analysis_ssm(rad, thr_r, 3)
    pct_s_init = ref.threshold_analysis_ssm(rad * 0.5, thr_r, 3)
    pct_sp = ref.threshold_analysis_ssm(ph, thr_p, 3)
    pct_sp_init = ref.threshold_analysis_ssm(ph * 0.5, thr_p, 3)
    ref.create_file_percentage_ssm(thr_r, thr_p, pct_s, pct_s_init, pct_sp, pct_sp_init)
    report_ssm = open("percentage_file.txt").read()
    np.savez(os.path.join(OUT, "thresholds.npz"), vals=vals, vals2=vals2, pct=pct, pct_init=pct_init,
             pct_phase=pct_phase, report=np.array(report), cplx=cplx, pct_s=pct_s, pct_s_init=pct_s_init,
             pct_sp=pct_sp, pct_sp_init=pct_sp_init, report_ssm=np.array(report_ssm))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
