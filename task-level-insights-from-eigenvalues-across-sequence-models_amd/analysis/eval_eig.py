"""Host-side mirror of the reference's ``analysis/eval_eig.py``: same function names, argument meaning, return
shapes / dtypes and error behaviour, with every arithmetic stage dispatched to the HIP kernels behind the C ABI
(``..ops``).  All file:line citations are relative to the reference repository root.

Differences that are deliberate and documented in DESIGN.md:
  * the (B, L, L, H) score tensor of the softmax / linear paths is never materialised (streaming kernels);
  * spectra stay on the GPU until the single device->host copy that the ``.npy`` artefacts require;
  * when ``torch.distributed`` is initialised the analysis batch is sharded across ranks and the spectra are
    collected with one all-gather (``..dist``); the reference is single-process;
  * optional analysis-YAML keys (all default to the reference's behaviour): ``ssm_dtype`` ("complex128" |
    "complex64"), ``s4_all_columns`` (bool), ``ssm_init`` ("jax" | "numpy"), ``s4_solver`` ("dense" | "dplr").
"""
import os
import tempfile
import warnings

import numpy as np
import torch
import yaml

from .. import dist as egs_dist
from .. import ops

THRESHOLDS_RADIUS = [0.1, 0.5, 0.9, 1.0, 10, 100]        # eval_eig.py:603, :665, :724
THRESHOLDS_PHASE = [1, 10, 45, 90, 180]                  # eval_eig.py:612, :671, :732


# ------------------------------------------------------------------------------------------------------------
# layer access (eval_eig.py:28-41)
# ------------------------------------------------------------------------------------------------------------
def get_layers(layer_type, model):
    if layer_type == "mamba":
        return [model.blocks[i] for i in range(len(model.blocks))]
    if layer_type == "transformer":
        return [model.layers[i] for i in range(len(model.layers))]
    return None


def _linear_rows(linear, lo, hi):
    """Rows [lo, hi) of an nn.Linear-like object's weight / bias (bias may be None)."""
    w = linear.weight.detach()[lo:hi]
    b = linear.bias.detach()[lo:hi] if getattr(linear, "bias", None) is not None else None
    return w, b


def _to_host(t, add_axis=True):
    arr = t.detach().cpu().numpy()
    return np.expand_dims(arr, axis=-1) if add_axis else arr


# ------------------------------------------------------------------------------------------------------------
# torch families: device versions (return torch tensors [B, T, H]) and reference-signature wrappers
# ------------------------------------------------------------------------------------------------------------
def eig_att_softmax_dev(x, layer, d_qk, num_heads, d_model):
    w, b = _linear_rows(layer.attention.Wqkv, 0, 2 * d_qk)       # q,k rows only (eval_eig.py:46-52)
    heads = d_qk // layer.attention.head_dim
    return ops.softmax_eta(x, w, b, d_qk, heads)


def eig_att_linear_dev(x, layer, d_qk, num_heads, d_model):
    w, b = _linear_rows(layer.attention.Wqkv, 0, 2 * d_qk)       # eval_eig.py:99-108
    heads = d_qk // layer.attention.head_dim
    return ops.linattn_eta(x, w, b, d_qk, heads)


def eig_att_norm_dev(x, layer, d_qk, num_heads, d_model, model_config):
    norm_fn = model_config["norm_fn"]
    _ = model_config["approx_fn"]                                # read like the reference (KeyError parity)
    use_offset = model_config["offset"]
    lo = d_model + 2 * d_qk                                      # eval_eig.py:156-158
    w, b = _linear_rows(layer.attention.Wvqkn, lo, lo + num_heads)
    offset = layer.attention.inner_attn.offset.detach() if use_offset else None
    return ops.normattn_eta(x, w, b, offset, norm_fn, num_heads)


def eig_mamba2_dev(x, layer):
    m = layer.mamba
    lo = m.d_inner + 2 * m.ngroups * m.d_state                   # eval_eig.py:179-181
    w, b = _linear_rows(m.in_proj, lo, lo + m.nheads)
    return ops.mamba2_lambda(x, w, b, m.dt_bias.detach(), m.A_log.detach())


def eig_mamba2_lti_dev(x, layer):
    m = layer.mamba
    return ops.mamba2_lti_lambda(m.A.detach(), m.beta.detach(), x.shape[0], x.shape[1])


def get_eig_att_softmax(x, layer, d_qk, num_heads, d_model):
    """(B, L-1, H, 1) float64 ndarray, as eval_eig.py:43-95."""
    return _to_host(eig_att_softmax_dev(x, layer, d_qk, num_heads, d_model))


def get_eig_att_linear(x, layer, d_qk, num_heads, d_model):
    """(B, L-1, H, 1) float64 ndarray, as eval_eig.py:97-135."""
    return _to_host(eig_att_linear_dev(x, layer, d_qk, num_heads, d_model))


def get_eig_att_norm(x, layer, d_qk, num_heads, d_model, model_config):
    """(B, L-1, H, 1) float64 ndarray, as eval_eig.py:137-174; RuntimeError for an unknown norm_fn (:151)."""
    return _to_host(eig_att_norm_dev(x, layer, d_qk, num_heads, d_model, model_config))


def get_eig_mamba2(x, layer):
    """(B, L, nheads, 1) float32 ndarray, as eval_eig.py:176-190."""
    return _to_host(eig_mamba2_dev(x, layer))


def get_eig_mamba2_LTI(x, layer):
    """(B, L, nheads, 1) float32 ndarray, as eval_eig.py:192-205."""
    return _to_host(eig_mamba2_lti_dev(x, layer))


# ------------------------------------------------------------------------------------------------------------
# SSM families
# ------------------------------------------------------------------------------------------------------------
def _dev(a, dtype=None, device="cuda"):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
    t = t.to(device)
    return t.to(dtype) if dtype is not None else t


def discrete_DPLR(Lambda, P, Q, B, C, step, L, compute_bc=False):
    """Bilinear discretisation (eval_eig.py:254-278).  Returns (Ab, Bb, Cb).  ``Ab`` comes from the HIP kernel; the
    reference's callers discard ``Bb`` / ``Cb`` (quirk Q18), so they are only produced (torch ops on the device)
    when ``compute_bc`` is set.  P and Q are separate as in the reference (its only caller passes Q = P, :294), and
    Lambda is taken as given: the clip of Re(Lambda) is the caller's (:288), not this function's."""
    lam = _dev(Lambda, torch.complex128).reshape(1, -1)
    p = _dev(P, torch.complex128).reshape(1, -1)
    q = _dev(Q, torch.complex128).reshape(1, -1)
    step_t = torch.as_tensor(step, dtype=torch.float64, device=lam.device).reshape(1)
    ab = ops.s4_dplr_build(lam.real, lam.imag, p, torch.log(step_t), q=q, clip=False)[0]
    if not compute_bc:
        return ab, None, None
    n = lam.shape[1]
    g = 2.0 / step_t
    d = 1.0 / (g - lam[0])
    denom = 1.0 + torch.sum(q[0].conj() * d * p[0])
    a1 = torch.diag(d) - torch.outer(d * p[0], q[0].conj() * d) / denom
    bb = 2 * a1 @ _dev(B, torch.complex128).reshape(-1, 1)
    eye = torch.eye(n, dtype=torch.complex128, device=lam.device)
    cb = _dev(C, torch.complex128).reshape(1, -1) @ torch.linalg.inv(eye - torch.linalg.matrix_power(ab, int(L))).conj()
    return ab, bb, cb.conj()


def _ssm_cast(w, dtype):
    return w.astype(np.complex64) if dtype == "complex64" else w


def _s4_spectra(lam_re, lam_im, p, log_step, solver):
    """eig(Ab) of `ncol` feature columns (eval_eig.py:285-296).  solver "dense" (default; the reference's arithmetic):
    DPLR build + batched Hessenberg / QR; "dplr" (analysis key ``s4_solver: dplr``): the eigenvalues of
    diag(Lambda) - P P^H from the secular equation, Cayley-mapped per column -- no matrix is formed, columns sharing
    (Lambda, P) share one solve."""
    if solver == "dplr":
        w, info = ops.s4_dplr_eigs(lam_re, lam_im, p, log_step)
    elif solver == "dense":
        ab = ops.s4_dplr_build(lam_re, lam_im, p, log_step)             # eval_eig.py:285-294
        w, info = ops.eig_general_batched(ab)                          # eval_eig.py:296
    else:
        raise ValueError("analysis config key 's4_solver' must be 'dense' or 'dplr', got {0!r}".format(solver))
    _raise_on_info(info)
    return w


def get_eigvals_ssm(model, layer_list, layer_nr, idx, SEQ_LEN, dtype="complex128", s4_solver="dense"):
    """(N, 1) complex ndarray (eval_eig.py:281-333); prints and returns None for an unknown family (:331-333).
    ``layer_list[layer_nr]`` maps parameter names to arrays with the flax layouts (models/s4.py:183-189,
    models/s5.py:154-155, :218-222, models/lru.py:55-61)."""
    if model in ["s4"]:
        layer = layer_list[layer_nr]
        log_step = _dev(np.asarray(layer["log_step"])[0, idx].reshape(1), torch.float64)
        lam_re = _dev(np.asarray(layer["Lambda_re"])[:, idx].reshape(1, -1), torch.float64)
        lam_im = _dev(np.asarray(layer["Lambda_im"])[:, idx].reshape(1, -1), torch.float64)
        p = _dev(np.asarray(layer["P"])[:, idx].reshape(1, -1), torch.complex128)
        w = _s4_spectra(lam_re, lam_im, p, log_step, s4_solver)
        return _ssm_cast(w.cpu().numpy().reshape(-1, 1), dtype)
    if model in ["s5"]:
        layer = layer_list[layer_nr]
        w = ops.diag_ssm_lambda("s5", _dev(layer["Lambda_re"]), _dev(layer["Lambda_im"]),
                                _dev(np.asarray(layer["log_step"]).reshape(-1)))
        return _ssm_cast(w.cpu().numpy().reshape(-1, 1), dtype)
    elif model in ["lru"]:
        layer = layer_list[layer_nr]
        w = ops.diag_ssm_lambda("lru", _dev(layer["nu_log"]), _dev(layer["theta_log"]))
        return _ssm_cast(w.cpu().numpy().reshape(-1, 1), dtype)
    else:
        print("model type {0} is not supported!".format(model))
        return None


def get_eigvals_s4_all(layer_list, dtype="complex128", s4_solver="dense"):
    """Generalisation of the S4 branch from the single column ``idx = 1`` (quirk Q9) to every (layer, feature) column:
    one batched DPLR build + one batched eigensolve.  Returns (N, H, num_layers) complex."""
    lam_re, lam_im, p, ls = [], [], [], []
    for layer in layer_list:
        lam_re.append(np.asarray(layer["Lambda_re"]).T)
        lam_im.append(np.asarray(layer["Lambda_im"]).T)
        p.append(np.asarray(layer["P"]).T)
        ls.append(np.asarray(layer["log_step"])[0])
    nl, h, n = len(layer_list), lam_re[0].shape[0], lam_re[0].shape[1]
    w = _s4_spectra(_dev(np.concatenate(lam_re), torch.float64), _dev(np.concatenate(lam_im), torch.float64),
                    _dev(np.concatenate(p), torch.complex128), _dev(np.concatenate(ls), torch.float64), s4_solver)
    return _ssm_cast(w.cpu().numpy().reshape(nl, h, n).transpose(2, 1, 0), dtype)


def _raise_on_info(info):
    bad = int((info != 0).sum().item())
    if bad:
        raise np.linalg.LinAlgError("Eigenvalues did not converge")       # numpy's eigvals raises the same


def _seq_params(params):
    layers = []
    for name in params["encoder"]:
        if name.startswith("layers"):
            layers.append(params["encoder"][name]["seq"])                 # eval_eig.py:234-237, :247-250
    return layers


def get_init_layers_ssm(seed, data_config, train_config, model_config, SEQ_LEN, init_fn, batch_size):
    """Freshly initialised per-layer SSM parameter sub-trees (eval_eig.py:207-239).  Needs jax + flax and the
    reference's ``models`` package, exactly like the reference; ``init_fn`` is the reference's init_S4 / init_S5 /
    init_LRU."""
    import jax
    import jax.numpy as jnp
    from functools import partial
    from jax import random
    from models import BatchClassificationModel

    key = random.PRNGKey(seed)
    init_rng, _ = random.split(key, num=2)
    ssm = init_fn(model_config["state_dim"], model_config["hidden_dim"], **model_config)
    model = partial(BatchClassificationModel, ssm=ssm, d_output=model_config["output_dim"],
                    d_model=model_config["hidden_dim"], n_layers=model_config["num_layers"],
                    activation=model_config["activation"], dropout=model_config["dropout"],
                    pooling=model_config["pooling"], prenorm=model_config["prenorm"], norm=model_config["norm"],
                    padded=False)(training=False)
    rng, dropout_rng = jax.random.split(init_rng, num=2)
    # a batch of one token row is enough to create the parameters (the reference allocates (batch, L, input_dim), Q20)
    variables = model.init({"params": rng, "dropout": dropout_rng},
                           jnp.ones((batch_size, SEQ_LEN, model_config["input_dim"])))
    return _seq_params(variables["params"])


def init_layers_ssm_numpy(kind, seed, model_config):
    """JAX-free stand-in for :func:`get_init_layers_ssm` (analysis YAML key ``ssm_init: numpy``).  The deterministic
    parts follow the reference exactly (HiPPO-LegS DPLR: models/common.py:215-241, models/s4.py:192-215,
    models/s5.py:253-296).  The random parts that reach the eigenvalues -- LRU nu / theta, S4 / S5 log_step, all
    ``jax.random.uniform`` draws -- come from :mod:`jax_prng`, a NumPy restatement of jax's threefry PRNG and of the
    keys flax hands to each parameter, evaluated in float32 like the reference.  That chain is pinned only by the
    Threefry known-answer vectors (no jax / flax here to compare a whole ``model.init``): see jax_prng's header.
    Parameters that do not reach the eigenvalues (B, C, D, gamma) are not produced."""
    from . import jax_prng as jr
    f32 = np.float32
    n, h, nl = model_config["state_dim"], model_config["hidden_dim"], model_config["num_layers"]
    dt_min, dt_max = model_config.get("dt_min", 0.001), model_config.get("dt_max", 0.1)
    span, lo = np.log(dt_max) - np.log(dt_min), np.log(dt_min)            # models/common.py:259-261 (Python floats)
    layers = []
    for li in range(nl):
        if kind == "lru":
            r_min, r_max = model_config.get("r_min", 0.0), model_config.get("r_max", 1.0)
            max_phase = model_config.get("max_phase", 6.28)
            u_theta, u_nu = jr.lru_uniforms(seed, li, n)
            nu = np.log(f32(-0.5) * np.log(u_nu * f32(r_max ** 2 - r_min ** 2) + f32(r_min ** 2)))   # models/lru.py:26-28
            theta = np.log(f32(max_phase) * u_theta)                                               # models/lru.py:31-33
            layers.append({"nu_log": nu.astype(f32), "theta_log": theta.astype(f32)})
        elif kind == "s4":
            lam, p, b = _hippo_dplr(n)
            u = jr.s4_log_step_uniforms(seed, li, h)
            layers.append({"Lambda_re": np.repeat(lam.real[:, None], h, 1), "Lambda_im": np.repeat(lam.imag[:, None], h, 1),
                           "P": np.repeat(p[:, None], h, 1), "B": np.repeat(b[:, None], h, 1),
                           "log_step": (u * f32(span) + f32(lo)).astype(f32)[None, :]})
        elif kind == "s5":
            blocks = model_config.get("num_blocks", 8)
            bs = n // blocks
            lam, _, _ = _hippo_dplr(bs)
            pp = n
            if model_config.get("conj_sym", True):
                bs, pp = bs // 2, n // 2
            lam = np.tile(lam[:bs], blocks)
            u = jr.s5_log_step_uniforms(seed, li, pp, model_config.get("bidirectional", False),
                                        model_config.get("C_init", "lecun_normal"))
            layers.append({"Lambda_re": lam.real, "Lambda_im": lam.imag,
                           "log_step": (u * f32(span) + f32(lo)).astype(f32)[:, None]})
        else:
            raise RuntimeError("{0} is not a valid model option".format(kind))
    return layers


def _hippo_dplr(n):
    """HiPPO-LegS in DPLR form (models/common.py:180-241): returns (Lambda, P, B) in the eigenbasis of the normal part."""
    q = np.sqrt(1 + 2 * np.arange(n))
    a = -(np.tril(q[:, None] * q[None, :]) - np.diag(np.arange(n)))
    p = np.sqrt(np.arange(n) + 0.5)
    b = np.sqrt(2 * np.arange(n) + 1.0)
    s = a + p[:, None] * p[None, :]
    lam_im, v = np.linalg.eigh(s * -1j)                       # Hermitian eigh (models/common.py:236), init-time only
    lam = np.mean(np.diagonal(s)) + 1j * lam_im
    return lam, v.conj().T @ p, v.conj().T @ b


def get_trained_layers_ssm(path):
    """Per-layer trained SSM parameter sub-trees (eval_eig.py:241-252).  An orbax checkpoint directory is restored
    with orbax when it is installed (as the reference does) and otherwise decoded from its aggregate msgpack file
    (ckpt.read_aggregate); a ``.npz`` file with keys ``encoder/layers_<i>/seq/<name>`` is read directly."""
    if str(path).endswith(".npz"):
        flat = np.load(path, allow_pickle=False)
        params = {"encoder": {}}
        for key in flat.files:
            parts = key.split("/")
            if len(parts) == 4 and parts[0] == "encoder" and parts[2] == "seq":
                params["encoder"].setdefault(parts[1], {"seq": {}})["seq"][parts[3]] = flat[key]
        ordered = sorted(params["encoder"], key=lambda s: int(s.split("_")[-1]))
        params["encoder"] = {k: params["encoder"][k] for k in ordered}
        return _seq_params(params)
    try:
        import orbax.checkpoint as ocp
    except ImportError:
        # no orbax on this machine: decode the aggregate msgpack file the reference's save_model writes
        # (jax_helpers/train_helpers.py:395-409) directly
        from .. import ckpt
        raw = ckpt.read_aggregate(path)
    else:
        raw = ocp.PyTreeCheckpointer().restore(path)
    return _seq_params(raw["model"]["params"])


# ------------------------------------------------------------------------------------------------------------
# histograms (eval_eig.py:335-391) -- device kernels behind the reference's signatures
# ------------------------------------------------------------------------------------------------------------
def _hist(vals, thresholds, transform):
    t = vals if isinstance(vals, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(vals))
    if not t.is_cuda:
        t = t.to("cuda")
    return ops.threshold_hist(t, np.asarray(thresholds, dtype=np.float64).flatten(), transform)


def threshold_analysis(eig_val, thresholds, num_layers, num_heads, batch_size, transform="value"):
    """(len(thresholds)+1, B, H, layers) float64 percentages over axis 1 of eig_val (B, N, H, layers)."""
    pct = _hist(eig_val, thresholds, transform)
    if tuple(pct.shape[1:]) != (batch_size, num_heads, num_layers):
        # the reference allocates its output with these sizes (eval_eig.py:342, quirk Q23) and fails on a mismatch
        raise ValueError("could not broadcast input array from shape {0} into shape {1}".format(
            tuple(pct.shape[1:]), (batch_size, num_heads, num_layers)))
    return pct.cpu().numpy()


def threshold_analysis_ssm(eig_val, thresholds, num_layers, transform="value"):
    """(len(thresholds)+1, layers) float64 percentages over axis 0 of eig_val (N, layers)."""
    t = eig_val if isinstance(eig_val, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(eig_val))
    pct = _hist(t.unsqueeze(0), thresholds, transform)
    return pct[:, 0].cpu().numpy()


# ------------------------------------------------------------------------------------------------------------
# text report (eval_eig.py:393-459); always ./percentage_file.txt (quirk Q15)
# ------------------------------------------------------------------------------------------------------------
def create_file_percentage(thresholds_radius, percentage, percentage_init, percentage_mean, percentage_init_mean,
                           percentage_std, percentage_init_std):
    n_heads, n_layers = np.shape(percentage)[2], np.shape(percentage)[3]
    picks = np.array([0, 2, 4, 6])                                    # hard-coded rows: needs B >= 7
    r1 = lambda v: np.round(v, 1)
    with open("percentage_file.txt", "w") as f:
        print("threshold radius:", thresholds_radius, "\n", file=f)
        print("batch selection:", picks, "\n", file=f)
        for pos, b in enumerate(picks):
            for h in range(n_heads):
                for tag, arr in (("radius init: ", percentage_init), ("radius: ", percentage)):
                    for l in range(n_layers):
                        print("percentage batch dimension", b, "head", h, "layer", l, tag, r1(arr[:, b, h, l]), file=f)
                if pos == 0:
                    for stat, tag, arr in (("mean", "radius init: ", percentage_init_mean),
                                           ("mean", "radius: ", percentage_mean),
                                           ("std", "radius init: ", percentage_init_std),
                                           ("std", "radius: ", percentage_std)):
                        for l in range(n_layers):
                            print("percentage batch", stat, "head", h, "layer", l, tag, r1(arr[:, h, l]), file=f)
                print("\n", file=f)
            print("\n", file=f)
    return


def create_file_percentage_ssm(thresholds_radius, thresholds_phase, percentage, percentage_init, percentage_phase,
                               percentage_phase_init):
    n_layers = np.shape(percentage)[1]
    sections = (("radius init: ", percentage_init), ("radius: ", percentage),
                ("phase init: ", percentage_phase_init), ("phase: ", percentage_phase))
    with open("percentage_file.txt", "w") as f:
        print("threshold radius:", thresholds_radius, "\n", file=f)
        print("threshold phase:", thresholds_phase, "\n", file=f)
        for pos, (tag, arr) in enumerate(sections):
            for l in range(n_layers):
                print("percentage layer", l, tag, np.round(arr[:, l], 1), file=f)
            if pos < len(sections) - 1:
                print("\n", file=f)
    return


# ------------------------------------------------------------------------------------------------------------
# driver (eval_eig.py:462-857)
# ------------------------------------------------------------------------------------------------------------
def _torch_family(layer_type, model_config, pseudo_lti):
    if layer_type == "mamba":
        return (lambda x, layer: eig_mamba2_lti_dev(x, layer)) if pseudo_lti else (lambda x, layer: eig_mamba2_dev(x, layer))
    d_qk, d_model, n_heads = model_config["state_dim"], model_config["hidden_dim"], model_config["num_heads"]
    fn = model_config["attention_fn"]
    if fn == "sm-attention":
        return lambda x, layer: eig_att_softmax_dev(x, layer, d_qk, n_heads, d_model)
    if fn == "lin-attention":
        return lambda x, layer: eig_att_linear_dev(x, layer, d_qk, n_heads, d_model)
    if fn == "norm-attention":
        return lambda x, layer: eig_att_norm_dev(x, layer, d_qk, n_heads, d_model, model_config)
    return None


def _has_live_dropout(model):
    return model.training and any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in model.modules())


def _spectra_pass(model, layer_type, loader, family, num_layers, gather=True):
    """One pass of eval_eig.py:500-564 / :574-663: encoder, then block i followed by the spectra of block i computed
    from block i's OUTPUT (quirk Q1).  The batch is sharded across ranks (contiguous slabs, no data-path collective);
    with gather=True the full (B, T, H, layers) tensor is returned on every rank (ONE all_gather), otherwise this
    rank's slab and its [lo, hi) range.

    The reference's "init" pass runs in train mode (quirk Q8): with live dropout the masks come out of one RNG stream
    over the whole batch, so a sharded pass could not reproduce the single-GPU eig_init.  That pass is therefore run
    unsharded on every rank (identical results, no exchange); everything else is sharded."""
    layers = get_layers(layer_type, model)
    X, _, _ = next(iter(loader))
    n_total = X.shape[0]
    full = _has_live_dropout(model)
    lo, hi = (0, n_total) if full else egs_dist.shard_range(n_total)
    with torch.no_grad():
        x = model.encoder(X[lo:hi].to("cuda"))
        per_layer = []
        for i in range(num_layers):
            x = layers[i](x)
            per_layer.append(family(x, layers[i]))
        local = torch.stack(per_layer, dim=-1)                          # (B_local, T, H, layers)
    if full:
        if gather:
            return local
        lo, hi = egs_dist.shard_range(n_total)
        return local[lo:hi], (lo, hi, n_total)
    if gather:
        return egs_dist.all_gather_slabs(local, n_total)
    return local, (lo, hi, n_total)


def _hist_exchange(eig_local, rng, thresholds, num_layers, num_heads, batch_size, transform):
    """Histogram-only exchange: percentages of this rank's batch rows, completed by one
    all_reduce(sum) of the (bins, B, H, layers) array -- the spectra themselves never leave their GPU."""
    lo, hi, n_total = rng
    nb = len(np.asarray(thresholds).flatten()) + 1
    full = torch.zeros((nb, n_total, num_heads, num_layers), dtype=torch.float64, device="cuda")
    if hi > lo:
        full[:, lo:hi] = _hist(eig_local, thresholds, transform)
    full = egs_dist.all_reduce_slab_sum(full)
    if tuple(full.shape[1:]) != (batch_size, num_heads, num_layers):
        raise ValueError("could not broadcast input array from shape {0} into shape {1}".format(
            tuple(full.shape[1:]), (batch_size, num_heads, num_layers)))
    return full.cpu().numpy()


def eval_eig(args, conf_args, wandb_config, data_config, loader, path_file, perf, model_classes=None):
    """Entry point with the reference's 7-argument signature and 6-tuple return (eval_eig.py:462, :855-857).
    ``model_classes`` (optional, keyword) overrides where ``Mamba`` / ``Transformer`` / ``init_*`` come from; by
    default they are imported from the caller's ``models`` package exactly like the reference (eval_eig.py:24-25)."""
    model_config = args["model"]
    train_config = args["train"]
    data_config = args["dataset"]                                        # the argument is ignored (eval_eig.py:465)
    seed = args["seed"]
    batch_size = conf_args["batch_size"]
    num_layers = model_config["num_layers"]
    pseudo_lti = model_config["pseudoLTI"] if "pseudoLTI" in model_config else False
    path = path_file if os.path.isabs(path_file) else os.path.abspath(os.getcwd()) + "/" + path_file
    layer_type = model_config.pop("layer")                               # mutates the caller's dict (eval_eig.py:479)
    ssm_dtype = conf_args.get("ssm_dtype", "complex128")
    # started under a distributed launcher (torchrun launch.py ...)?  Then this call is one rank of the job: join the
    # process group here, because the reference's caller never does (launch.py:176, :183).
    rank, _world = egs_dist.bootstrap()
    hist_only = conf_args.get("exchange", "spectra") == "histogram" and _world > 1

    if layer_type in ["mamba", "transformer"]:
        num_heads = model_config["num_heads"]
        torch.manual_seed(seed)
        if model_classes is None:
            import models as _models
            model_classes = {"mamba": _models.Mamba, "transformer": _models.Transformer}
        model = model_classes[layer_type](model_config).to("cuda")
        family = _torch_family(layer_type, model_config, pseudo_lti)
        thresholds_radius = np.array(THRESHOLDS_RADIUS)
        thresholds_phase = np.array(THRESHOLDS_PHASE)
        if layer_type == "mamba":            # radius / phase of the (real) lambdas (eval_eig.py:603-618)
            rad_tr, ph_tr = "radius", "phase"
        else:                                # eta as is; phase histograms of 0*eta (quirk Q13, eval_eig.py:665-674)
            rad_tr, ph_tr = "value", "zero"
        if hist_only:
            # spectra stay sharded (the returned eig / eig_init are this rank's batch rows); one tiny all_reduce per histogram
            eig_init_t, rng_i = _spectra_pass(model, layer_type, loader, family, num_layers, gather=False)   # train mode (Q8)
            model.load_state_dict(torch.load(path, weights_only=True))
            model.eval()
            eig_t, rng_t = _spectra_pass(model, layer_type, loader, family, num_layers, gather=False)
            percentage_init = _hist_exchange(eig_init_t, rng_i, thresholds_radius, num_layers, num_heads, batch_size, rad_tr)
            percentage = _hist_exchange(eig_t, rng_t, thresholds_radius, num_layers, num_heads, batch_size, rad_tr)
            percentage_phase_init = _hist_exchange(eig_init_t, rng_i, thresholds_phase, num_layers, num_heads, batch_size, ph_tr)
            percentage_phase = _hist_exchange(eig_t, rng_t, thresholds_phase, num_layers, num_heads, batch_size, ph_tr)
        else:
            eig_init_t = _spectra_pass(model, layer_type, loader, family, num_layers)     # train mode (quirk Q8)
            model.load_state_dict(torch.load(path, weights_only=True))
            model.eval()
            eig_t = _spectra_pass(model, layer_type, loader, family, num_layers)
            percentage_init = threshold_analysis(eig_init_t, thresholds_radius, num_layers, num_heads, batch_size, rad_tr)
            percentage = threshold_analysis(eig_t, thresholds_radius, num_layers, num_heads, batch_size, rad_tr)
            percentage_phase_init = threshold_analysis(eig_init_t, thresholds_phase, num_layers, num_heads, batch_size, ph_tr)
            percentage_phase = threshold_analysis(eig_t, thresholds_phase, num_layers, num_heads, batch_size, ph_tr)
        percentage_init_mean = np.mean(percentage_init, axis=1)
        percentage_init_std = np.std(percentage_init, axis=1)
        percentage_mean = np.mean(percentage, axis=1)
        percentage_std = np.std(percentage, axis=1)
        eig, eig_init = eig_t.cpu().numpy(), eig_init_t.cpu().numpy()    # the one device->host copy of the spectra
        if rank == 0:                                                    # one writer of ./percentage_file.txt
            create_file_percentage(thresholds_radius, percentage, percentage_init, percentage_mean, percentage_init_mean,
                                   percentage_std, percentage_init_std)

    elif layer_type in ["lru", "s4", "s5"]:
        SEQ_LEN = model_config["seq_len"]
        dim_idx = 1                                                      # eval_eig.py:689 (quirk Q9)
        ssm_init = conf_args.get("ssm_init", "jax")
        if ssm_init not in ("jax", "numpy"):
            raise ValueError("analysis config key 'ssm_init' must be 'jax' or 'numpy', got {0!r}".format(ssm_init))
        if ssm_init == "numpy":
            note = ("ssm_init: numpy -- the random parts of the {0} initialisation (log_step / nu / theta) were drawn "
                    "by a NumPy restatement of jax's threefry PRNG and flax's parameter keys (analysis/jax_prng.py), "
                    "not by jax itself; that chain is pinned by the Threefry known-answer vectors only, so eig_init, "
                    "percentage_init and percentage_phase_init are guaranteed to follow the reference's "
                    "distributions and are bit-identical to it only if the restated key derivation is "
                    "exact".format(layer_type))
            warnings.warn(note, stacklevel=2)
            args["analysis_notes"] = [note]                              # lands in used_config.yaml next to the arrays
            init_layers = init_layers_ssm_numpy(layer_type, seed, model_config)
        else:
            if model_classes is None:
                try:
                    import jax  # noqa: F401
                    import flax  # noqa: F401
                except ImportError as exc:
                    raise RuntimeError(
                        "analysis config key 'ssm_init' (default 'jax'): the reference initialises the {0} parameters "
                        "with jax + flax (analysis/eval_eig.py:207-239), which are not installed in this environment. "
                        "Either install jax==0.4.25 / flax==0.8.1, or set `ssm_init: numpy` in the analysis YAML to draw "
                        "the random parts (log_step, nu, theta) with NumPy -- same distributions, different PRNG stream, "
                        "so the *_init artefacts are then not bit-identical to the reference's".format(layer_type)) from exc
                import models as _models
                model_classes = {"lru": _models.init_LRU, "s4": _models.init_S4, "s5": _models.init_S5}
            init_layers = get_init_layers_ssm(seed, data_config, train_config, model_config, SEQ_LEN,
                                              model_classes[layer_type], batch_size)
        trained_layers = get_trained_layers_ssm(path)
        s4_solver = conf_args.get("s4_solver", "dense")
        if s4_solver not in ("dense", "dplr"):
            raise ValueError("analysis config key 's4_solver' must be 'dense' or 'dplr', got {0!r}".format(s4_solver))
        if layer_type == "s4" and conf_args.get("s4_all_columns", False):
            eig_init = get_eigvals_s4_all(init_layers, ssm_dtype, s4_solver).reshape(-1, num_layers)
            eig = get_eigvals_s4_all(trained_layers, ssm_dtype, s4_solver).reshape(-1, num_layers)
        else:
            eig_init = np.concatenate([get_eigvals_ssm(layer_type, init_layers, i, dim_idx, SEQ_LEN, ssm_dtype, s4_solver)
                                       for i in range(num_layers)], axis=-1)
            eig = np.concatenate([get_eigvals_ssm(layer_type, trained_layers, i, dim_idx, SEQ_LEN, ssm_dtype, s4_solver)
                                  for i in range(num_layers)], axis=-1)
        thresholds_radius = np.array(THRESHOLDS_RADIUS)
        thresholds_phase = np.array(THRESHOLDS_PHASE)
        percentage_init = threshold_analysis_ssm(eig_init, thresholds_radius, num_layers, "radius")
        percentage = threshold_analysis_ssm(eig, thresholds_radius, num_layers, "radius")
        percentage_phase_init = threshold_analysis_ssm(eig_init, thresholds_phase, num_layers, "phase")
        percentage_phase = threshold_analysis_ssm(eig, thresholds_phase, num_layers, "phase")
        percentage_init_mean = percentage_init_std = percentage_mean = percentage_std = 0      # quirk Q14
        if rank == 0:
            create_file_percentage_ssm(thresholds_radius, thresholds_phase, percentage, percentage_init, percentage_phase,
                                       percentage_phase_init)
    else:
        raise RuntimeError("{0} is not a valid model option".format(layer_type))

    if rank == 0:
        shard = None
        if hist_only and layer_type in ["mamba", "transformer"]:
            # `exchange: histogram`: eig / eig_init hold only THIS rank's batch rows while the percentages cover the whole
            # batch.  The arrays are saved under names that say so and the shard is recorded in used_config.yaml.
            shard = {"exchange": "histogram", "world_size": int(_world), "rank": 0,
                     "batch_rows": [int(rng_t[0]), int(rng_t[1])], "batch_size": int(rng_t[2])}
        _save_artifacts(args, conf_args, wandb_config, data_config, model_config, train_config, perf,
                        [eig, eig_init, percentage, percentage_init, percentage_phase, percentage_phase_init,
                         percentage_mean, percentage_init_mean, percentage_std, percentage_init_std], shard)
    return eig, eig_init, percentage, percentage_init, percentage_phase, percentage_phase_init


_ARRAY_FILES = ["eig", "eig_init", "percentage", "percentage_init", "percentage_phase", "percentage_phase_init",
                "percentage_mean", "percentage_init_mean", "percentage_std", "percentage_init_std"]
# artefact member names, including the reference's "pase" typo (quirk Q16, eval_eig.py:792-801)
_ARTIFACT_NAMES = ["eigen_values_", "eigen_values_init_", "percentage_", "percentage_init_", "percentage_phase_",
                   "percentage_pase_init_", "percentage_mean_", "percentage_init_mean_", "percentage_std_",
                   "percentage_init_std_"]


def _save_artifacts(args, conf_args, wandb_config, data_config, model_config, train_config, perf, arrays, shard=None):
    """eval_eig.py:750-851: 10 .npy + used_config.yaml (+ percentage_file.txt), to a W&B artifact or a local directory.
    shard (histogram exchange only): the two spectra arrays are one rank's rows -- they get a `_rank0_rows<lo>-<hi>`
    suffix instead of the reference's names, and the shard description is written into used_config.yaml."""
    files, members = list(_ARRAY_FILES), list(_ARTIFACT_NAMES)
    if shard is not None:
        tag = "_rank{0}_rows{1}-{2}".format(shard["rank"], shard["batch_rows"][0], shard["batch_rows"][1])
        files[0], files[1] = files[0] + tag, files[1] + tag
        members[0], members[1] = members[0] + tag[1:] + "_", members[1] + tag[1:] + "_"
        args = dict(args)
        args["eigspec_sharding"] = shard
    report = os.path.abspath(os.getcwd()) + "/percentage_file.txt"
    dim_conv = model_config["dim_conv"] if "dim_conv" in model_config else 0
    if wandb_config is not None:
        import wandb
        print("Saving artifact on W&B....")
        stem = data_config["name"] + "{0}-dmodel{1}-seed{4}-num_layers{5}-dqk{2}-conv_dim{6}-lr{3}".format(
            wandb_config["name"], model_config["hidden_dim"], model_config["state_dim"], train_config["lr"],
            args["seed"], model_config["num_layers"], dim_conv)
        name = stem + "-perf{0:0.3f}".format(perf)
        wandb.init(group="artifact_upload", entity=wandb_config["entity"], project=wandb_config["project"],
                   name="upload" + name, job_type="add-dataset")
        artifact = wandb.Artifact(name="eigen_values_" + stem, type="dataset")
        with tempfile.TemporaryDirectory() as tmpdir:
            for fname, arr, member in zip(files, arrays, members):
                fpath = os.path.join(tmpdir, fname + ".npy")
                np.save(fpath, arr)
                artifact.add_file(local_path=fpath, name=member + name)
            cfg = os.path.join(tmpdir, "used_config.yaml")
            with open(cfg, "w") as fh:
                yaml.dump(args, fh, default_flow_style=False, sort_keys=False)
            artifact.add_file(local_path=report, name="percentage_file_" + name)
            artifact.add_file(local_path=cfg, name="used_config-" + name)
            artifact.save()
        try:
            wandb.finish()
        except Exception:
            pass
        return
    print("Saving artifact locally....")
    save_path = conf_args["save_path"] if "save_path" in conf_args else ""
    stem = data_config["name"] + "dmodel{0}-seed{3}-num_layers{4}-dqk{1}-conv_dim{5}-lr{2}".format(
        model_config["hidden_dim"], model_config["state_dim"], train_config["lr"], args["seed"],
        model_config["num_layers"], dim_conv)
    directory = save_path + stem + "-perf{0:0.3f}".format(perf)
    try:
        os.mkdir(directory)                                              # non-recursive on purpose (quirk Q17)
        print(f"Directory '{directory}' created successfully.")
    except FileExistsError:
        print(f"Directory '{directory}' already exists.")
    except PermissionError:
        print(f"Permission denied: Unable to create '{directory}'.")
    except Exception as e:
        print(f"An error occurred: {e}")
    for fname, arr in zip(files, arrays):
        np.save(os.path.join(directory, fname + ".npy"), arr)
    with open(os.path.join(directory, "used_config.yaml"), "w") as fh:
        yaml.dump(args, fh, default_flow_style=False, sort_keys=False)
