"""NumPy restatement of the reference's eigenvalue-extraction arithmetic (TEST INFRASTRUCTURE).

All ``file:line`` citations are relative to ``/root/reference``.  The restatement works on plain
arrays (weights, biases, activations) instead of ``nn.Module`` / flax pytrees so that it can travel
to the GPU box, where the reference does not exist.

Precision model: stages the reference runs in torch fp32 are evaluated in NumPy
float32; the places where it casts to float64 (``eval_eig.py:74, :120, :166``) are reproduced.
The SSM branch runs in complex128 by default (a double-precision restatement; the reference's JAX
default is complex64 -- pass ``dtype=np.complex64`` to emulate it).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "linear_f32", "eta_softmax", "eta_softmax_exact", "eta_softmax_from_qk", "eta_linear", "eta_linear_streaming", "eta_norm",
    "lambda_mamba2", "lambda_mamba2_lti",
    "discrete_dplr", "s4_dplr_ab", "s4_column_inputs", "eigvals_ssm", "lambda_lru", "lambda_s5",
    "radius", "phase_deg", "threshold_analysis", "threshold_analysis_ssm",
    "write_percentage_file", "write_percentage_file_ssm",
    "make_hippo", "make_nplr_hippo", "make_dplr_hippo",
    "eigvals_dense", "sort_spectrum", "max_abs_dlambda",
    "THRESHOLDS_RADIUS", "THRESHOLDS_PHASE", "ZERO_REPLACEMENT",
]

# analysis/eval_eig.py:603, :612, :665, :671, :724, :732
THRESHOLDS_RADIUS = np.array([0.1, 0.5, 0.9, 1.0, 10, 100])
THRESHOLDS_PHASE = np.array([1, 10, 45, 90, 180])
# analysis/eval_eig.py:127, :167  (quirk Q6)
ZERO_REPLACEMENT = 2e-23

_f32 = np.float32


# ----------------------------------------------------------------------------------------------
# torch-family helpers (fp32 stage)
# ----------------------------------------------------------------------------------------------
def linear_f32(x, weight, bias=None):
    """``nn.Linear`` in fp32: ``x @ W.T + b`` (used at analysis/eval_eig.py:46, :99, :154, :177)."""
    y = np.matmul(np.asarray(x, _f32), np.asarray(weight, _f32).T)
    if bias is not None:
        y = y + np.asarray(bias, _f32)
    return y.astype(_f32)


def _softplus_f32(v):
    """``F.softplus`` with torch defaults beta=1, threshold=20 (analysis/eval_eig.py:147, :182, :194)."""
    v = np.asarray(v, _f32)
    with np.errstate(over="ignore"):
        sp = np.log1p(np.exp(np.minimum(v, _f32(20.0)))).astype(_f32)
    return np.where(v > _f32(20.0), v, sp).astype(_f32)


def _elu_f32(v):
    """``F.elu`` alpha=1 (analysis/eval_eig.py:109-110, :145)."""
    v = np.asarray(v, _f32)
    return np.where(v > 0, v, np.expm1(np.minimum(v, _f32(0.0)))).astype(_f32)


def _sigmoid_f32(v):
    v = np.asarray(v, _f32)
    return (_f32(1.0) / (_f32(1.0) + np.exp(-v))).astype(_f32)


def _split_qk(x, w_qkv, b_qkv, d_qk, num_heads):
    """qkv projection and the ``(two h d)`` split of analysis/eval_eig.py:46-52 / :99-108.

    Returns q, k of shape (B, L, H, hd) in fp32.  ``head_dim = d_qk // num_heads`` as in
    models/attention.py:117."""
    qkv = linear_f32(x, w_qkv, b_qkv)
    hd = d_qk // num_heads
    qk = qkv[..., : 2 * d_qk]
    bsz, seqlen = qk.shape[0], qk.shape[1]
    qk = qk.reshape(bsz, seqlen, 2, num_heads, hd)
    return qk[:, :, 0], qk[:, :, 1]


def _masked_scores(q, k):
    """(B, L, L, H) fp32 scores with the strictly-upper triangle multiplied by zero
    (analysis/eval_eig.py:57-59, :115-117).  Quadratic on purpose: this *is* the reference's form."""
    scores = np.einsum("bthd,bshd->btsh", q, k, dtype=_f32).astype(_f32)
    seqlen = q.shape[1]
    tril = np.tril(np.ones((seqlen, seqlen), dtype=_f32))
    return scores * tril[None, :, :, None]


def eta_softmax(x, w_qkv, b_qkv, d_qk, num_heads, d_model):
    """Softmax-attention normaliser ratios, following analysis/eval_eig.py:43-95 (quirks Q3-Q5).

    x: (B, L, D).  Returns (B, L-1, H, 1) float64."""
    q, k = _split_qk(x, w_qkv, b_qkv, d_qk, num_heads)
    seqlen = q.shape[1]
    scores = _masked_scores(q, k)                                  # :57-59
    row_max = scores.max(axis=2)                                   # :62   (zeros take part, Q5)
    tril = np.tril(np.ones((seqlen, seqlen), dtype=_f32))
    shifted = scores - row_max[:, :, None, :] * tril[None, :, :, None]   # :65-72 (fp32)
    shifted = shifted.astype(np.float64)                           # :73-74
    with np.errstate(over="ignore"):
        nu = np.nan_to_num(np.exp(shifted)).sum(axis=2)            # :77-80 (masked entries add 1, Q4)
    eta = nu[:, :-1, :] / nu[:, 1:, :]                             # :83
    max_diff = (-row_max[:, 1:, :] + row_max[:, :-1, :]).astype(np.float64)   # :86-88 (fp32 diff, then f64)
    eta = eta * np.exp(max_diff)                                   # :90
    return eta[..., None]                                          # :93


def eta_softmax_exact(x, w_qkv, b_qkv, d_qk, num_heads):
    """The quantity analysis/eval_eig.py:43-95 defines (quirks Q3-Q5 kept: unscaled scores, zeros of the causal mask
    take part in the row maximum, masked entries add exp(0 - m)), evaluated in float64 FROM the float32 inputs:
    projection, scores, maxima, exponentials and sums all in double.  Both the reference (fp32 projection and scores)
    and the HIP kernels approximate this; the tests measure each one's distance to it."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w_qkv, np.float64)[:2 * d_qk]
    b = np.asarray(b_qkv, np.float64)[:2 * d_qk]
    qk = x @ w.T + b
    bsz, seqlen, _ = x.shape
    hd = d_qk // num_heads
    q = qk[..., :d_qk].reshape(bsz, seqlen, num_heads, hd)
    k = qk[..., d_qk:].reshape(bsz, seqlen, num_heads, hd)
    scores = np.einsum("bthd,bshd->btsh", q, k)
    tril = np.tril(np.ones((seqlen, seqlen)))
    scores = scores * tril[None, :, :, None]                        # masked entries are exact zeros (Q5)
    m = scores.max(axis=2)                                          # (B, L, H)
    # the row maximum is subtracted from the causal entries only (:65-72): every masked entry adds exp(0) = 1 (Q4)
    nu = (np.exp(scores - m[:, :, None, :]) * tril[None, :, :, None]).sum(axis=2) \
        + (seqlen - 1 - np.arange(seqlen))[None, :, None]
    with np.errstate(over="ignore", invalid="ignore"):
        eta = nu[:, :-1] / nu[:, 1:] * np.exp(m[:, :-1] - m[:, 1:])
    return eta[..., None]


def eta_linear(x, w_qkv, b_qkv, d_qk, num_heads, d_model):
    """Linear-attention normaliser ratios, following analysis/eval_eig.py:97-135 (quirk Q6).

    Returns (B, L-1, H, 1) float64.  Materialises (B, L, L, H) like the reference."""
    q, k = _split_qk(x, w_qkv, b_qkv, d_qk, num_heads)
    q = (_elu_f32(q) + _f32(1.0)).astype(_f32)                     # :109
    k = (_elu_f32(k) + _f32(1.0)).astype(_f32)                     # :110
    scores = _masked_scores(q, k).astype(np.float64)               # :115-120
    nu = np.nan_to_num(scores).sum(axis=2)                         # :123-126
    nu[nu == 0.0] = ZERO_REPLACEMENT                               # :127
    eta = nu[:, :-1, :] / nu[:, 1:, :]                             # :130
    return eta[..., None]


def eta_linear_streaming(x, w_qkv, b_qkv, d_qk, num_heads, d_model):
    """O(L) algebraic rewrite of :func:`eta_linear` -- nu_t = phi(q_t) . sum_{s<=t} phi(k_s) -- used as
    the CPU baseline at sizes where the quadratic form cannot run.  The
    prefix sum and the dot run in float64, so it agrees with the quadratic form to fp32 rounding."""
    q, k = _split_qk(x, w_qkv, b_qkv, d_qk, num_heads)
    q = (_elu_f32(q) + _f32(1.0)).astype(np.float64)
    k = (_elu_f32(k) + _f32(1.0)).astype(np.float64)
    nu = np.einsum("blhd,blhd->blh", q, np.cumsum(k, axis=1))
    nu[nu == 0.0] = ZERO_REPLACEMENT
    return (nu[:, :-1, :] / nu[:, 1:, :])[..., None]


_NORM_FNS = {
    "exp": lambda v: np.exp(np.asarray(v, _f32)).astype(_f32),
    "elu": _elu_f32,
    "softplus": _softplus_f32,
    "sigmoid": _sigmoid_f32,
}


def eta_norm(x, w_vqkn, b_vqkn, offset, norm_fn, d_qk, num_heads, d_model):
    """Normalised-attention ratios n_{t+1}/n_t, following analysis/eval_eig.py:137-174.

    ``offset`` is the (H,) parameter or None (``model_config['offset']`` false).  Unknown
    ``norm_fn`` raises RuntimeError with the reference's message (:151)."""
    if norm_fn not in _NORM_FNS:
        raise RuntimeError("normalization function {0} not implemented!".format(norm_fn))
    fn = _NORM_FNS[norm_fn]
    vqkn = linear_f32(x, w_vqkn, b_vqkn)                           # :154
    n = vqkn[..., d_model + 2 * d_qk: d_model + 2 * d_qk + num_heads]   # :156-158
    if offset is not None:
        n = (n + np.asarray(offset, _f32)).astype(_f32)            # :161
    with np.errstate(over="ignore"):
        n = np.exp(-fn(n)).astype(_f32)                            # :161/:163 (fp32)
    n = n.astype(np.float64)                                       # :165-166
    n[n == 0.0] = ZERO_REPLACEMENT                                 # :167
    eta = n[:, 1:, :] / n[:, :-1, :]                               # :169 (inverse orientation)
    return eta[..., None]


def lambda_mamba2(x, w_in, b_in, dt_bias, a_log, d_inner, ngroups, d_state, nheads):
    """Mamba-2 (SSD) per-token transition scalars, analysis/eval_eig.py:176-190.  fp32 throughout (Q7).

    Returns (B, L, nheads, 1) float32."""
    xbcdt = linear_f32(x, w_in, b_in)                              # :177
    a = -np.exp(np.asarray(a_log, _f32))                           # :178
    off = d_inner + 2 * ngroups * d_state
    dt = xbcdt[..., off: off + nheads]                             # :179-181
    dt = _softplus_f32(dt + np.asarray(dt_bias, _f32))             # :182
    lam = np.exp((dt * a).astype(_f32)).astype(_f32)               # :184
    return lam[..., None]                                          # :188


def lambda_mamba2_lti(x, a_param, beta):
    """Pseudo-LTI Mamba-2, analysis/eval_eig.py:192-205: ``exp(beta * -softplus(A))`` broadcast over
    (B, L) (quirk Q19: the in_proj GEMM the reference runs is dead work and is not restated)."""
    bsz, seqlen = x.shape[0], x.shape[1]
    a = -_softplus_f32(a_param)                                    # :194
    lam = np.exp((np.asarray(beta, _f32) * a).astype(_f32)).astype(_f32)   # :197-199
    return np.broadcast_to(lam, (bsz, seqlen, lam.shape[-1])).copy()[..., None]


# ----------------------------------------------------------------------------------------------
# SSM families
# ----------------------------------------------------------------------------------------------
def discrete_dplr(Lambda, P, Q, B, C, step, L, dtype=np.complex128):
    """Bilinear discretisation of a DPLR SSM, analysis/eval_eig.py:254-278 (twin models/s4.py:16-40).

    Returns (Ab, Bb, Cb) like the reference, including the dead Bb/Cb work (Q18)."""
    Lambda = np.asarray(Lambda, dtype)
    P = np.asarray(P, dtype)
    Q = np.asarray(Q, dtype)
    n = Lambda.shape[0]
    eye = np.eye(n, dtype=dtype)
    A = np.diag(Lambda) - np.outer(P, Q.conj())                    # :260
    A0 = (2.0 / step) * eye + A                                    # :264
    dvec = 1.0 / ((2.0 / step) - Lambda)                           # :267
    D = np.diag(dvec)
    Qc = Q.conj().reshape(1, -1)
    P2 = P.reshape(-1, 1)
    A1 = D - (D @ P2 * (1.0 / (1 + (Qc @ D @ P2))) * Qc @ D)      # :270
    Ab = A1 @ A0                                                   # :273
    Bb = 2 * A1 @ np.asarray(B, dtype)[:, None]                    # :274
    Ct = np.asarray(C, dtype)[None, :]
    Cb = Ct @ np.linalg.inv(eye - np.linalg.matrix_power(Ab, L)).conj()   # :277
    return Ab.astype(dtype), Bb.astype(dtype), Cb.conj().astype(dtype)


def s4_column_inputs(layer, idx):
    """The per-column slicing of analysis/eval_eig.py:285-292 (quirks Q9, Q11).

    ``layer`` maps names to arrays with the vmapped (N, H) layout of models/s4.py:183-189.
    Returns (Lambda, P, B, C_tilde, step)."""
    step = np.exp(np.asarray(layer["log_step"])[0, idx])
    lam_re = np.asarray(layer["Lambda_re"])[:, idx]
    lam_im = np.asarray(layer["Lambda_im"])[:, idx]
    Lambda = np.minimum(lam_re, -1e-4) + 1j * lam_im               # :288
    Bv = np.asarray(layer["B"])[:, idx]
    # C is sliced by the reference (:290-291) but its discretised form is discarded by the call at :294; parameter
    # sets that only carry what reaches the eigenvalues (init_layers_ssm_numpy) have none
    C_tilde = None
    if "C" in layer:
        Cm = np.asarray(layer["C"])[:, idx, :]
        C_tilde = Cm[:, 0] + 1j * Cm[:, 1]
    Pv = np.asarray(layer["P"])[:, idx]
    return Lambda, Pv, Bv, C_tilde, step


def s4_dplr_ab(Lambda, P, step, dtype=np.complex128):
    """Only the ``Ab`` of :func:`discrete_dplr` with Q = P (the call at analysis/eval_eig.py:294),
    skipping the discarded Bb/Cb."""
    Lambda = np.asarray(Lambda, dtype)
    P = np.asarray(P, dtype)
    n = Lambda.shape[0]
    g = 2.0 / step
    A0 = g * np.eye(n, dtype=dtype) + np.diag(Lambda) - np.outer(P, P.conj())
    d = 1.0 / (g - Lambda)
    denom = 1.0 + np.sum(P.conj() * d * P)
    A1 = np.diag(d) - np.outer(d * P, P.conj() * d) / denom
    return (A1 @ A0).astype(dtype)


def lambda_lru(nu_log, theta_log, dtype=np.complex128):
    """analysis/eval_eig.py:321-324 (models/lru.py:88)."""
    rdt = np.float64 if dtype == np.complex128 else np.float32
    nu = np.asarray(nu_log, rdt)
    th = np.asarray(theta_log, rdt)
    return np.exp(-np.exp(nu) + 1j * np.exp(th)).astype(dtype)


def lambda_s5(lambda_re, lambda_im, log_step, dtype=np.complex128):
    """analysis/eval_eig.py:306-311 -- always ZOH, never clipped (quirk Q10)."""
    rdt = np.float64 if dtype == np.complex128 else np.float32
    step = np.exp(np.asarray(log_step, rdt).reshape(-1))
    lam = np.asarray(lambda_re, rdt) + 1j * np.asarray(lambda_im, rdt)
    return np.exp(lam * step).astype(dtype)


def eigvals_dense(A):
    """LAPACK ``?geev`` eigenvalues, the third-party arithmetic behind analysis/eval_eig.py:296
    (numpy==1.24.1 in requirements.txt:84; numpy 2.2.6 / OpenBLAS here).  Accepts stacks."""
    return np.linalg.eigvals(np.asarray(A))


def eigvals_ssm(model, layer_list, layer_nr, idx, seq_len, dtype=np.complex128):
    """analysis/eval_eig.py:281-333.  Returns (N, 1) complex, or None (with the reference's message)
    for an unknown family (:331-333)."""
    if model == "s4":
        Lambda, Pv, Bv, Ct, step = s4_column_inputs(layer_list[layer_nr], idx)
        if Ct is None:
            Ct = np.zeros_like(Bv)                                     # only feeds the discarded Cb
        Ab, _, _ = discrete_dplr(Lambda, Pv, Pv, Bv, Ct, step, seq_len, dtype=dtype)   # :294
        return eigvals_dense(Ab)[:, None]                          # :296-299
    if model == "s5":
        layer = layer_list[layer_nr]
        return lambda_s5(layer["Lambda_re"], layer["Lambda_im"], layer["log_step"], dtype)[:, None]
    if model == "lru":
        layer = layer_list[layer_nr]
        return lambda_lru(layer["nu_log"], layer["theta_log"], dtype)[:, None]
    print("model type {0} is not supported!".format(model))
    return None


# ----------------------------------------------------------------------------------------------
# radius / phase / histograms / report
# ----------------------------------------------------------------------------------------------
def radius(eig):
    """analysis/eval_eig.py:605-606, :726-727."""
    eig = np.asarray(eig)
    return np.sqrt(np.power(eig.real, 2) + np.power(eig.imag, 2))


def phase_deg(eig):
    """analysis/eval_eig.py:614-615, :734-735."""
    eig = np.asarray(eig)
    return np.arctan2(eig.imag, eig.real) * 180 / np.pi


def _bin_masks(values, thresholds):
    """The 7 (=len+1) inclusive-edge bins of analysis/eval_eig.py:349-360 (quirk Q12): bin 0 is
    [0, e0], bins 1..n-1 are [e_{i-1}, e_i] (both edges inclusive), the last bin is (e_last, inf)."""
    e = np.asarray(thresholds).flatten()
    masks = [(values >= 0) & (values <= e[0])]
    for lo, hi in zip(e[:-1], e[1:]):
        masks.append((values >= lo) & (values <= hi))
    masks.append(values > e[-1])
    return masks


def threshold_analysis(eig_val, thresholds, num_layers, num_heads, batch_size):
    """analysis/eval_eig.py:335-362.  eig_val (B, T, H, layers) -> (len(thresholds)+1, B, H, layers)
    percentages over the T axis."""
    eig_val = np.asarray(eig_val)
    masks = _bin_masks(eig_val, thresholds)
    out = np.empty([len(masks), batch_size, num_heads, num_layers])
    count = eig_val.shape[1]
    for i, m in enumerate(masks):
        out[i, :, :, :] = m.sum(axis=1) / count * 100
    return out


def threshold_analysis_ssm(eig_val, thresholds, num_layers):
    """analysis/eval_eig.py:364-391.  eig_val (N, layers) -> (len(thresholds)+1, layers)."""
    eig_val = np.asarray(eig_val)
    masks = _bin_masks(eig_val, thresholds)
    out = np.empty([len(masks), num_layers])
    count = eig_val.shape[0]
    for i, m in enumerate(masks):
        out[i, :] = m.sum(axis=0) / count * 100
    return out


def write_percentage_file(path, thresholds_radius, percentage, percentage_init, percentage_mean,
                          percentage_init_mean, percentage_std, percentage_init_std):
    """Text report of analysis/eval_eig.py:393-433 (rows [0,2,4,6] hard-coded, Q15)."""
    num_heads = np.shape(percentage)[2]
    num_layers = np.shape(percentage)[3]
    rows = np.array([0, 2, 4, 6])
    with open(path, "w") as f:
        print("threshold radius:", thresholds_radius, "\n", file=f)
        print("batch selection:", rows, "\n", file=f)
        for bi, b in enumerate(rows):
            for h in range(num_heads):
                for l in range(num_layers):
                    print("percentage batch dimension", b, "head", h, "layer", l, "radius init: ",
                          np.round(percentage_init[:, b, h, l], 1), file=f)
                for l in range(num_layers):
                    print("percentage batch dimension", b, "head", h, "layer", l, "radius: ",
                          np.round(percentage[:, b, h, l], 1), file=f)
                if bi == 0:
                    for l in range(num_layers):
                        print("percentage batch mean head", h, "layer", l, "radius init: ",
                              np.round(percentage_init_mean[:, h, l], 1), file=f)
                    for l in range(num_layers):
                        print("percentage batch mean head", h, "layer", l, "radius: ",
                              np.round(percentage_mean[:, h, l], 1), file=f)
                    for l in range(num_layers):
                        print("percentage batch std head", h, "layer", l, "radius init: ",
                              np.round(percentage_init_std[:, h, l], 1), file=f)
                    for l in range(num_layers):
                        print("percentage batch std head", h, "layer", l, "radius: ",
                              np.round(percentage_std[:, h, l], 1), file=f)
                print("\n", file=f)
            print("\n", file=f)


def write_percentage_file_ssm(path, thresholds_radius, thresholds_phase, percentage, percentage_init,
                              percentage_phase, percentage_phase_init):
    """Text report of analysis/eval_eig.py:435-459."""
    num_layers = np.shape(percentage)[1]
    with open(path, "w") as f:
        print("threshold radius:", thresholds_radius, "\n", file=f)
        print("threshold phase:", thresholds_phase, "\n", file=f)
        for l in range(num_layers):
            print("percentage layer", l, "radius init: ", np.round(percentage_init[:, l], 1), file=f)
        print("\n", file=f)
        for l in range(num_layers):
            print("percentage layer", l, "radius: ", np.round(percentage[:, l], 1), file=f)
        print("\n", file=f)
        for l in range(num_layers):
            print("percentage layer", l, "phase init: ", np.round(percentage_phase_init[:, l], 1), file=f)
        print("\n", file=f)
        for l in range(num_layers):
            print("percentage layer", l, "phase: ", np.round(percentage_phase[:, l], 1), file=f)


# ----------------------------------------------------------------------------------------------
# HiPPO initialisation (needed to synthesise config C5 inputs)
# ----------------------------------------------------------------------------------------------
def make_hippo(n):
    """models/common.py:180-192."""
    p = np.sqrt(1 + 2 * np.arange(n))
    a = np.tril(p[:, None] * p[None, :]) - np.diag(np.arange(n))
    return -a


def make_nplr_hippo(n):
    """models/common.py:194-212."""
    return make_hippo(n), np.sqrt(np.arange(n) + 0.5), np.sqrt(2 * np.arange(n) + 1.0)


def make_dplr_hippo(n):
    """models/common.py:215-241.  Returns (Lambda, P, B, V, B_orig); Hermitian ``eigh`` at :236."""
    a, p, b = make_nplr_hippo(n)
    s = a + p[:, None] * p[None, :]
    lam_real = np.mean(np.diagonal(s)) * np.ones(n)
    lam_imag, v = np.linalg.eigh(s * -1j)
    return lam_real + 1j * lam_imag, v.conj().T @ p, v.conj().T @ b, v, b


# ----------------------------------------------------------------------------------------------
# comparison helpers used by the parity tests
# ----------------------------------------------------------------------------------------------
def sort_spectrum(w):
    """Canonical order for comparing spectra (``|dlambda| after sorting``, BASELINE.json metric):
    lexicographic on (real, imag) rounded to 1e-9 to keep conjugate pairs adjacent."""
    w = np.asarray(w, np.complex128)
    key_re = np.round(w.real, 9)
    order = np.lexsort((w.imag, key_re), axis=-1)
    return np.take_along_axis(w, order, axis=-1)


def max_abs_dlambda(w_a, w_b):
    """max over the batch of the optimal-matching distance between two spectra.  Uses a greedy
    nearest-neighbour matching on the sorted spectra with a linear-sum-assignment fallback for small
    n; robust to the order ambiguity that plain sorting has for nearly-equal real parts."""
    from scipy.optimize import linear_sum_assignment

    w_a = np.asarray(w_a, np.complex128)
    w_b = np.asarray(w_b, np.complex128)
    if w_a.ndim == 1:
        w_a, w_b = w_a[None], w_b[None]
    worst = 0.0
    for a, b in zip(w_a, w_b):
        if not (np.all(np.isfinite(a)) and np.all(np.isfinite(b))):
            return float("inf")
        cost = np.abs(a[:, None] - b[None, :])
        r, c = linear_sum_assignment(cost)
        worst = max(worst, float(cost[r, c].max()))
    return worst


def eta_softmax_from_qk(q, k):
    """notebooks/lm_eigvals.ipynb cell 13 (get_eig_from_qkv_att_softmax): the arithmetic of eta_softmax starting from
    q, k [B, L, H, hd] (fp32 scores, zero mask, row max over the masked row, fp64 exp / sum).  Returns
    (B, L-1, H, 1) float64."""
    q = np.asarray(q, _f32)
    k = np.asarray(k, _f32)
    seqlen = q.shape[1]
    scores = np.einsum("bthd,bshd->btsh", q, k).astype(_f32)
    tril = np.tril(np.ones((seqlen, seqlen), _f32))
    scores = scores * tril[None, :, :, None]
    smax = scores.max(axis=2)                                       # over s, masked zeros included
    smax_r = smax[:, :, None, :] * tril[None, :, :, None]
    norm = (scores - smax_r).astype(_f32).astype(np.float64)
    nu = np.nan_to_num(np.exp(norm)).sum(axis=2)
    eta = nu[:, :-1, :] / nu[:, 1:, :]
    diff = (-smax[:, 1:, :] + smax[:, :-1, :]).astype(_f32)
    return (eta * np.exp(diff.astype(np.float64)))[..., None]
