"""CPU restatement (NumPy, test infrastructure only) of the reference's token mixers -- SURVEY 8(f) rank 2.

Only tests/, __graft_entry__.smoke() and bench.py's CPU legs may import this module.  Every function is a plain
sequential-in-time loop that follows the cited reference lines; fp32 arithmetic like the torch modules.

Pinning: SelfLinAttention / SelfNormAttention / SelfAttention are plain torch in the reference and were run in this
container (tests/golden/make_golden_blocks.py -> tests/golden/blocks.npz).  The Mamba-2 SSD scan lives in the
third-party mamba_ssm package (requirements: mamba-ssm 2.x, Triton; absent here and not runnable on CPU), so
`ssd_scan` restates the published recurrence of the SSD paper (Dao & Gu 2024, eq. h_t = exp(dt_t A) h_{t-1} +
dt_t B_t x_t^T, y_t = C_t^T h_t + D x_t) as the reference calls it (models/mamba.py:138-150): parity unpinned for it.
"""
import numpy as np


def elu1(x):
    x = np.asarray(x, dtype=np.float32)
    return np.where(x > 0, x + np.float32(1), np.expm1(np.minimum(x, 0)).astype(np.float32) + np.float32(1))


def linear_recurrence(q, k, v, decay=None, vscale=None, w=None, skip=None, cols_per_decay=None):
    """Generic fp32 recurrence S_t = a_t S_{t-1} + k_t (s_t v_t)^T, y_t = w_t q_t^T S_t + skip v_t.
    q, k: [B, L, G, d]; v: [B, L, C]; decay, vscale: [B, L, C // cols_per_decay]; w: [B, L, G]; skip: [C]."""
    q = np.asarray(q, np.float32)
    k = np.asarray(k, np.float32)
    v = np.asarray(v, np.float32)
    B, L, G, d = q.shape
    C = v.shape[-1]
    cpg = C // G
    cpd = cpg if cols_per_decay is None else cols_per_decay
    y = np.zeros((B, L, C), np.float32)
    S = np.zeros((B, G, d, cpg), np.float32)
    for t in range(L):
        vt = v[:, t]
        sv = vt if vscale is None else vt * np.repeat(np.asarray(vscale, np.float32)[:, t], cpd, axis=-1)
        if decay is not None:
            a = np.repeat(np.asarray(decay, np.float32)[:, t], cpd, axis=-1).reshape(B, G, 1, cpg)
            S = S * a
        S = S + k[:, t][:, :, :, None] * sv.reshape(B, G, 1, cpg)
        yt = np.einsum("bgd,bgdc->bgc", q[:, t], S).astype(np.float32)
        if w is not None:
            yt = yt * np.asarray(w, np.float32)[:, t][:, :, None]
        yt = yt.reshape(B, C)
        if skip is not None:
            yt = yt + np.asarray(skip, np.float32)[None, :] * vt
        y[:, t] = yt
    return y


def lin_attention(qk, v):
    """SelfLinAttention.forward (models/attention.py:58-83): qk [B, L, 2, H, d], v [B, L, H, dv] -> [B, L, H, dv]."""
    q = elu1(qk[:, :, 0])
    k = elu1(qk[:, :, 1])
    B, L, H, dv = v.shape
    ksum = np.cumsum(k, axis=1, dtype=np.float32)
    n = np.einsum("bshd,bshd->bsh", q, ksum).astype(np.float32)
    w = np.float32(1) / n
    return linear_recurrence(q, k, np.asarray(v, np.float32).reshape(B, L, H * dv), w=w).reshape(B, L, H, dv)


def norm_fn_apply(name, x):
    x = np.asarray(x, np.float32)
    if name == "exp":
        return np.exp(x)
    if name == "elu":
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(np.float32)
    if name == "softplus":
        return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(np.float32)   # torch threshold 20
    if name == "sigmoid":
        return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)
    raise RuntimeError("normalization function {0} not implemented!".format(name))


def norm_attention(qk, v, n, norm_fn="exp", approx_fn="none", scale_B=False, offset=None):
    """SelfNormAttention.forward (models/norm_attention.py:61-89)."""
    q = np.asarray(qk[:, :, 0], np.float32)
    k = np.asarray(qk[:, :, 1], np.float32)
    if approx_fn == "elu":
        q, k = elu1(q), elu1(k)
    scale = np.float32(1.0 / np.sqrt(q.shape[-1])) if scale_B else np.float32(1)
    B, L, H, dv = v.shape
    raw = np.asarray(n, np.float32)
    if offset is not None:
        raw = raw + np.asarray(offset, np.float32)
    wn = np.exp(-norm_fn_apply(norm_fn, raw)).astype(np.float32)
    return linear_recurrence(q, k * scale, np.asarray(v, np.float32).reshape(B, L, H * dv), w=wn).reshape(B, L, H, dv)


def softmax_attention(qk, v):
    """SelfAttention.forward (models/attention.py:17-35): scale 1/sqrt(d), additive -10000 causal mask, fp32."""
    q = np.asarray(qk[:, :, 0], np.float32)
    k = np.asarray(qk[:, :, 1], np.float32)
    v = np.asarray(v, np.float32)
    L = q.shape[1]
    scores = np.einsum("bthd,bshd->bhts", q, k * np.float32(1.0 / np.sqrt(q.shape[-1]))).astype(np.float32)
    scores = scores + np.triu(np.full((L, L), -10000.0, np.float32), 1)
    scores = scores - scores.max(axis=-1, keepdims=True)
    p = np.exp(scores)
    p = (p / p.sum(axis=-1, keepdims=True)).astype(np.float32)
    return np.einsum("bhts,bshd->bthd", p, v).astype(np.float32)


def ssd_scan(x, dt, A, Bm, Cm, D=None):
    """mamba_chunk_scan_combined(x, dt, A, B, C, D=D, z=None) semantics (models/mamba.py:138-150):
    x [B, L, H, P], dt [B, L, H] (already softplus'd), A [H] (negative), Bm, Cm [B, L, G, N], D [H] -> y [B, L, H, P]."""
    x = np.asarray(x, np.float32)
    dt = np.asarray(dt, np.float32)
    B_, L, H, P = x.shape
    decay = np.exp(dt * np.asarray(A, np.float32)[None, None, :]).astype(np.float32)
    skip = None if D is None else np.repeat(np.asarray(D, np.float32), P)
    y = linear_recurrence(Cm, Bm, x.reshape(B_, L, H * P), decay=decay, vscale=dt, skip=skip, cols_per_decay=P)
    return y.reshape(B_, L, H, P)
