"""GPU parity of the block-forward kernels against oracle/blocks_oracle.py and the goldens the
reference's own torch modules produced (tests/golden/blocks.npz).  Everything goes through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import blocks_oracle as BO

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(got, want, rtol=2e-5, atol=1e-6):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max())
    assert err <= rtol * scale + atol, (err, scale)


def test_lin_attention_inner_golden(gpu_ops, golden):
    g = golden("blocks")
    qk, v = dev(g["inner_qk"]), dev(g["inner_v"])
    B, L, H, dv = v.shape
    y = gpu_ops.linear_recurrence_fwd(qk[:, :, 0], qk[:, :, 1], v.reshape(B, L, H * dv), phi=True, normalise=True)
    close(y.reshape(B, L, H, dv), g["inner_lin"])


@pytest.mark.parametrize("nf", ["exp", "elu", "softplus", "sigmoid"])
@pytest.mark.parametrize("af", ["none", "elu"])
@pytest.mark.parametrize("sb", [0, 1])
def test_norm_attention_inner_golden(gpu_ops, golden, nf, af, sb):
    g = golden("blocks")
    qk, v, n = dev(g["inner_qk"]), dev(g["inner_v"]), g["inner_n"]
    B, L, H, dv = v.shape
    w = np.exp(-BO.norm_fn_apply(nf, n + g[f"inner_norm_offset_{nf}"])).astype(np.float32)
    y = gpu_ops.linear_recurrence_fwd(qk[:, :, 0], qk[:, :, 1], v.reshape(B, L, H * dv), w=dev(w), phi=(af == "elu"),
                                      k_scale=(1.0 / np.sqrt(qk.shape[-1]) if sb else 1.0))
    close(y.reshape(B, L, H, dv), g[f"inner_norm_{nf}_{af}_{sb}"])


@pytest.mark.parametrize("shape", [
    # B, L, G, d, cols_per_group, cols_per_decay
    (2, 19, 1, 16, 64, 16),      # Mamba-2 like: 4 heads of 16 share one B/C group
    (3, 33, 2, 24, 32, 32),      # half-filled waves
    (1, 70, 1, 128, 128, 32),    # two column tiles, d = 128, row split over 4 waves
    (2, 25, 1, 200, 96, 96),     # d > 128
    (40, 9, 4, 7, 8, 4),         # tiny tiles, odd d
    (1, 1, 1, 5, 3, 3),          # single token
    (600, 12, 2, 128, 64, 64),   # many tiles: one wave per tile, 128 state rows each
    (2, 40, 1, 512, 16, 16),     # maximum d_state
])
@pytest.mark.parametrize("decay", [False, True])
def test_linear_recurrence_vs_oracle(gpu_ops, shape, decay):
    B, L, G, d, cpg, cpd = shape
    C = G * cpg
    rng = np.random.default_rng(sum(shape) + decay)
    q = (rng.standard_normal((B, L, G, d)) / np.sqrt(d)).astype(np.float32)
    k = (rng.standard_normal((B, L, G, d)) / np.sqrt(d)).astype(np.float32)
    v = rng.standard_normal((B, L, C)).astype(np.float32)
    nh = C // cpd
    a = rng.uniform(0.3, 1.0, (B, L, nh)).astype(np.float32) if decay else None
    s = rng.uniform(0.01, 0.5, (B, L, nh)).astype(np.float32) if decay else None
    w = rng.uniform(0.5, 2.0, (B, L, G)).astype(np.float32) if not decay else None
    skip = rng.standard_normal(C).astype(np.float32)
    want = BO.linear_recurrence(q, k, v, decay=a, vscale=s, w=w, skip=skip, cols_per_decay=cpd)
    got = gpu_ops.linear_recurrence_fwd(dev(q), dev(k), dev(v), decay=None if a is None else dev(a),
                                        vscale=None if s is None else dev(s), w=None if w is None else dev(w),
                                        skip=dev(skip), cols_per_decay=cpd)
    close(got, want, rtol=3e-5)


def test_linear_recurrence_strided_views(gpu_ops):
    """q, k, v taken as slices of one projection output [B, L, 2 d_qk + D] (what MHA.forward hands over)."""
    rng = np.random.default_rng(5)
    B, L, H, hd, D = 3, 21, 2, 8, 24
    qkv = rng.standard_normal((B, L, 2 * H * hd + D)).astype(np.float32)
    t = dev(qkv)
    qk = t[..., : 2 * H * hd].unflatten(-1, (2, H, hd))
    v = t[..., 2 * H * hd:]
    got = gpu_ops.linear_recurrence_fwd(qk[:, :, 0], qk[:, :, 1], v, phi=True, normalise=True)
    want = BO.lin_attention(qkv[..., : 2 * H * hd].reshape(B, L, 2, H, hd), qkv[..., 2 * H * hd:].reshape(B, L, H, D // H))
    close(got, want.reshape(B, L, D))


def test_ssd_scan_vs_oracle(gpu_ops):
    """Mamba-2 SSD as SSD.forward calls it (models/mamba.py:138-150): decay = exp(dt A) is the analysed eigenvalue."""
    rng = np.random.default_rng(8)
    B, L, H, P, N = 2, 50, 4, 16, 16
    x = rng.standard_normal((B, L, H, P)).astype(np.float32)
    dt = np.log1p(np.exp(rng.standard_normal((B, L, H)))).astype(np.float32)
    A = -rng.uniform(1, 16, H).astype(np.float32)
    Bm = rng.standard_normal((B, L, 1, N)).astype(np.float32)
    Cm = rng.standard_normal((B, L, 1, N)).astype(np.float32)
    D = rng.standard_normal(H).astype(np.float32)
    want = BO.ssd_scan(x, dt, A, Bm, Cm, D)
    decay = np.exp(dt * A[None, None, :]).astype(np.float32)
    got = gpu_ops.linear_recurrence_fwd(dev(Cm), dev(Bm), dev(x.reshape(B, L, H * P)), decay=dev(decay), vscale=dev(dt),
                                        skip=dev(np.repeat(D, P)), cols_per_decay=P)
    close(got.reshape(B, L, H, P), want, rtol=3e-5)


def test_linear_recurrence_argument_errors(gpu_ops):
    q = torch.zeros(1, 4, 1, 600, device="cuda")
    v = torch.zeros(1, 4, 8, device="cuda")
    with pytest.raises(Exception):
        gpu_ops.linear_recurrence_fwd(q, q, v)                       # d_state > 512
    with pytest.raises(ValueError):
        gpu_ops.linear_recurrence_fwd(q[..., :8], q[..., :8], torch.zeros(1, 4, 7, device="cuda")[..., ::2])
    with pytest.raises(TypeError):
        gpu_ops.linear_recurrence_fwd(q.double(), q.double(), v)
    empty = gpu_ops.linear_recurrence_fwd(q[:0, :, :, :8], q[:0, :, :, :8], v[:0])
    assert empty.shape == (0, 4, 8)


def test_softmax_attention_inner_golden(gpu_ops, golden):
    g = golden("blocks")
    qk, v = dev(g["inner_qk"]), dev(g["inner_v"])
    B, L, H, dv = v.shape
    y = gpu_ops.softmax_attention_fwd(qk[:, :, 0], qk[:, :, 1], v.reshape(B, L, H * dv))
    close(y.reshape(B, L, H, dv), g["inner_sm"])


@pytest.mark.parametrize("shape", [
    (2, 1, 1, 4, 4),          # single token
    (3, 63, 2, 8, 16),        # one partial tile
    (2, 64, 1, 64, 128),      # exactly one tile, full v chunk
    (1, 200, 3, 24, 40),      # several tiles, ragged head / value widths
    (2, 130, 1, 130, 200),    # head_dim > 64 (two d chunks), v_dim > 128 (two v chunks)
])
def test_softmax_attention_vs_oracle(gpu_ops, shape):
    B, L, H, hd, dv = shape
    rng = np.random.default_rng(sum(shape))
    qk = rng.standard_normal((B, L, 2, H, hd)).astype(np.float32)
    v = rng.standard_normal((B, L, H, dv)).astype(np.float32)
    want = BO.softmax_attention(qk, v)
    t = dev(qk)
    got = gpu_ops.softmax_attention_fwd(t[:, :, 0], t[:, :, 1], dev(v.reshape(B, L, H * dv)))
    close(got.reshape(B, L, H, dv), want, rtol=2e-5)


def test_softmax_attention_large_scores(gpu_ops):
    """Scores of magnitude 1e2..1e3: the online max keeps exp() in range; rows become near one-hot."""
    rng = np.random.default_rng(3)
    B, L, H, hd, dv = 1, 90, 1, 16, 8
    qk = (rng.standard_normal((B, L, 2, H, hd)) * 8).astype(np.float32)
    v = rng.standard_normal((B, L, H, dv)).astype(np.float32)
    want = BO.softmax_attention(qk, v)
    t = dev(qk)
    got = gpu_ops.softmax_attention_fwd(t[:, :, 0], t[:, :, 1], dev(v.reshape(B, L, H * dv)))
    close(got.reshape(B, L, H, dv), want, rtol=1e-4)
