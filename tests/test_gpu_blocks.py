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
        gpu_ops.linear_recurrence_fwd(q[..., :8], q[..., :8], torch.zeros(2, 4, 8, device="cuda"))   # batch mismatch
    with pytest.raises(ValueError):
        gpu_ops.linear_recurrence_fwd(q[..., :8], q[..., :8], v, decay=torch.ones(1, 4, 3, device="cuda"))
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


# ---- whole models against the reference's own forward (tests/golden/blocks.npz) ------------------------------
CASES = ["lin", "lin_conv_gate", "sm", "sm_conv", "norm_exp", "norm_softplus", "norm_sigmoid", "norm_elu"]


def _golden_model(pkg, g, name):
    import json
    cfg = json.loads(bytes(g[f"{name}/cfg"]).decode())
    keys = json.loads(bytes(g[f"{name}/keys"]).decode())
    torch.manual_seed(0)
    model = pkg.models.Transformer(dict(cfg))
    assert list(model.state_dict().keys()) == keys, "parameter names / order differ from the reference's state_dict"
    model.load_state_dict({k: torch.from_numpy(g[f"{name}/sd/{k}"]) for k in keys})
    return cfg, model.to("cuda").eval()


@pytest.mark.parametrize("name", CASES)
def test_transformer_forward_matches_reference(pkg, gpu_ops, golden, name):
    """State dict from the reference's Transformer loads by name; every block output and the logits match."""
    g = golden("blocks")
    cfg, model = _golden_model(pkg, g, name)
    x = torch.from_numpy(g[f"{name}/x"]).cuda()
    with torch.no_grad():
        h = model.dropout(model.encoder(x))
        close(h, g[f"{name}/h0"], rtol=1e-5)
        for i, layer in enumerate(model.layers):
            h = layer(h)
            close(h, g[f"{name}/h{i + 1}"], rtol=5e-5, atol=1e-5)
        close(model(x), g[f"{name}/y"], rtol=5e-5, atol=1e-5)


def test_models_shim_and_seeded_init(pkg, gpu_ops, golden):
    """`from models import Transformer` resolves to the drop-in; construction after manual_seed draws the same
    parameters as the reference did (creation order and initialisers are mirrored): compare against the golden
    state_dict minus the perturbation is not possible, so check determinism and the documented init values."""
    import models as shim
    assert shim.Transformer is pkg.models.Transformer and shim.Mamba is pkg.models.Mamba
    with pytest.raises(ImportError):
        shim.init_S4
    import json
    g = golden("blocks")
    cfg = json.loads(bytes(g["norm_exp/cfg"]).decode())
    torch.manual_seed(7)
    a = shim.Transformer(dict(cfg)).state_dict()
    torch.manual_seed(7)
    b = shim.Transformer(dict(cfg)).state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.all(a["layers.0.Wz.bias"] == 1.0)                               # models/transformer.py:64
    off = a["layers.0.attention.inner_attn.offset"]
    assert off.shape == (cfg["num_heads"],) and float(off.min()) == 8.0 and abs(float(off.max()) - 14.0) < 1e-5


def test_mamba_block_forward_vs_oracle(pkg, gpu_ops):
    """Mamba-2 block: the HIP scan inside the torch block against the NumPy restatement of the SSD recurrence
    (no golden: the reference's scan is mamba_ssm's Triton kernel)."""
    cfg = dict(version="mamba2", num_layers=2, input_dim=5, output_dim=3, hidden_dim=32, num_heads=4, state_dim=16,
               conv_dim=4, expansion=2, dropout=0.0, glu=True, norm="layer", prenorm=True, dual=False, pooling="none",
               token_embedding=False)
    for lti in (False, True):
        torch.manual_seed(11)
        model = pkg.models.Mamba(dict(cfg, pseudoLTI=lti)).to("cuda").eval()
        x = torch.randn(3, 40, 5, device="cuda")
        with torch.no_grad():
            h = model.encoder(x)
            blk = model.blocks[0]
            m = blk.mamba
            u = blk.norm(h)
            got = m(u)
            # restate the mixer on the host from the same projections
            n_dt = m.ngroups if lti else m.nheads
            xs, Bm, Cm, dt = m._split(u, n_dt)
            xs, Bm, Cm, dt = (t.cpu().numpy() for t in (xs, Bm, Cm, dt))
            D = m.D.detach().cpu().numpy()
            B_, L = xs.shape[:2]
            if lti:
                A = -np.log1p(np.exp(m.A.detach().cpu().numpy()))
                Bm = np.repeat(dt, m.khead_dim, axis=-1) * Bm
                dt_eff = np.ones((B_, L, m.nheads), np.float32)
            else:
                A = -np.exp(m.A_log.detach().cpu().numpy())
                dt_eff = dt
            y = BO.ssd_scan(xs.reshape(B_, L, m.nheads, m.headdim), dt_eff, A, Bm.reshape(B_, L, m.ngroups, m.d_state),
                            Cm.reshape(B_, L, m.ngroups, m.d_state), D).reshape(B_, L, -1)
            want = m.out_proj(torch.from_numpy(y).cuda())
            close(got, want.cpu().numpy(), rtol=5e-5, atol=1e-5)
            out = model(x)
            assert out.shape == (3, 40, 3) and bool(torch.isfinite(out).all())


def test_softmax_attention_mfma_and_valu_paths_agree(gpu_ops, monkeypatch):
    """The matrix-core kernel (head_dim % 4 == 0) and the VALU tile kernel compute the same function."""
    rng = np.random.default_rng(21)
    B, L, H, hd, dv = 2, 150, 2, 32, 48
    qk = dev(rng.standard_normal((B, L, 2, H, hd)).astype(np.float32))
    v = dev(rng.standard_normal((B, L, H * dv)).astype(np.float32))
    a = gpu_ops.softmax_attention_fwd(qk[:, :, 0], qk[:, :, 1], v).cpu().numpy()
    monkeypatch.setenv("EGS_SOFTMAX_VALU", "1")
    b = gpu_ops.softmax_attention_fwd(qk[:, :, 0], qk[:, :, 1], v).cpu().numpy()
    close(a, b, rtol=1e-5)
    want = BO.softmax_attention(qk.cpu().numpy(), v.cpu().numpy().reshape(B, L, H, dv))
    close(a.reshape(B, L, H, dv), want, rtol=2e-5)


@pytest.mark.parametrize("shape", [
    # B, L, G, d, cols_per_group
    (2, 16, 1, 16, 64),        # exactly one chunk
    (3, 37, 2, 8, 12),         # ragged chunk, narrow tiles (one live wave), two groups
    (1, 100, 1, 128, 128),     # d = 128, all four waves, several chunks
    (2, 65, 1, 64, 200),       # two channel chunks per group (128 + 72)
    (5, 1, 3, 4, 5),           # single token
    (2, 50, 1, 20, 40),        # d % 16 != 0 (zero-padded state rows)
])
@pytest.mark.parametrize("mode", ["normalise", "w", "plain"])
def test_linear_recurrence_chunked_matrix_core_path(gpu_ops, monkeypatch, shape, mode):
    """The un-gated form runs chunk-wise on the matrix cores; it must agree with the oracle and with the token-by-token
    kernel (forced through EGS_REC_VALU)."""
    B, L, G, d, cpg = shape
    C = G * cpg
    rng = np.random.default_rng(sum(shape))
    q = rng.standard_normal((B, L, G, d)).astype(np.float32)
    k = rng.standard_normal((B, L, G, d)).astype(np.float32)
    v = rng.standard_normal((B, L, C)).astype(np.float32)
    phi = mode != "plain"
    kw = {}
    if mode == "w":
        kw["w"] = dev(rng.uniform(0.5, 2.0, (B, L, G)).astype(np.float32))
        kw["k_scale"] = 0.37
    if mode == "plain":
        q, k = (q / np.sqrt(d)).astype(np.float32), (k / np.sqrt(d)).astype(np.float32)
    args = (dev(q), dev(k), dev(v))
    got = gpu_ops.linear_recurrence_fwd(*args, phi=phi, normalise=(mode == "normalise"), **kw).cpu().numpy()
    qf, kf = (BO.elu1(q), BO.elu1(k)) if phi else (q, k)
    if mode == "normalise":
        wn = np.float32(1) / np.einsum("blgd,blgd->blg", qf, np.cumsum(kf, axis=1, dtype=np.float32)).astype(np.float32)
    elif mode == "w":
        wn, kf = kw["w"].cpu().numpy(), kf * np.float32(0.37)
    else:
        wn = None
    want = BO.linear_recurrence(qf, kf, v, w=wn)
    close(got, want, rtol=3e-5)
    monkeypatch.setenv("EGS_REC_VALU", "1")
    valu = gpu_ops.linear_recurrence_fwd(*args, phi=phi, normalise=(mode == "normalise"), **kw).cpu().numpy()
    close(got, valu, rtol=3e-5)


@pytest.mark.parametrize("shape", [
    # B, L, H, P (headdim), N (d_state), G (B/C groups)
    (2, 16, 4, 32, 16, 1),      # one chunk, Mamba-2 C3-like heads
    (3, 53, 2, 64, 64, 1),      # ragged chunk, 64-wide heads (two waves per head)
    (1, 130, 8, 32, 128, 1),    # 256 channels: two channel chunks per group, d = 128
    (2, 40, 4, 32, 8, 2),       # two B/C groups
    (2, 33, 3, 32, 20, 1),      # d % 16 != 0, 96 channels (one idle wave)
])
def test_ssd_chunked_matrix_core_path(gpu_ops, monkeypatch, shape):
    """Gated recurrence (Mamba-2 SSD) chunk-wise on the matrix cores against the oracle and the token-by-token kernel,
    including strong decays (a ~ e^-8 per token) and a decay of exactly zero."""
    B, L, H, P, N, G = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((B, L, H, P)).astype(np.float32)
    dt = np.log1p(np.exp(rng.standard_normal((B, L, H)) * 1.5)).astype(np.float32)
    A = -rng.uniform(0.5, 16, H).astype(np.float32)
    Bm = rng.standard_normal((B, L, G, N)).astype(np.float32)
    Cm = rng.standard_normal((B, L, G, N)).astype(np.float32)
    D = rng.standard_normal(H).astype(np.float32)
    decay = np.exp(dt * A[None, None, :]).astype(np.float32)
    decay[0, L // 2, 0] = 0.0                                          # hard reset of one head
    skip = np.repeat(D, P)
    want = BO.linear_recurrence(Cm, Bm, x.reshape(B, L, H * P), decay=decay, vscale=dt, skip=skip, cols_per_decay=P)
    args = (dev(Cm), dev(Bm), dev(x.reshape(B, L, H * P)))
    kw = dict(decay=dev(decay), vscale=dev(dt), skip=dev(skip), cols_per_decay=P)
    got = gpu_ops.linear_recurrence_fwd(*args, **kw).cpu().numpy()
    close(got, want, rtol=5e-5)
    monkeypatch.setenv("EGS_REC_VALU", "1")
    valu = gpu_ops.linear_recurrence_fwd(*args, **kw).cpu().numpy()
    close(got, valu, rtol=5e-5)


def test_mixers_randomised_shapes(gpu_ops):
    """Forty random small configurations (ragged lengths, odd head counts / widths, gated and un-gated, both kernel
    families) against the oracle: shape handling, tails and padding of every path."""
    rng = np.random.default_rng(2027)
    for it in range(40):
        B = int(rng.integers(1, 4))
        L = int(rng.integers(1, 70))
        G = int(rng.integers(1, 4))
        d = int(rng.choice([3, 4, 8, 12, 16, 20, 32, 64, 100, 128, 132]))
        gated = bool(rng.integers(0, 2))
        if gated:
            cpd = int(rng.choice([4, 16, 32, 64]))
            heads_per_group = int(rng.integers(1, 4))
            cpg = cpd * heads_per_group
        else:
            cpg = int(rng.choice([1, 5, 16, 32, 48, 64, 130]))
            cpd = cpg
        C = G * cpg
        q = (rng.standard_normal((B, L, G, d)) / np.sqrt(d)).astype(np.float32)
        k = (rng.standard_normal((B, L, G, d)) / np.sqrt(d)).astype(np.float32)
        v = rng.standard_normal((B, L, C)).astype(np.float32)
        kw, okw = {}, {}
        if gated:
            nh = C // cpd
            a = rng.uniform(0.0, 1.0, (B, L, nh)).astype(np.float32)
            s = rng.uniform(0.01, 1.0, (B, L, nh)).astype(np.float32)
            kw.update(decay=dev(a), vscale=dev(s), cols_per_decay=cpd)
            okw.update(decay=a, vscale=s, cols_per_decay=cpd)
            if rng.integers(0, 2):
                sk = rng.standard_normal(C).astype(np.float32)
                kw["skip"], okw["skip"] = dev(sk), sk
        elif rng.integers(0, 2):
            w = rng.uniform(0.5, 2.0, (B, L, G)).astype(np.float32)
            kw["w"], okw["w"] = dev(w), w
        want = BO.linear_recurrence(q, k, v, **okw)
        got = gpu_ops.linear_recurrence_fwd(dev(q), dev(k), dev(v), **kw)
        close(got, want, rtol=5e-5, atol=1e-6)
        # softmax attention on the same q / k (as heads), value width = cpg
        H, hd = G, d
        y = gpu_ops.softmax_attention_fwd(dev(q), dev(k), dev(v))
        want_sm = BO.softmax_attention(np.stack([q, k], axis=2), v.reshape(B, L, H, cpg))
        close(y.reshape(B, L, H, cpg), want_sm, rtol=5e-5, atol=1e-6)


@pytest.mark.parametrize("cfg", [(2, 37, 5, 1), (3, 16, 64, 4), (1, 100, 130, 3), (2, 5, 8, 8), (4, 33, 257, 2)])
def test_causal_conv1d_silu_vs_torch(gpu_ops, cfg):
    """The fused depthwise causal conv + SiLU against the reference's formulation
    act(conv1d(x^T, padding = K-1)[..., :L])^T (models/attention.py:152-155), also on a strided view."""
    B, L, C, K = cfg
    torch.manual_seed(sum(cfg))
    conv = torch.nn.Conv1d(C, C, kernel_size=K, groups=C, padding=K - 1, bias=True).cuda()
    wide = torch.randn(B, L, C + 7, device="cuda")
    x = wide[..., 3: 3 + C]                                          # channel-contiguous view with a larger token stride
    with torch.no_grad():
        want = torch.nn.functional.silu(conv(x.transpose(1, 2))[..., :L]).transpose(1, 2)
        got = gpu_ops.causal_conv1d_silu(x, conv.weight, conv.bias)
        nobias = gpu_ops.causal_conv1d_silu(x, conv.weight.reshape(C, K), None)
        want_nb = torch.nn.functional.silu(torch.nn.functional.conv1d(x.transpose(1, 2), conv.weight, None, padding=K - 1,
                                                                      groups=C)[..., :L]).transpose(1, 2)
    close(got, want.cpu().numpy(), rtol=2e-6, atol=1e-6)
    close(nobias, want_nb.cpu().numpy(), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("shape", [(7, 128), (3, 5, 32), (2, 3, 4, 200), (1000, 1), (5, 1024), (33, 130)])
def test_layer_norm_vs_torch(gpu_ops, shape):
    torch.manual_seed(sum(shape))
    d = shape[-1]
    x = torch.randn(*shape, device="cuda") * 3 + 1
    ln = torch.nn.LayerNorm(d).cuda()
    with torch.no_grad():
        ln.weight.normal_()
        ln.bias.normal_()
        want = ln(x)
        got = gpu_ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
        plain = gpu_ops.layer_norm(x)
        want_plain = torch.nn.functional.layer_norm(x, (d,))
        wide = torch.randn(6, d + 3, device="cuda")
        strided = gpu_ops.layer_norm(wide[:, :d], ln.weight, ln.bias)
        want_strided = ln(wide[:, :d])
    close(got, want.cpu().numpy(), rtol=3e-6, atol=2e-6)
    close(plain, want_plain.cpu().numpy(), rtol=3e-6, atol=2e-6)
    close(strided, want_strided.cpu().numpy(), rtol=3e-6, atol=2e-6)


def test_backward_through_the_hip_mixers_raises(pkg, gpu_ops):
    """The drop-in models are forward-only.  They carry the names the reference's train.py imports, so a backward pass
    through them must fail loudly instead of silently detaching everything in front of the last HIP op."""
    ln = pkg.models.LayerNorm(16).cuda()
    x = torch.randn(4, 16, device="cuda", requires_grad=True)
    y = ln(x)
    assert y.requires_grad and y.grad_fn is not None
    with pytest.raises(NotImplementedError, match="forward-only"):
        y.sum().backward()
    with torch.no_grad():
        assert not ln(x).requires_grad
    mix = pkg.models.SelfLinAttention().cuda()
    qk = torch.randn(2, 8, 2, 1, 4, device="cuda", requires_grad=True)
    v = torch.randn(2, 8, 1, 8, device="cuda")
    out = mix(qk, v)
    with pytest.raises(NotImplementedError):
        out.sum().backward()


def test_ssd_dt_limit_clamps_decay_and_input_scale(pkg, gpu_ops):
    """mamba_chunk_scan_combined clamps dt to dt_limit before forming exp(dt A) and dt x (models/mamba.py:121, :149):
    a forward with dt_limit must equal the forward of an unlimited mixer fed the clamped dt."""
    torch.manual_seed(3)
    a = pkg.models.SSD(32, d_state=8, d_conv=4, expand=1, headdim=8, dt_limit=(0.02, 0.05)).cuda().eval()
    b = pkg.models.SSD(32, d_state=8, d_conv=4, expand=1, headdim=8).cuda().eval()
    b.load_state_dict(a.state_dict())
    u = torch.randn(2, 16, 32, device="cuda")
    with torch.no_grad():
        ya = a(u)
        x, Bm, Cm, dt = b._split(u, b.nheads)
        dtc = dt.float().clamp(0.02, 0.05)
        yb = b.out_proj(b._scan(x, torch.exp(dtc * -torch.exp(b.A_log.float())), dtc, Bm, Cm))
    assert (dt.float() != dtc).any()                       # the limit is active on this input
    torch.testing.assert_close(ya, yb, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_softmax_attention_more_units_than_resident_workgroups(gpu_ops):
    """Round 4: the flash-forward kernel runs a resident grid that strides over (sequence-head, tile pair) units.  With
    3,000 units every workgroup takes several: each sequence's result must not depend on which other sequences share the
    launch (bit for bit: the two halves of the batch run alone give the same values), and a sample of sequences -- first,
    last, middle -- must match the oracle."""
    rng = np.random.default_rng(44)
    B, L, H, hd, dv = 300, 520, 2, 32, 32                     # T = 9 tiles (odd: the middle tile pairs with itself)
    qk = rng.standard_normal((B, L, 2, H, hd)).astype(np.float32)
    v = rng.standard_normal((B, L, H, dv)).astype(np.float32)
    t, tv = dev(qk), dev(v.reshape(B, L, H * dv))
    full = gpu_ops.softmax_attention_fwd(t[:, :, 0], t[:, :, 1], tv)
    lo = gpu_ops.softmax_attention_fwd(t[:150, :, 0], t[:150, :, 1], tv[:150])
    hi = gpu_ops.softmax_attention_fwd(t[150:, :, 0], t[150:, :, 1], tv[150:])
    assert torch.equal(full[:150], lo) and torch.equal(full[150:], hi)
    pick = [0, 1, 149, 150, 298, 299]
    want = BO.softmax_attention(qk[pick], v[pick])
    close(full[pick].reshape(len(pick), L, H, dv), want, rtol=2e-5)
