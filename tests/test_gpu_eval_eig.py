"""End-to-end GPU tests of the drop-in ``eval_eig`` entry point (same 7-argument call as launch.py:183) with small
stand-in models that expose exactly the attributes the reference's analysis touches (model.encoder, model.layers /
model.blocks, layer.attention.Wqkv / .head_dim / .Wvqkn / .inner_attn.offset, layer.mamba.in_proj / .A_log / ...).
The expected values come from the CPU oracle applied to the same block outputs."""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle as O
from oracle import ssm_init_oracle as SI

pytestmark = pytest.mark.gpu

B, L, D, DQK, H, NL, VOCAB = 8, 24, 16, 8, 2, 2, 32


class _Attn(nn.Module):
    def __init__(self, kind):
        super().__init__()
        self.head_dim = DQK // H
        if kind == "norm-attention":
            self.Wvqkn = nn.Linear(D, D + 2 * DQK + H)
            self.inner_attn = nn.Module()
            self.inner_attn.offset = nn.Parameter(torch.tensor([0.3, -0.2]))
        else:
            self.Wqkv = nn.Linear(D, 2 * DQK + D)


class _TBlock(nn.Module):
    def __init__(self, kind):
        super().__init__()
        self.attention = _Attn(kind)
        self.mix = nn.Linear(D, D)

    def forward(self, x):
        return x + torch.tanh(self.mix(x))


class ToyTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.encoder = nn.Embedding(VOCAB, D)
        self.layers = nn.ModuleList([_TBlock(cfg["attention_fn"]) for _ in range(cfg["num_layers"])])


class _SSD(nn.Module):
    def __init__(self):
        super().__init__()
        self.d_inner, self.ngroups, self.d_state, self.nheads = 32, 1, 4, 4
        self.in_proj = nn.Linear(D, self.d_inner + 2 * self.ngroups * self.d_state + self.nheads, bias=False)
        self.A_log = nn.Parameter(torch.log(torch.rand(self.nheads) * 15 + 1))
        self.dt_bias = nn.Parameter(torch.randn(self.nheads))
        self.A = nn.Parameter(-torch.rand(self.nheads) * 6 - 2)
        self.register_buffer("beta", torch.ones(self.nheads))


class _MBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.mamba = _SSD()
        self.mix = nn.Linear(D, D)

    def forward(self, x):
        return x + torch.tanh(self.mix(x))


class ToyMamba(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.encoder = nn.Embedding(VOCAB, D)
        self.blocks = nn.ModuleList([_MBlock() for _ in range(cfg["num_layers"])])


def _args(layer, **model_extra):
    model = {"layer": layer, "num_layers": NL, "num_heads": H if layer == "transformer" else 4, "hidden_dim": D,
             "state_dim": DQK, "seq_len": L}
    model.update(model_extra)
    return {"model": model, "train": {"lr": 0.01}, "dataset": {"name": "TOY"}, "seed": 1919}


def _loader():
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.integers(0, VOCAB, (B, L)))
    return [(X, torch.zeros(B), None)]


def _block_outputs(model, layers, X):
    with torch.no_grad():
        x = model.encoder(X.cuda())
        outs = []
        for layer in layers:
            x = layer(x)
            outs.append(x.cpu().numpy())
    return outs


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("attention_fn", ["sm-attention", "lin-attention", "norm-attention"])
def test_eval_eig_transformer_families(pkg, tmp_path, monkeypatch, attention_fn):
    monkeypatch.chdir(tmp_path)
    os.mkdir(tmp_path / "results")
    extra = {"attention_fn": attention_fn}
    if attention_fn == "norm-attention":
        extra.update(norm_fn="softplus", approx_fn="none", offset=True)
    args = _args("transformer", **extra)
    torch.manual_seed(7)
    trained = ToyTransformer(args["model"])
    torch.save(trained.state_dict(), tmp_path / "ckpt.pth")
    conf = {"batch_size": B, "save_path": str(tmp_path / "results") + "/"}
    loader = _loader()
    out = pkg.eval_eig(copy.deepcopy(args), conf, None, None, loader, "ckpt.pth", 0.5,
                       model_classes={"transformer": ToyTransformer})
    eig, eig_init, pct, pct_init, pct_phase, pct_phase_init = out
    assert eig.shape == eig_init.shape == (B, L - 1, H, NL) and eig.dtype == np.float64
    assert pct.shape == pct_init.shape == (7, B, H, NL) and pct_phase.shape == (6, B, H, NL)

    # expected: oracle on the block outputs of the same trained weights (quirk Q1: spectra of block i's OUTPUT)
    trained = trained.cuda().eval()
    xs = _block_outputs(trained, trained.layers, loader[0][0])
    want = []
    for x, layer in zip(xs, trained.layers):
        if attention_fn == "sm-attention":
            want.append(O.eta_softmax(x, _np(layer.attention.Wqkv.weight), _np(layer.attention.Wqkv.bias), DQK, H, D))
        elif attention_fn == "lin-attention":
            want.append(O.eta_linear(x, _np(layer.attention.Wqkv.weight), _np(layer.attention.Wqkv.bias), DQK, H, D))
        else:
            want.append(O.eta_norm(x, _np(layer.attention.Wvqkn.weight), _np(layer.attention.Wvqkn.bias),
                                   _np(layer.attention.inner_attn.offset), "softplus", DQK, H, D))
    want = np.concatenate(want, axis=-1)
    np.testing.assert_allclose(eig, want, rtol=5e-5)
    np.testing.assert_array_equal(pct, O.threshold_analysis(eig, O.THRESHOLDS_RADIUS, NL, H, B))
    with np.errstate(invalid="ignore"):
        np.testing.assert_array_equal(pct_phase, O.threshold_analysis(0 * eig, O.THRESHOLDS_PHASE, NL, H, B))   # Q13

    # the init pass used a model built under torch.manual_seed(seed) (eval_eig.py:484)
    torch.manual_seed(1919)
    init = ToyTransformer(args["model"]).cuda()
    xs = _block_outputs(init, init.layers, loader[0][0])
    want_init = []
    for x, layer in zip(xs, init.layers):                      # every layer of the init pass, every family
        if attention_fn == "sm-attention":
            want_init.append(O.eta_softmax(x, _np(layer.attention.Wqkv.weight), _np(layer.attention.Wqkv.bias), DQK, H, D))
        elif attention_fn == "lin-attention":
            want_init.append(O.eta_linear(x, _np(layer.attention.Wqkv.weight), _np(layer.attention.Wqkv.bias), DQK, H, D))
        else:
            want_init.append(O.eta_norm(x, _np(layer.attention.Wvqkn.weight), _np(layer.attention.Wvqkn.bias),
                                        _np(layer.attention.inner_attn.offset), "softplus", DQK, H, D))
    np.testing.assert_allclose(eig_init, np.concatenate(want_init, axis=-1), rtol=5e-5)
    np.testing.assert_array_equal(pct_init, O.threshold_analysis(eig_init, O.THRESHOLDS_RADIUS, NL, H, B))

    # artefacts (eval_eig.py:807-851) and the text report (eval_eig.py:393-433)
    name = "TOYdmodel{0}-seed1919-num_layers{1}-dqk{2}-conv_dim0-lr0.01-perf0.500".format(D, NL, DQK)
    saved = sorted(os.listdir(tmp_path / "results" / name))
    assert saved == sorted([f + ".npy" for f in ["eig", "eig_init", "percentage", "percentage_init", "percentage_phase",
                                                 "percentage_phase_init", "percentage_mean", "percentage_init_mean",
                                                 "percentage_std", "percentage_init_std"]] + ["used_config.yaml"])
    np.testing.assert_array_equal(np.load(tmp_path / "results" / name / "eig.npy"), eig)
    O.write_percentage_file(str(tmp_path / "expected.txt"), O.THRESHOLDS_RADIUS, pct, pct_init, pct.mean(axis=1),
                            pct_init.mean(axis=1), pct.std(axis=1), pct_init.std(axis=1))
    assert open(tmp_path / "percentage_file.txt").read() == open(tmp_path / "expected.txt").read()


@pytest.mark.parametrize("pseudo_lti", [False, True])
def test_eval_eig_mamba(pkg, tmp_path, monkeypatch, pseudo_lti):
    monkeypatch.chdir(tmp_path)
    args = _args("mamba", pseudoLTI=pseudo_lti)
    torch.manual_seed(11)
    trained = ToyMamba(args["model"])
    torch.save(trained.state_dict(), tmp_path / "ckpt.pth")
    loader = _loader()
    eig, eig_init, pct, pct_init, pct_phase, pct_phase_init = pkg.eval_eig(
        copy.deepcopy(args), {"batch_size": B}, None, None, loader, str(tmp_path / "ckpt.pth"), 0.25,
        model_classes={"mamba": ToyMamba})
    assert eig.shape == (B, L, 4, NL) and eig.dtype == np.float32            # fp32, full length (quirk Q7)
    trained = trained.cuda().eval()
    xs = _block_outputs(trained, trained.blocks, loader[0][0])
    want = []
    for x, blk in zip(xs, trained.blocks):
        m = blk.mamba
        if pseudo_lti:
            want.append(O.lambda_mamba2_lti(x, _np(m.A), _np(m.beta)))
        else:
            want.append(O.lambda_mamba2(x, _np(m.in_proj.weight), None, _np(m.dt_bias), _np(m.A_log), m.d_inner,
                                        m.ngroups, m.d_state, m.nheads))
    want = np.concatenate(want, axis=-1)
    np.testing.assert_allclose(eig, want, rtol=1e-5, atol=1e-7)
    # the init pass: a model built under torch.manual_seed(seed) (eval_eig.py:484), every layer
    torch.manual_seed(1919)
    init = ToyMamba(args["model"]).cuda()
    xs_i = _block_outputs(init, init.blocks, loader[0][0])
    want_i = []
    for x, blk in zip(xs_i, init.blocks):
        m = blk.mamba
        if pseudo_lti:
            want_i.append(O.lambda_mamba2_lti(x, _np(m.A), _np(m.beta)))
        else:
            want_i.append(O.lambda_mamba2(x, _np(m.in_proj.weight), None, _np(m.dt_bias), _np(m.A_log), m.d_inner,
                                          m.ngroups, m.d_state, m.nheads))
    np.testing.assert_allclose(eig_init, np.concatenate(want_i, axis=-1), rtol=1e-5, atol=1e-7)
    rad = np.sqrt(np.power(eig.real, 2) + np.power(eig.imag, 2))
    np.testing.assert_array_equal(pct, O.threshold_analysis(rad, O.THRESHOLDS_RADIUS, NL, 4, B))
    ph = np.arctan2(eig.imag, eig.real) * 180 / np.pi
    np.testing.assert_array_equal(pct_phase, O.threshold_analysis(ph, O.THRESHOLDS_PHASE, NL, 4, B))
    assert os.path.isdir(tmp_path / "TOYdmodel{0}-seed1919-num_layers{1}-dqk{2}-conv_dim0-lr0.01-perf0.250".format(D, NL, DQK))


@pytest.mark.parametrize("kind", ["lru", "s5", "s4"])
def test_eval_eig_ssm_families(pkg, tmp_path, monkeypatch, kind):
    monkeypatch.chdir(tmp_path)
    ev = sys.modules[pkg.__name__ + ".analysis.eval_eig"]
    n, h = (8 if kind == "s4" else 16), 4      # HiPPO-LegS spectra are too ill-conditioned beyond N ~ 8 for 1e-5
    args = _args(kind, state_dim=n, hidden_dim=h, num_blocks=2, r_min=0.9, r_max=0.99)
    trained = ev.init_layers_ssm_numpy(kind, 77, args["model"])
    flat = {f"encoder/layers_{i}/seq/{k}": v for i, layer in enumerate(trained) for k, v in layer.items()}
    np.savez(tmp_path / "ckpt.npz", **flat)
    conf = {"batch_size": B, "ssm_init": "numpy"}
    eig, eig_init, pct, pct_init, pct_phase, pct_phase_init = pkg.eval_eig(
        copy.deepcopy(args), conf, None, None, None, str(tmp_path / "ckpt.npz"), 0.9)
    rows = n // 2 if kind == "s5" else n
    assert eig.shape == eig_init.shape == (rows, NL) and np.iscomplexobj(eig)
    assert pct.shape == (7, NL) and pct_phase.shape == (6, NL)
    want = np.concatenate([O.eigvals_ssm(kind, trained, i, 1, L) for i in range(NL)], axis=-1)
    assert O.max_abs_dlambda(eig.T, want.T) < (1e-5 if kind == "s4" else 1e-12)
    # expected initial parameters from the ORACLE's own restatement of the reference's draw (oracle/ssm_init_oracle.py,
    # scalar-integer Threefry + its own flax key bookkeeping), not from the product's init_layers_ssm_numpy
    init = SI.init_layers(kind, 1919, args["model"])
    want_init = np.concatenate([O.eigvals_ssm(kind, init, i, 1, L) for i in range(NL)], axis=-1)
    assert O.max_abs_dlambda(eig_init.T, want_init.T) < (1e-5 if kind == "s4" else 1e-12)
    if kind != "s4":          # histograms of exactly-matching spectra are bit-identical
        np.testing.assert_array_equal(pct, O.threshold_analysis_ssm(O.radius(eig), O.THRESHOLDS_RADIUS, NL))
        np.testing.assert_array_equal(pct_phase, O.threshold_analysis_ssm(O.phase_deg(eig), O.THRESHOLDS_PHASE, NL))
    O.write_percentage_file_ssm(str(tmp_path / "expected.txt"), O.THRESHOLDS_RADIUS, O.THRESHOLDS_PHASE, pct, pct_init,
                                pct_phase, pct_phase_init)
    assert open(tmp_path / "percentage_file.txt").read() == open(tmp_path / "expected.txt").read()
    saved = np.load(tmp_path / "TOYdmodel{0}-seed1919-num_layers{1}-dqk{2}-conv_dim0-lr0.01-perf0.900".format(h, NL, n)
                    / "percentage_mean.npy")
    assert saved.shape == () and saved == 0                                    # quirk Q14


def test_eval_eig_ssm_init_key_is_explicit(pkg, tmp_path, monkeypatch):
    """Without jax / flax the default `ssm_init: jax` must stop with a message that names the analysis-YAML key (not an
    ImportError from deep inside a shim); `ssm_init: numpy` works, warns, and records the substitution in
    used_config.yaml next to the arrays it affects."""
    try:
        import jax  # noqa: F401
        import flax  # noqa: F401
        pytest.skip("jax + flax are installed: the reference's own init path is available")
    except ImportError:
        pass
    monkeypatch.chdir(tmp_path)
    ev = sys.modules[pkg.__name__ + ".analysis.eval_eig"]
    args = _args("lru", state_dim=16, hidden_dim=4, r_min=0.9, r_max=0.99)
    trained = ev.init_layers_ssm_numpy("lru", 77, args["model"])
    np.savez(tmp_path / "ckpt.npz", **{f"encoder/layers_{i}/seq/{k}": v for i, layer in enumerate(trained)
                                       for k, v in layer.items()})
    with pytest.raises(RuntimeError, match="ssm_init"):
        pkg.eval_eig(copy.deepcopy(args), {"batch_size": B}, None, None, None, str(tmp_path / "ckpt.npz"), 0.9)
    with pytest.raises(ValueError, match="ssm_init"):
        pkg.eval_eig(copy.deepcopy(args), {"batch_size": B, "ssm_init": "threefry"}, None, None, None,
                     str(tmp_path / "ckpt.npz"), 0.9)
    with pytest.warns(UserWarning, match="ssm_init: numpy"):
        pkg.eval_eig(copy.deepcopy(args), {"batch_size": B, "ssm_init": "numpy"}, None, None, None,
                     str(tmp_path / "ckpt.npz"), 0.9)
    import yaml
    used = yaml.safe_load(open(tmp_path / "TOYdmodel4-seed1919-num_layers{0}-dqk16-conv_dim0-lr0.01-perf0.900".format(NL)
                               / "used_config.yaml"))
    assert "ssm_init: numpy" in used["analysis_notes"][0]


def test_eval_eig_s4_all_columns_extension(pkg, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    ev = sys.modules[pkg.__name__ + ".analysis.eval_eig"]
    n, h = 8, 4
    args = _args("s4", state_dim=n, hidden_dim=h)
    trained = ev.init_layers_ssm_numpy("s4", 5, args["model"])
    np.savez(tmp_path / "ckpt.npz", **{f"encoder/layers_{i}/seq/{k}": v for i, layer in enumerate(trained)
                                       for k, v in layer.items()})
    eig = pkg.eval_eig(copy.deepcopy(args), {"batch_size": B, "ssm_init": "numpy", "s4_all_columns": True}, None, None,
                       None, str(tmp_path / "ckpt.npz"), 0.1)[0]
    assert eig.shape == (n * h, NL)
    cube = eig.reshape(n, h, NL)
    for c in range(h):
        want = np.concatenate([O.eigvals_ssm("s4", trained, i, c, L) for i in range(NL)], axis=-1)
        assert O.max_abs_dlambda(cube[:, c, :].T, want.T) < 1e-5


# ---- the drop-in torch models (`from models import Mamba, Transformer`, as the reference's eval_eig.py:24-25) ----
def _real_transformer_cfg(attention_fn):
    cfg = dict(layer="transformer", input_dim=1, output_dim=VOCAB, num_layers=NL, hidden_dim=D, embedding=True,
               vocab_size=VOCAB, max_pos_embed=0, pooling="none", dual=False, classifier=False, mixer_dim=24,
               norm="layer", dropout=0.1, state_dim=DQK, num_heads=H, att_dropout=0.0, mixer="mlp", use_flash=False,
               attention_fn=attention_fn, dim_conv=3, seq_len=L)
    if attention_fn == "norm-attention":
        cfg.update(mode="attention", norm_fn="softplus", approx_fn="elu", scale_B=True, offset=True,
                   offset_init="uniform", learn_A=False)
    return cfg


@pytest.mark.parametrize("attention_fn", ["sm-attention", "lin-attention", "norm-attention"])
def test_eval_eig_with_dropin_transformer(pkg, tmp_path, monkeypatch, attention_fn):
    """No model_classes override: eval_eig imports Transformer from ``models`` like the reference does, builds it
    from the YAML-style config, loads the .pth checkpoint by parameter name and forwards through the HIP mixers."""
    monkeypatch.chdir(tmp_path)
    os.mkdir(tmp_path / "results")
    cfg = _real_transformer_cfg(attention_fn)
    args = {"model": cfg, "train": {"lr": 0.01}, "dataset": {"name": "TOY"}, "seed": 1919}
    torch.manual_seed(3)
    trained = pkg.models.Transformer({k: v for k, v in cfg.items() if k != "layer"})
    with torch.no_grad():
        for p in trained.parameters():
            p.add_(0.05 * torch.randn_like(p))
    torch.save(trained.state_dict(), tmp_path / "ckpt.pth")
    conf = {"batch_size": B, "save_path": str(tmp_path / "results") + "/"}
    loader = _loader()
    eig, eig_init, pct, pct_init, _, _ = pkg.eval_eig(copy.deepcopy(args), conf, None, None, loader, "ckpt.pth", 0.9)
    assert eig.shape == eig_init.shape == (B, L - 1, H, NL)
    assert not np.allclose(eig, eig_init)                          # init weights (seed 1919) vs the checkpoint

    trained = trained.cuda().eval()
    xs = _block_outputs(trained, trained.layers, loader[0][0])
    want = []
    for x, layer in zip(xs, trained.layers):
        att = layer.attention
        if attention_fn == "sm-attention":
            want.append(O.eta_softmax(x, _np(att.Wqkv.weight), _np(att.Wqkv.bias), DQK, H, D))
        elif attention_fn == "lin-attention":
            want.append(O.eta_linear(x, _np(att.Wqkv.weight), _np(att.Wqkv.bias), DQK, H, D))
        else:
            want.append(O.eta_norm(x, _np(att.Wvqkn.weight), _np(att.Wvqkn.bias), _np(att.inner_attn.offset),
                                   "softplus", DQK, H, D))
    np.testing.assert_allclose(eig, np.concatenate(want, axis=-1), rtol=5e-5)
    np.testing.assert_array_equal(pct, O.threshold_analysis(eig, O.THRESHOLDS_RADIUS, NL, H, B))
    assert sorted(os.listdir(tmp_path / "results"))                # artefacts were written


@pytest.mark.parametrize("pseudo_lti", [False, True])
def test_eval_eig_with_dropin_mamba(pkg, tmp_path, monkeypatch, pseudo_lti):
    monkeypatch.chdir(tmp_path)
    os.mkdir(tmp_path / "results")
    nheads = 4
    cfg = dict(layer="mamba", version="mamba2", num_layers=NL, input_dim=1, output_dim=VOCAB, hidden_dim=D,
               num_heads=nheads, state_dim=8, conv_dim=4, expansion=1, dropout=0.0, glu=True, norm="layer", prenorm=True,
               dual=False, pooling="none", token_embedding=True, vocab_size=VOCAB, pseudoLTI=pseudo_lti, seq_len=L)
    args = {"model": cfg, "train": {"lr": 0.01}, "dataset": {"name": "TOY"}, "seed": 1919}
    torch.manual_seed(5)
    trained = pkg.models.Mamba({k: v for k, v in cfg.items() if k != "layer"})
    torch.save(trained.state_dict(), tmp_path / "ckpt.pth")
    conf = {"batch_size": B, "save_path": str(tmp_path / "results") + "/"}
    loader = _loader()
    eig, eig_init, pct, _, pct_phase, _ = pkg.eval_eig(copy.deepcopy(args), conf, None, None, loader, "ckpt.pth", 0.9)
    m0 = trained.blocks[0].mamba
    assert eig.shape == (B, L, m0.nheads, NL) and eig.dtype == np.float32
    trained = trained.cuda().eval()
    xs = _block_outputs(trained, trained.blocks, loader[0][0])
    want = []
    for x, blk in zip(xs, trained.blocks):
        m = blk.mamba
        if pseudo_lti:
            want.append(O.lambda_mamba2_lti(x, _np(m.A), _np(m.beta)))
        else:
            want.append(O.lambda_mamba2(x, _np(m.in_proj.weight), None, _np(m.dt_bias), _np(m.A_log), m.d_inner,
                                        m.ngroups, m.d_state, m.nheads))
    np.testing.assert_allclose(eig, np.concatenate(want, axis=-1), rtol=2e-5)
