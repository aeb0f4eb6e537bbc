#!/usr/bin/env python3
"""Golden vectors for the block forward from the reference's OWN torch modules.

Runs only in the dev container (needs /root/reference).  The reference's ``models/common.py``,
``models/attention.py``, ``models/norm_attention.py`` and ``models/transformer.py`` are loaded *by file path,
unmodified*, as members of a placeholder ``models`` package, with inert ``sys.modules`` stubs for the third-party
packages that are not installed here (jax, flash_attn, mamba_ssm, fla) -- none of which the plain-torch code paths
exercised below (use_flash = False) touch.  ``models/mamba.py`` cannot run here at all (its scan is mamba_ssm's
Triton kernel), so the Mamba-2 block has no golden: oracle/blocks_oracle.py says "parity unpinned" for it.

Committed artefact: tests/golden/blocks.npz = seeded inputs, the modules' state_dicts (these pin the parameter
names of the drop-in models) and the outputs the reference returned.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_blocks.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/models"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(pkg, name):
    spec = importlib.util.spec_from_file_location(f"models.{name}", os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"models.{name}"] = mod
    spec.loader.exec_module(mod)
    setattr(pkg, name, mod)
    return mod


def load_reference_models():
    sys.dont_write_bytecode = True
    jnp = _stub("jax.numpy")
    jnp.linalg = _stub("jax.numpy.linalg", eigh=np.linalg.eigh)
    _stub("jax.random")
    _stub("jax.nn.initializers", lecun_normal=None)
    _stub("jax.nn", initializers=sys.modules["jax.nn.initializers"])
    _stub("jax", numpy=jnp, random=sys.modules["jax.random"], nn=sys.modules["jax.nn"])
    _stub("flash_attn", flash_attn_qkvpacked_func=None)
    _stub("mamba_ssm")
    _stub("mamba_ssm.ops")
    _stub("mamba_ssm.ops.selective_scan_interface", selective_scan_fn=None)
    _stub("mamba_ssm.ops.triton")
    _stub("mamba_ssm.ops.triton.ssd_combined", mamba_chunk_scan_combined=None, mamba_split_conv1d_scan_combined=None)
    _stub("fla")
    _stub("fla.ops")
    _stub("fla.ops.simple_gla", chunk_simple_gla=None, fused_recurrent_simple_gla=None)
    pkg = _stub("models")
    pkg.__path__ = []
    common = _load(pkg, "common")
    for n in ("MATCH", "MLP", "GLU", "LAMBDA", "ClassifierHead", "TokenEmbeddings"):
        setattr(pkg, n, getattr(common, n))
    att = _load(pkg, "attention")
    pkg.MHA = att.MHA
    natt = _load(pkg, "norm_attention")
    pkg.MHNA = natt.MHNA
    tr = _load(pkg, "transformer")
    pkg.Transformer = tr.Transformer
    return pkg


BASE = dict(input_dim=6, output_dim=5, num_layers=2, hidden_dim=32, embedding=False, vocab_size=50, max_pos_embed=0,
            pooling="none", dual=False, classifier=False, mixer_dim=48, norm="layer", dropout=0.1, state_dim=16,
            num_heads=2, att_dropout=0.1, mixer="mlp", use_flash=False, use_gate=False, conv_type="full",
            attention_fn="lin-attention", dim_conv=0)

CASES = {
    "lin": dict(),
    "lin_conv_gate": dict(dim_conv=3, use_gate=True, mixer="glu", num_heads=1, state_dim=24),
    "sm": dict(attention_fn="sm-attention", mixer="hybrid", embedding=True, max_pos_embed=64, classifier=True,
               pooling="mean", num_heads=4, state_dim=32),
    "sm_conv": dict(attention_fn="sm-attention", dim_conv=4, conv_type="partial", state_dim=8),
    "norm_exp": dict(attention_fn="norm-attention", mode="attention", norm_fn="exp", approx_fn="none", scale_B=True,
                     offset=True, offset_init="uniform", learn_A=False, dim_conv=3, conv_type="partial", mixer="none",
                     use_gate=True),
    "norm_softplus": dict(attention_fn="norm-attention", mode="attention", norm_fn="softplus", approx_fn="elu",
                          scale_B=False, offset=False, offset_init="uniform", learn_A=False, dim_conv=0, num_heads=4),
    "norm_sigmoid": dict(attention_fn="norm-attention", mode="attention", norm_fn="sigmoid", approx_fn="none",
                         scale_B=False, offset=True, offset_init="exp", learn_A=False, dim_conv=0, num_layers=1),
    "norm_elu": dict(attention_fn="norm-attention", mode="attention", norm_fn="elu", approx_fn="elu", scale_B=True,
                     offset=False, offset_init="uniform", learn_A=False, dim_conv=2, conv_type="full", num_layers=1),
}


def npy(t):
    return t.detach().cpu().numpy()


def main():
    pkg = load_reference_models()
    out = {}
    rng = np.random.default_rng(20261004)

    # ---- inner mixers on raw tensors ----
    B, L, H, d, dv = 3, 37, 2, 8, 12
    qk = torch.from_numpy(rng.standard_normal((B, L, 2, H, d)).astype(np.float32))
    v = torch.from_numpy(rng.standard_normal((B, L, H, dv)).astype(np.float32))
    n = torch.from_numpy((rng.standard_normal((B, L, H)) * 0.7).astype(np.float32))
    out["inner_qk"], out["inner_v"], out["inner_n"] = npy(qk), npy(v), npy(n)
    with torch.no_grad():
        out["inner_lin"] = npy(pkg.attention.SelfLinAttention(0.0)(qk, v))
        out["inner_sm"] = npy(pkg.attention.SelfAttention(0.0)(qk, v))
        for nf in ("exp", "elu", "softplus", "sigmoid"):
            for af in ("none", "elu"):
                for sb in (False, True):
                    m = pkg.norm_attention.SelfNormAttention(nf, af, num_heads=H, scale_B=sb, offset=True,
                                                             offset_init="uniform", dropout=0.0)
                    out[f"inner_norm_{nf}_{af}_{int(sb)}"] = npy(m(qk, v, n))
                    out[f"inner_norm_offset_{nf}"] = npy(m.offset)

    # ---- whole Transformer stacks (eval mode), block by block ----
    for name, over in CASES.items():
        cfg = dict(BASE)
        cfg.update(over)
        torch.manual_seed(1234)
        model = pkg.Transformer(dict(cfg))
        if cfg["embedding"]:
            model.encoder.device = "cpu"                       # the default 'cuda' only places position ids
            x = torch.from_numpy(rng.integers(0, cfg["vocab_size"], size=(4, 29)))
        else:
            x = torch.from_numpy(rng.standard_normal((4, 29, cfg["input_dim"])).astype(np.float32))
        # perturb the parameters so that nothing sits at a special init value (biases, LayerNorm, offsets)
        with torch.no_grad():
            for p_ in model.parameters():
                p_.add_(torch.from_numpy((rng.standard_normal(tuple(p_.shape)) * 0.05).astype(np.float32)))
        model.eval()
        with torch.no_grad():
            h = model.dropout(model.encoder(x))
            out[f"{name}/h0"] = npy(h)
            for i, layer in enumerate(model.layers):
                h = layer(h)
                out[f"{name}/h{i + 1}"] = npy(h)
            out[f"{name}/y"] = npy(model(x))
        out[f"{name}/x"] = npy(x)
        out[f"{name}/cfg"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
        sd = model.state_dict()
        out[f"{name}/keys"] = np.frombuffer(json.dumps(list(sd.keys())).encode(), dtype=np.uint8)
        for k_, t in sd.items():
            out[f"{name}/sd/{k_}"] = npy(t)
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **out)
    print("wrote", os.path.join(OUT, "blocks.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
