"""GPU tests of the notebook path: softmax eta from hooked q / k, shards, resume, collection."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_softmax_eta_from_qk_golden(gpu_ops, golden, case):
    g = golden("notebook")
    eta = gpu_ops.softmax_eta_from_qk(dev(g[f"q_{case}"]), dev(g[f"k_{case}"])).cpu().numpy()
    np.testing.assert_allclose(eta, g[f"eta_{case}"][..., 0], rtol=3e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_softmax_eta_from_qk_grouped_query(gpu_ops, dtype):
    """H = 8 query heads on 2 key heads: query head h reads key head h // 4; fp16 inputs are widened to fp32."""
    rng = np.random.default_rng(4)
    B, L, H, HKV, hd = 2, 45, 8, 2, 12
    q = torch.from_numpy(rng.standard_normal((B, L, H, hd)).astype(np.float32)).to(dtype)
    k = torch.from_numpy(rng.standard_normal((B, L, HKV, hd)).astype(np.float32)).to(dtype)
    got = gpu_ops.softmax_eta_from_qk(q.cuda(), k.cuda()).cpu().numpy()
    want = O.eta_softmax_from_qk(q.float().numpy(), np.repeat(k.float().numpy(), H // HKV, axis=2))[..., 0]
    np.testing.assert_allclose(got, want, rtol=3e-5)
    with pytest.raises(ValueError):
        gpu_ops.softmax_eta_from_qk(q.cuda(), k[:, :, :1].repeat(1, 1, 3, 1).cuda())      # 8 % 3 != 0


class _Attn(nn.Module):
    def __init__(self, d, heads, kv_heads, hd):
        super().__init__()
        self.q_proj = nn.Linear(d, heads * hd, bias=False)
        self.k_proj = nn.Linear(d, kv_heads * hd, bias=False)
        self.v_proj = nn.Linear(d, kv_heads * hd, bias=False)


class _Layer(nn.Module):
    def __init__(self, d, heads, kv_heads, hd):
        super().__init__()
        self.self_attn = _Attn(d, heads, kv_heads, hd)
        self.mix = nn.Linear(d, d)

    def forward(self, x):
        a = self.self_attn
        _ = a.q_proj(x), a.k_proj(x), a.v_proj(x)                     # the hooks see these
        return x + torch.tanh(self.mix(x))


class _Inner(nn.Module):
    def __init__(self, d, heads, kv_heads, hd, layers, vocab):
        super().__init__()
        self.embed = nn.Embedding(vocab, d)
        self.layers = nn.ModuleList([_Layer(d, heads, kv_heads, hd) for _ in range(layers)])


class FakeLM(nn.Module):
    """The module structure the notebook relies on: model.model.layers[i].self_attn.{q,k,v}_proj and model.config."""

    def __init__(self, d=16, heads=4, kv_heads=2, hd=4, layers=3, vocab=20):
        super().__init__()
        self.model = _Inner(d, heads, kv_heads, hd, layers, vocab)
        self.config = type("Cfg", (), {"num_attention_heads": heads, "num_key_value_heads": kv_heads})()

    def forward(self, ids):
        x = self.model.embed(ids)
        for layer in self.model.layers:
            x = layer(x)
        return x


def test_lm_eigvals_run_resume_collect(pkg, tmp_path):
    lm = sys.modules[pkg.__name__ + ".analysis.lm_eigvals"] if pkg.__name__ + ".analysis.lm_eigvals" in sys.modules \
        else __import__("importlib").import_module(pkg.__name__ + ".analysis.lm_eigvals")
    torch.manual_seed(0)
    model = FakeLM().cuda()
    rng = np.random.default_rng(1)
    batches = [(torch.from_numpy(rng.integers(0, 20, (2, 19))), None, None) for _ in range(3)]
    out_dir = str(tmp_path / "eigs")
    assert lm.run(model, batches, out_dir) == 3
    assert sorted(os.listdir(out_dir)) == ["eigs_0.pkl", "eigs_1.pkl", "eigs_2.pkl"]
    assert lm.last_finished_shard(out_dir) == 2
    assert lm.run(model, batches, out_dir) == 0                       # everything already there (cell 15 resume)
    os.remove(os.path.join(out_dir, "eigs_2.pkl"))
    assert lm.run(model, batches, out_dir) == 1
    all_eigs = lm.collect(out_dir)
    assert all_eigs.shape == (6, 18, 4, 3) and all_eigs.dtype == np.float64
    assert os.path.exists(os.path.join(out_dir, "all_eigs.npy"))

    # expected values: the oracle on q / k recomputed from the model (layer 1 of batch 1), GQA 4 heads on 2
    with torch.no_grad():
        x = model.model.embed(batches[1][0].cuda())
        x = model.model.layers[0](x)
        att = model.model.layers[1].self_attn
        x1 = x                                                        # input of layer 1
        q = att.q_proj(x1).view(2, 19, 4, 4).cpu().numpy()
        k = att.k_proj(x1).view(2, 19, 2, 4).cpu().numpy()
    want = O.eta_softmax_from_qk(q, np.repeat(k, 2, axis=2))[..., 0]
    np.testing.assert_allclose(all_eigs[2:4, :, :, 1], want, rtol=3e-5)
    pct = lm.threshold_analysis(all_eigs, O.THRESHOLDS_RADIUS)
    np.testing.assert_array_equal(pct, O.threshold_analysis(all_eigs, O.THRESHOLDS_RADIUS, 3, 4, 6))
    got = lm.get_eig_from_qkv_att_softmax(torch.from_numpy(q), torch.from_numpy(np.repeat(k, 2, axis=2)))
    assert got.shape == (2, 18, 4, 1)
