"""Notebook path: softmax-attention eta of a pretrained causal LM from forward hooks on its
q / k projections -- the code path of the reference's ``notebooks/lm_eigvals.ipynb`` (cells 12-25) as a module.

    attach_qkv_hooks(model)              cell 12   hooks on layer.self_attn.{q,k,v}_proj, outputs kept per layer
    get_eig_from_qkv_att_softmax(q,k,v)  cell 13   (B, L-1, H, 1) float64; arithmetic on the GPU (egs_softmax_eta_qk)
    last_finished_shard(dir)             cell 15   resume: index of the last eigs_<i>.pkl present
    run(model, loader, out_dir, ...)     cell 16   per-batch shards ``eigs_<batch>.pkl`` (dill when available)
    collect(out_dir)                     cells 19-20  concatenate shards -> all_eigs.npy
    threshold_analysis(eig, thresholds)  cell 22   shapes inferred from the array (device histogram kernel)

Differences, all documented: the hooked tensors stay on the GPU (the notebook copies them to the host); grouped-query
attention is handled by repeating each key head H / H_kv times (the notebook's einsum silently requires H_kv == H);
scores are computed in fp32 whatever the model's dtype (the notebook multiplies fp16 tensors).  Loading the model /
tokenizer / dataset (cells 6-11: from_pretrained, load_dataset) needs the network and is the caller's business.
"""
import glob
import os
import pickle
import re
from collections import defaultdict

import numpy as np
import torch

from .. import ops

try:                                            # the notebook pickles with dill; plain pickle reads / writes the same arrays
    import dill as _pickler
except ImportError:                             # pragma: no cover
    _pickler = pickle

_SHARD = re.compile(r"^eigs_(\d+)\.pkl$")


def attach_qkv_hooks(model, layers=None):
    """Forward hooks that keep the outputs of every layer's q_proj / k_proj / v_proj (cell 12).  Returns
    (cache, handles): cache[layer_idx]["Q" | "K" | "V"] is the latest output (on its device, detached)."""
    cache = defaultdict(dict)

    def make_hook(idx, name):
        def hook(module, inputs, output):
            cache[idx][name] = output.detach()
        return hook

    layers = model.model.layers if layers is None else layers
    handles = []
    for idx, layer in enumerate(layers):
        att = layer.self_attn
        handles.append(att.q_proj.register_forward_hook(make_hook(idx, "Q")))
        handles.append(att.k_proj.register_forward_hook(make_hook(idx, "K")))
        handles.append(att.v_proj.register_forward_hook(make_hook(idx, "V")))
    return cache, handles


def eig_from_qk_dev(q, k):
    """Device result [B, L-1, H] float64 for q [B, L, H, hd], k [B, L, H_kv, hd]."""
    if not q.is_cuda:
        q, k = q.to("cuda"), k.to("cuda")
    return ops.softmax_eta_from_qk(q, k)


def get_eig_from_qkv_att_softmax(q, k, v=None):
    """(batch, seq_len-1, num_heads, 1) float64 ndarray, as the notebook's function of the same name (cell 13); ``v`` is
    accepted and ignored like there."""
    return np.expand_dims(eig_from_qk_dev(q, k).cpu().numpy(), axis=-1)


def last_finished_shard(out_dir):
    """Largest i with ``eigs_<i>.pkl`` in out_dir, or -1 (cell 15)."""
    last = -1
    if os.path.isdir(out_dir):
        for name in os.listdir(out_dir):
            m = _SHARD.match(name)
            if m:
                last = max(last, int(m.group(1)))
    return last


def run(model, loader, out_dir, num_heads=None, num_kv_heads=None, resume=True, layers=None, on_batch=None):
    """Cell 16 without the loss bookkeeping: for every batch (X, y, aux) of ``loader`` run the model, turn the hooked
    q / k of every layer into eta and write ``out_dir/eigs_<batch>.pkl`` = float64 (B, L-1, H, layers).  Batches whose
    shard already exists are skipped when ``resume``.  Returns the number of shards written."""
    cfg = getattr(model, "config", None)
    num_heads = num_heads or cfg.num_attention_heads
    num_kv_heads = num_kv_heads or getattr(cfg, "num_key_value_heads", num_heads)
    os.makedirs(out_dir, exist_ok=True)
    skip_upto = last_finished_shard(out_dir) if resume else -1
    cache, handles = attach_qkv_hooks(model, layers)
    written = 0
    try:
        model.eval()
        with torch.inference_mode():
            for batch, item in enumerate(loader):
                X = item[0] if isinstance(item, (tuple, list)) else item
                if batch <= skip_upto:
                    continue
                device = next(model.parameters()).device
                output = model(X.to(device))
                per_layer = []
                for idx in range(len(cache)):
                    Q, K = cache[idx]["Q"], cache[idx]["K"]
                    hd = Q.shape[-1] // num_heads
                    kv_hd = K.shape[-1] // num_kv_heads
                    per_layer.append(eig_from_qk_dev(Q.view(Q.shape[0], Q.shape[1], num_heads, hd),
                                                     K.view(K.shape[0], K.shape[1], num_kv_heads, kv_hd)))
                eig = torch.stack(per_layer, dim=-1).cpu().numpy()          # (B, L-1, H, layers)
                with open(os.path.join(out_dir, f"eigs_{batch}.pkl"), "wb") as f:
                    _pickler.dump(eig, f)
                written += 1
                if on_batch is not None:
                    on_batch(batch, output, eig)
    finally:
        for h in handles:
            h.remove()
    return written


def collect(out_dir, save=True):
    """Concatenate all shards along the batch axis (cells 19-20); writes ``all_eigs.npy`` when ``save``."""
    files = sorted(glob.glob(os.path.join(out_dir, "eigs_*.pkl")),
                   key=lambda p: int(_SHARD.match(os.path.basename(p)).group(1)))
    arrays = []
    for name in files:
        with open(name, "rb") as f:
            arr = _pickler.load(f)
        if not isinstance(arr, np.ndarray):
            raise TypeError(f"{name} does not contain a numpy ndarray")
        arrays.append(arr)
    if not arrays:
        raise FileNotFoundError(f"no eigs_*.pkl shards in {out_dir}")
    all_eigs = np.concatenate(arrays, axis=0)
    if save:
        np.save(os.path.join(out_dir, "all_eigs.npy"), all_eigs)
    return all_eigs


def threshold_analysis(eig_val, thresholds):
    """Cell 22: (len(thresholds)+1, B, H, layers) percentages with the sizes taken from the array."""
    t = eig_val if isinstance(eig_val, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(eig_val))
    if not t.is_cuda:
        t = t.to("cuda")
    return ops.threshold_hist(t, np.asarray(thresholds, dtype=np.float64).flatten(), "value").cpu().numpy()
