#!/usr/bin/env python3
"""Golden vectors for the notebook path from the reference's OWN function.

Runs only in the dev container (needs /root/reference).  The source of ``get_eig_from_qkv_att_softmax`` is taken from
cell 13 of ``/root/reference/notebooks/lm_eigvals.ipynb`` at generation time and executed unmodified (it needs torch,
einops and numpy only); nothing of it is stored in the repository -- the committed artefact is
tests/golden/notebook.npz = seeded q / k and the eta the reference's function returned.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_notebook.py
"""
import json
import os

import numpy as np
import torch

NB = "/root/reference/notebooks/lm_eigvals.ipynb"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_function():
    nb = json.load(open(NB))
    for cell in nb["cells"]:
        src = "".join(cell["source"])
        if cell["cell_type"] == "code" and "def get_eig_from_qkv_att_softmax" in src:
            ns = {"torch": torch}
            exec(compile(src, NB + ":cell13", "exec"), ns)
            return ns["get_eig_from_qkv_att_softmax"]
    raise RuntimeError("cell with get_eig_from_qkv_att_softmax not found")


def main():
    fn = load_reference_function()
    rng = np.random.default_rng(20261004)
    out = {}
    for name, (B, L, H, hd, scale) in {"a": (2, 33, 3, 8, 0.4), "b": (1, 70, 2, 16, 1.0), "c": (3, 17, 4, 4, 2.5)}.items():
        q = (rng.standard_normal((B, L, H, hd)) * scale).astype(np.float32)
        k = (rng.standard_normal((B, L, H, hd)) * scale).astype(np.float32)
        v = rng.standard_normal((B, L, H, hd)).astype(np.float32)
        eta = fn(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v))
        out[f"q_{name}"], out[f"k_{name}"], out[f"eta_{name}"] = q, k, np.asarray(eta)
    np.savez_compressed(os.path.join(OUT, "notebook.npz"), **out)
    print("wrote notebook.npz", {k_: v_.shape for k_, v_ in out.items()})


if __name__ == "__main__":
    main()
