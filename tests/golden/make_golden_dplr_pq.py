#!/usr/bin/env python3
"""Golden vectors for the GENERAL form of discrete_DPLR -- separate P and Q, Lambda as given (the function itself does
not clip; its caller does, analysis/eval_eig.py:288) -- from the reference's own function, loaded exactly as
make_golden.py loads it (by path, unmodified, jax.numpy aliased to NumPy: complex128 instead of JAX's complex64).
The reference's only caller passes Q = P (:294); these cases pin the other argument.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_dplr_pq.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference  # noqa: E402


def main():
    ref = load_reference()
    rng = np.random.default_rng(2542)
    out = {}
    for i, n in enumerate((8, 13)):
        lam = -np.abs(rng.standard_normal(n)) * 0.5 + 0.05 + 1j * rng.standard_normal(n) * 3.0   # some Re > -1e-4: NOT clipped here
        lam[1] = 0.02 + lam[1].imag * 1j                       # surely above the caller's clip level of -1e-4
        p = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7
        q = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.7
        b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        c = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        step, L = float(np.exp(rng.uniform(np.log(1e-3), np.log(1e-1)))), 16
        ab, bb, cb = ref.discrete_DPLR(lam, p, q, b, c, step, L)
        for k, v in dict(Lambda=lam, P=p, Q=q, B=b, C=c, step=step, L=L, Ab=np.asarray(ab), Bb=np.asarray(bb),
                         Cb=np.asarray(cb)).items():
            out[f"c{i}_{k}"] = v
    np.savez(os.path.join(HERE, "dplr_pq.npz"), **out)
    print("wrote dplr_pq.npz", {k: np.shape(v) for k, v in out.items() if k.startswith("c0_")})


if __name__ == "__main__":
    main()
