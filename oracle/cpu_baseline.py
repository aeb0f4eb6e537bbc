#!/usr/bin/env python3
"""CPU baseline of config C2 on all the host cores bench.py may use (TEST / MEASUREMENT INFRASTRUCTURE).

Generates the C2 workload's matrices (same construction as bench.py: S5-style eigenvalues exp((-0.5 + i w) dt) in
conjugate pairs under a non-normal orthogonal similarity) with NumPy and times stacked ``np.linalg.eigvals`` -- LAPACK
dgeev, the third-party arithmetic behind analysis/eval_eig.py:296 -- over a process pool, one BLAS thread per process.
Run as a separate process by bench.py *before* it touches the GPU, so forking the pool is safe.

    python oracle/cpu_baseline.py --procs 16 --seconds 10  ->  one JSON line
"""
import argparse
import json
import os
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np  # noqa: E402


def gen(cnt, n, seed):
    rng = np.random.default_rng(seed)
    npair = n // 2
    om = rng.uniform(0, 40, (cnt, npair))
    dt = np.exp(rng.uniform(np.log(1e-3), np.log(1e-1), (cnt, npair)))
    lam = np.exp((-0.5 + 1j * om) * dt)
    a = np.triu(rng.standard_normal((cnt, n, n)), 2) * (0.1 / np.sqrt(n))
    idx = np.arange(npair) * 2
    a[:, idx, idx] = lam.real
    a[:, idx + 1, idx + 1] = lam.real
    a[:, idx, idx + 1] = lam.imag
    a[:, idx + 1, idx] = -lam.imag
    for _ in range(3):
        u = rng.standard_normal((cnt, n, 1))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        a = a - 2.0 * u @ (u.transpose(0, 2, 1) @ a)
        a = a - 2.0 * (a @ u) @ u.transpose(0, 2, 1)
    return a


def work(args):
    seed, cnt, n, budget = args
    a = gen(cnt, n, seed)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:       # keep solving the same block until the budget is spent
        np.linalg.eigvals(a)
        done += cnt
    return done, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--block", type=int, default=512)
    a = ap.parse_args()
    import multiprocessing as mp
    with mp.get_context("fork").Pool(a.procs) as pool:
        t0 = time.perf_counter()
        res = pool.map(work, [(64 + p, a.block, a.n, a.seconds) for p in range(a.procs)])
        wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    print(json.dumps({"value": total / busy, "unit": "spectra/s", "cores": a.procs, "kind": "port",
                      "sample": "%d processes x blocks of %d C2-style 64x64 matrices (NumPy-generated, same construction), "
                                "stacked np.linalg.eigvals, 1 BLAS thread each, %.1f s" % (a.procs, a.block, busy),
                      "wall_s": wall, "host_cpu_count": os.cpu_count()}))


if __name__ == "__main__":
    main()
