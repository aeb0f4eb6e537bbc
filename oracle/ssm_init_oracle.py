"""Independent CPU restatement of the reference's SSM initialisation (TEST INFRASTRUCTURE, never imported by the product).

``get_init_layers_ssm`` (analysis/eval_eig.py:207-239) builds the flax model and calls ``model.init`` with
``PRNGKey(seed) -> split -> init_rng -> split -> rng`` (:208, :209, :227, :228); the layers' ``seq`` sub-trees are what
``get_eigvals_ssm`` reads.  Only three random draws reach the eigenvalues, all ``jax.random.uniform``:
  LRU  theta_log, nu_log   models/lru.py:24-33 (formulas), :55-60 (parameter order: theta_log first, nu_log second)
  S4   log_step            models/common.py:244-265 (``log_step_initializer``), models/s4.py:129 (8th ``self.param`` call:
                           Lambda_re, Lambda_im, P, B, C, D, log_step -> counter 7), :183-189 (vmap, split_rngs)
  S5   log_step            models/common.py:268-284 (``init_log_steps``: split per state, uniform on the sub-key),
                           models/s5.py:218-220
and the deterministic HiPPO-LegS DPLR parts follow models/common.py:180-241, models/s4.py:192-215, models/s5.py:253-296.

Written independently of the product's ``analysis/jax_prng.py`` (scalar Python integers instead of NumPy vectors, its own
key bookkeeping) so that ``eig_init`` of the SSM branch can be checked against something other than the code under test.
jax 0.4.25's default PRNG (threefry2x32, non-partitionable) and flax 0.8.1's ``LazyRng`` key derivation are restated from
their published sources; neither package is installed here, so the chain is PINNED ONLY by the Threefry-2x32 known-answer
vectors (Salmon et al., SC'11) and jax's documented ``PRNGKey(0)`` examples -- see tests/test_host_logic.py."""
import hashlib
import math
import struct

import numpy as np

M32 = 0xFFFFFFFF
_ROTATIONS = (13, 15, 26, 6, 17, 29, 16, 24)


def threefry2x32_pair(k0, k1, c0, c1):
    """One Threefry-2x32-20 block on Python integers."""
    ks = (k0 & M32, k1 & M32, (k0 ^ k1 ^ 0x1BD11BDA) & M32)
    x0 = (c0 + ks[0]) & M32
    x1 = (c1 + ks[1]) & M32
    for rnd in range(20):
        r = _ROTATIONS[(rnd % 4) + 4 * ((rnd // 4) % 2)]
        x0 = (x0 + x1) & M32
        x1 = ((x1 << r) | (x1 >> (32 - r))) & M32
        x1 ^= x0
        if rnd % 4 == 3:
            inj = rnd // 4 + 1
            x0 = (x0 + ks[inj % 3]) & M32
            x1 = (x1 + ks[(inj + 1) % 3] + inj) & M32
    return x0, x1


def _bits(key, counts):
    """jax._src.prng.threefry_2x32 on a flat list of counters: pad to even length, first half = word 0, second = word 1."""
    counts = list(counts)
    odd = len(counts) % 2
    if odd:
        counts.append(0)
    half = len(counts) // 2
    lo, hi = [], []
    for a, b in zip(counts[:half], counts[half:]):
        y0, y1 = threefry2x32_pair(key[0], key[1], a, b)
        lo.append(y0)
        hi.append(y1)
    out = lo + hi
    return out[:-1] if odd else out


def key_of(seed):
    return ((int(seed) >> 32) & M32, int(seed) & M32)


def split(key, num=2):
    b = _bits(key, range(2 * num))
    return [(b[2 * i], b[2 * i + 1]) for i in range(num)]


def fold_in(key, data):
    b = _bits(key, key_of(int(data) & M32))
    return (b[0], b[1])


def uniform32(key, n):
    """jax.random.uniform(key, (n,)) as float32: 23 mantissa bits of a float in [1, 2), minus 1, clamped at 0."""
    out = np.empty(n, dtype=np.float32)
    for idx, word in enumerate(_bits(key, range(n))):
        f = struct.unpack("<f", struct.pack("<I", (word >> 9) | 0x3F800000))[0]
        out[idx] = max(np.float32(0.0), np.float32(f) - np.float32(1.0))
    return out


def param_key(rng, path, counter):
    """flax.core.scope.LazyRng: fold the first four bytes of SHA-1(path names ... counter bytes) into the stream's rng."""
    sha = hashlib.sha1()
    for name in path:
        sha.update(name.encode("utf-8"))
    sha.update(int(counter).to_bytes((int(counter).bit_length() + 7) // 8, "big"))
    return fold_in(rng, int.from_bytes(sha.digest()[:4], "big"))


def _params_rng(seed):
    init_rng = split(key_of(seed), 2)[0]            # analysis/eval_eig.py:208-209
    return split(init_rng, 2)[0]                    # :227


def _seq_path(layer):
    return ("encoder", "layers_%d" % layer, "seq")


def hippo_dplr(n):
    """(Lambda, P, B) of models/common.py:215-241."""
    q = np.sqrt(1 + 2 * np.arange(n))
    a = -(np.tril(np.outer(q, q)) - np.diag(np.arange(n)))
    p = np.sqrt(np.arange(n) + 0.5)
    b = np.sqrt(2 * np.arange(n) + 1.0)
    s = a + np.outer(p, p)
    lam_im, v = np.linalg.eigh(s * -1j)
    lam = np.mean(np.diagonal(s)) * np.ones(n) + 1j * lam_im
    return lam, v.conj().T @ p, v.conj().T @ b


def init_layers(kind, seed, cfg):
    """The per-layer ``seq`` parameter trees get_init_layers_ssm returns (only the members that reach the eigenvalues)."""
    f32 = np.float32
    n, h, nl = cfg["state_dim"], cfg["hidden_dim"], cfg["num_layers"]
    dt_min, dt_max = cfg.get("dt_min", 0.001), cfg.get("dt_max", 0.1)
    span, lo = math.log(dt_max) - math.log(dt_min), math.log(dt_min)
    rng = _params_rng(seed)
    layers = []
    for li in range(nl):
        path = _seq_path(li)
        if kind == "lru":
            r_min, r_max, max_phase = cfg.get("r_min", 0.0), cfg.get("r_max", 1.0), cfg.get("max_phase", 6.28)
            u_theta = uniform32(param_key(rng, path, 1), n)
            u_nu = uniform32(param_key(rng, path, 2), n)
            nu = np.log(f32(-0.5) * np.log(u_nu * f32(r_max ** 2 - r_min ** 2) + f32(r_min ** 2)))
            theta = np.log(f32(max_phase) * u_theta)
            layers.append({"nu_log": nu.astype(f32), "theta_log": theta.astype(f32)})
        elif kind == "s4":
            lam, p, b = hippo_dplr(n)
            cols = split(rng, h)                                     # nn.vmap(split_rngs={"params": True})
            u = np.array([uniform32(param_key(cols[c], path, 7), 1)[0] for c in range(h)], dtype=f32)
            layers.append({"Lambda_re": np.repeat(lam.real[:, None], h, 1), "Lambda_im": np.repeat(lam.imag[:, None], h, 1),
                           "P": np.repeat(p[:, None], h, 1), "B": np.repeat(b[:, None], h, 1),
                           "log_step": (u * f32(span) + f32(lo)).astype(f32)[None, :]})
        elif kind == "s5":
            blocks = cfg.get("num_blocks", 8)
            bs = n // blocks
            lam, _, _ = hippo_dplr(bs)
            states = n
            if cfg.get("conj_sym", True):
                bs, states = bs // 2, n // 2
            lam = np.tile(lam[:bs], blocks)
            two_c = cfg.get("bidirectional", False) and cfg.get("C_init", "lecun_normal") != "complex_normal"
            key = param_key(rng, path, 7 if two_c else 6)
            u = np.empty(states, dtype=f32)
            for s_ in range(states):
                key, sub = split(key, 2)
                u[s_] = uniform32(sub, 1)[0]
            layers.append({"Lambda_re": lam.real, "Lambda_im": lam.imag,
                           "log_step": (u * f32(span) + f32(lo)).astype(f32)[:, None]})
        else:
            raise RuntimeError("{0} is not a valid model option".format(kind))
    return layers
