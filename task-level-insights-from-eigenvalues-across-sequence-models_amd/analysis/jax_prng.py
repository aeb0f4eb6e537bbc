"""JAX-free restatement of the random draws behind the reference's SSM initialisation (analysis/eval_eig.py:207-239).

``get_init_layers_ssm`` builds ``BatchClassificationModel`` with flax and calls ``model.init`` -- the only random parts
that reach the eigenvalues are three ``jax.random.uniform`` draws: LRU ``theta_log`` / ``nu_log`` (models/lru.py:26-33,
:55-60), S4 ``log_step`` per feature column (models/s4.py:129, under ``nn.vmap(split_rngs={"params": True})``,
models/s4.py:183-189) and S5 ``log_step`` (models/s5.py:218-220 through ``init_log_steps``, models/common.py:263-284).
Everything they need is public and small, and is restated here with NumPy so that the analysis can run where jax / flax
are not installed:

* ``threefry2x32`` -- the Threefry-2x32 block function (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
  SC'11; 20 rounds), pinned by its published known-answer vectors (tests/test_host_logic.py);
* ``prng_key / split / fold_in / uniform`` -- jax 0.4.25's default (non-partitionable) threefry PRNG: key = (0, seed),
  ``split`` = block function on ``iota(2 num)``, ``fold_in`` = block function on the key of the datum, 32 random bits
  -> float32 in [0, 1) by filling the mantissa of 1.0;
* ``flax_param_key`` -- flax 0.8.1's parameter keys: ``Scope.make_rng`` folds the SHA-1 of (module path names + the
  scope's running parameter counter) into the rng the user passed (``LazyRng`` / ``_fold_in_static``,
  flax/core/scope.py); a lifted ``nn.vmap`` with ``split_rngs`` splits the base key and keeps that suffix.

PARITY UNPINNED beyond the Threefry vectors: neither jax nor flax is installed in the build or test environment, so the
key-derivation chain (module path names, counter values, split-before-fold order) is restated from the published
sources and could not be compared with a real ``model.init``.  If it is off, the initial spectra are still draws from
the reference's distributions (uniform in the same ranges), exactly as with any other generator."""
import hashlib

import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return ((x << _U32(r)) | (x >> _U32(32 - r))).astype(_U32)


def threefry2x32(key, x0, x1):
    """Threefry-2x32, 20 rounds: key (k0, k1), counter words x0, x1 (uint32 arrays of equal shape)."""
    k0, k1 = _U32(key[0]), _U32(key[1])
    ks = (k0, k1, _U32(k0 ^ k1 ^ _U32(0x1BD11BDA)))
    x0 = np.asarray(x0, dtype=_U32).copy()
    x1 = np.asarray(x1, dtype=_U32).copy()
    with np.errstate(over="ignore"):
        x0 = (x0 + ks[0]).astype(_U32)
        x1 = (x1 + ks[1]).astype(_U32)
        for group in range(5):
            for r in _ROT[group % 2]:
                x0 = (x0 + x1).astype(_U32)
                x1 = _rotl(x1, r)
                x1 = (x1 ^ x0).astype(_U32)
            x0 = (x0 + ks[(group + 1) % 3]).astype(_U32)
            x1 = (x1 + ks[(group + 2) % 3] + _U32(group + 1)).astype(_U32)
    return x0, x1


def _block_on_counts(key, counts):
    """jax._src.prng.threefry_2x32: the count array is padded to even length, its two halves are the counter words,
    the outputs are concatenated back (and the pad dropped)."""
    counts = np.asarray(counts, dtype=_U32).ravel()
    odd = counts.size % 2
    if odd:
        counts = np.concatenate([counts, np.zeros(1, dtype=_U32)])
    half = counts.size // 2
    y0, y1 = threefry2x32(key, counts[:half], counts[half:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def prng_key(seed):
    """jax.random.PRNGKey(seed) for the default threefry implementation: (seed >> 32, seed & 0xffffffff)."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=_U32)


def split(key, num=2):
    """jax.random.split: rows of threefry(key, iota(2 num)).reshape(num, 2)."""
    return _block_on_counts(key, np.arange(2 * num, dtype=_U32)).reshape(num, 2)


def fold_in(key, data):
    """jax.random.fold_in: threefry(key, PRNGKey(data))."""
    return _block_on_counts(key, prng_key(int(data) & 0xFFFFFFFF))


def random_bits(key, n):
    return _block_on_counts(key, np.arange(n, dtype=_U32))


def uniform(key, shape):
    """jax.random.uniform(key, shape) in float32: 23 random mantissa bits of 1.0, minus 1."""
    n = int(np.prod(shape)) if len(shape) else 1
    bits = random_bits(key, n)
    floats = ((bits >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
    return np.maximum(np.float32(0.0), floats).reshape(shape)


def flax_param_key(base_key, path, counter):
    """Key that ``self.param`` number ``counter`` (1-based, per scope) receives in the module at ``path`` (tuple of
    names) when the user passed ``base_key`` for the "params" stream: fold_in(base, first 4 bytes of SHA-1 of the
    concatenated path names and the counter's big-endian bytes)."""
    m = hashlib.sha1()
    for x in tuple(path) + (int(counter),):
        if isinstance(x, str):
            m.update(x.encode("utf-8"))
        else:
            m.update(x.to_bytes((x.bit_length() + 7) // 8, byteorder="big"))
    return fold_in(base_key, int.from_bytes(m.digest()[:4], byteorder="big"))


def params_rng(seed):
    """The "params" rng of analysis/eval_eig.py:208-230: PRNGKey(seed) -> split -> init_rng -> split -> rng."""
    init_rng = split(prng_key(seed), 2)[0]
    return split(init_rng, 2)[0]


def ssm_path(layer):
    return ("encoder", "layers_%d" % layer, "seq")          # models/jax_layers.py:109-123, :30-33, :193-205


def lru_uniforms(seed, layer, n):
    """(u_theta, u_nu): models/lru.py:55-60 -- theta_log is parameter 1, nu_log parameter 2 of the LRU scope."""
    rng = params_rng(seed)
    return (uniform(flax_param_key(rng, ssm_path(layer), 1), (n,)),
            uniform(flax_param_key(rng, ssm_path(layer), 2), (n,)))


def s4_log_step_uniforms(seed, layer, hidden):
    """u[h] of log_step for every feature column: parameter 7 of S4Layer (models/s4.py:105-129) under
    nn.vmap(split_rngs={"params": True}) -- the base key is split per column, the path suffix is kept."""
    rng = params_rng(seed)
    cols = split(rng, hidden)
    return np.array([uniform(flax_param_key(cols[h], ssm_path(layer), 7), (1,))[0] for h in range(hidden)],
                    dtype=np.float32)


def s5_log_step_uniforms(seed, layer, p, bidirectional=False, c_init="lecun_normal"):
    """u[i] of S5's log_step (models/s5.py:218-220): parameter 6 (7 with two C matrices) of the scope, drawn by
    init_log_steps (models/common.py:263-284): one split per state, uniform on the sub-key."""
    rng = params_rng(seed)
    counter = 7 if (bidirectional and c_init != "complex_normal") else 6
    key = flax_param_key(rng, ssm_path(layer), counter)
    out = np.empty(p, dtype=np.float32)
    for i in range(p):
        key, skey = split(key, 2)
        out[i] = uniform(skey, (1,))[0]
    return out
