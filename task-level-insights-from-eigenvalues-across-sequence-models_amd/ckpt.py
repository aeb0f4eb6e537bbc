"""Checkpoint decoding without JAX / orbax.

The reference writes its JAX models with ``ocp.PyTreeCheckpointer().save(dir, ckpt,
save_args=orbax_utils.save_args_from_target(ckpt))`` (jax_helpers/train_helpers.py:395-409; orbax-checkpoint 0.5.3 /
0.6.3, flax 0.8.1).  ``save_args_from_target`` marks every single-device array ``aggregate=True``, so the whole
PyTree -- parameters included -- lands in ONE file, ``<dir>/checkpoint``, in flax's msgpack serialisation
(flax.serialization: ext type 1 = ndarray as msgpack((shape, dtype.name, C-order bytes)), 2 = Python complex as
(real, imag), 3 = NumPy scalar; arrays above 2**30 bytes are split into a ``__msgpack_chunked_array__`` dict).  This
module decodes exactly that file with the ``msgpack`` package, which is what ``get_trained_layers_ssm``
(analysis/eval_eig.py:241-252) needs to reach ``model/params/encoder/layers_i/seq``.

Parity unpinned: orbax / flax are not installed in this image and the reference ships no checkpoint, so the decoder is
written against the published serialisation format and is exercised with fixtures written by ``write_aggregate``
below (same format, round trip) -- not with a file produced by orbax itself.  Parameters that orbax stored outside the
aggregate file (tensorstore / OCDBT, used for multi-device arrays) show up as ``PLACEHOLDER://`` strings and raise.
"""
import os

import numpy as np

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_CHUNK_KEY = "__msgpack_chunked_array__"


def _dtype(name):
    """NumPy dtype of a flax-serialised array.  bfloat16 is not a NumPy type: it needs ml_dtypes (what JAX itself uses)."""
    if name == "bfloat16":
        try:
            import ml_dtypes
        except ImportError as exc:
            raise NotImplementedError("this checkpoint stores bfloat16 arrays; decoding them needs the ml_dtypes "
                                      "package (not installed here)") from exc
        return np.dtype(ml_dtypes.bfloat16)
    return np.dtype(name)


def _ndarray_from_bytes(data):
    import msgpack
    shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
    return np.frombuffer(buf, dtype=_dtype(dtype_name)).reshape(shape).copy()


def _ext_hook(code, data):
    import msgpack
    if code == _EXT_NDARRAY:
        return _ndarray_from_bytes(data)
    if code == _EXT_NPSCALAR:
        return _ndarray_from_bytes(data)[()]
    if code == _EXT_COMPLEX:
        re, im = msgpack.unpackb(data, raw=False)
        return complex(re, im)
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get(_CHUNK_KEY):
            chunks = tree["chunks"]
            flat = np.concatenate([np.asarray(chunks[str(i)]).reshape(-1) for i in range(len(chunks))])
            shape = tree["shape"]
            if isinstance(shape, dict):          # flax stores tuples as {"0": n0, "1": n1, ...} (_tuple_to_dict)
                shape = [shape[str(i)] for i in range(len(shape))]
            return flat.reshape(tuple(int(v) for v in shape))
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def _check_placeholders(tree, prefix=""):
    if isinstance(tree, dict):
        for k, v in tree.items():
            _check_placeholders(v, f"{prefix}/{k}")
    elif isinstance(tree, str) and tree.startswith("PLACEHOLDER://"):
        raise NotImplementedError(f"{prefix} was stored outside the aggregate file ({tree}): tensorstore / OCDBT "
                                  "parameter files are not decoded here")


def read_aggregate(path):
    """The PyTree stored in an orbax checkpoint directory's aggregate file (or that file itself) as nested dicts of
    NumPy arrays / scalars."""
    import msgpack
    fname = os.path.join(path, "checkpoint") if os.path.isdir(path) else path
    if not os.path.isfile(fname):
        raise FileNotFoundError(f"{fname}: no aggregate 'checkpoint' file -- not an orbax PyTreeCheckpointer directory "
                                "written with aggregate=True (flax orbax_utils.save_args_from_target)")
    with open(fname, "rb") as f:
        tree = msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)
    tree = _unchunk(tree)
    _check_placeholders(tree)
    return tree


def _pack_default(x):
    import msgpack
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(x.shape), x.dtype.name, x.tobytes("C")),
                                                           use_bin_type=True))
    if isinstance(x, np.generic):
        a = np.asarray(x)
        return msgpack.ExtType(_EXT_NPSCALAR, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes("C")),
                                                            use_bin_type=True))
    if isinstance(x, complex):
        return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((x.real, x.imag)))
    raise TypeError(f"cannot serialise {type(x).__name__}")


def write_aggregate(path, tree):
    """Write ``tree`` (nested dicts of NumPy arrays / scalars) as ``<path>/checkpoint`` in the same msgpack layout."""
    import msgpack
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "checkpoint"), "wb") as f:
        f.write(msgpack.packb(tree, default=_pack_default, use_bin_type=True))
