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
aggregate file show up as ``PLACEHOLDER://<flattened.key.path>`` strings.  When the directory holds such a leaf as a
plain **zarr v2** array (``<dir>/<flattened.key.path>/.zarray`` + chunk files: what tensorstore's zarr driver writes
without OCDBT) it is decoded here from the published zarr v2 layout -- uncompressed chunks and the compressors Python's
standard library has (zlib / gzip, bz2, lzma); zstd / blosc chunks (orbax's usual choice) and OCDBT containers
(``manifest.ocdbt``) raise with the reason.  This reader is **parity unpinned** like the rest of the module: its test
builds the directory from the zarr v2 specification, nothing in this image can write one with tensorstore.
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


def _zarr_decompress(buf, compressor, where):
    if compressor is None:
        return buf
    cid = compressor.get("id")
    if cid in ("zlib", "gzip"):
        import zlib
        return zlib.decompress(buf, 15 + 32)                 # zlib or gzip header, detected
    if cid == "bz2":
        import bz2
        return bz2.decompress(buf)
    if cid == "lzma":
        import lzma
        return lzma.decompress(buf)
    raise NotImplementedError(f"{where}: chunks are compressed with {cid!r}; only uncompressed, zlib / gzip, bz2 and lzma "
                              "zarr chunks can be decoded here (no zstd / blosc codec in this environment)")


def read_zarr_v2(path):
    """One zarr v2 array directory (``.zarray`` + chunk files ``i.j.k`` / ``i/j/k``) as a NumPy array -- the layout
    tensorstore's zarr driver gives an orbax parameter stored outside the aggregate file
    (jax_helpers/train_helpers.py:395-409 saves through ``ocp.PyTreeCheckpointer``).  Written from the zarr v2
    specification: C or F order inside a chunk, edge chunks stored at full chunk size, missing chunks = ``fill_value``."""
    import itertools
    import json
    meta_file = os.path.join(path, ".zarray")
    with open(meta_file) as f:
        meta = json.load(f)
    if meta.get("zarr_format") != 2:
        raise NotImplementedError(f"{meta_file}: zarr_format {meta.get('zarr_format')!r}, only version 2 is decoded")
    if meta.get("filters"):
        raise NotImplementedError(f"{meta_file}: filters {meta['filters']!r} are not decoded")
    shape, chunks = tuple(int(v) for v in meta["shape"]), tuple(int(v) for v in meta["chunks"])
    dtype = np.dtype(meta["dtype"]) if isinstance(meta["dtype"], str) else np.dtype([tuple(t) for t in meta["dtype"]])
    order = meta.get("order", "C")
    sep = meta.get("dimension_separator", ".")
    fill = meta.get("fill_value")
    if fill in ("NaN", "Infinity", "-Infinity"):
        fill = {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}[fill]
    out = np.zeros(shape, dtype=dtype) if fill is None else np.full(shape, fill, dtype=dtype)
    if len(shape) == 0:
        fname = os.path.join(path, "0")
        if os.path.isfile(fname):
            with open(fname, "rb") as f:
                out[...] = np.frombuffer(_zarr_decompress(f.read(), meta.get("compressor"), fname), dtype=dtype)[0]
        return out
    grid = [range((n + c - 1) // c) for n, c in zip(shape, chunks)]
    for idx in itertools.product(*grid):
        fname = os.path.join(path, sep.join(str(i) for i in idx)) if sep == "." else os.path.join(path, *[str(i) for i in idx])
        if not os.path.isfile(fname):
            continue                                          # an unwritten chunk reads as fill_value
        with open(fname, "rb") as f:
            raw = _zarr_decompress(f.read(), meta.get("compressor"), fname)
        block = np.frombuffer(raw, dtype=dtype)
        if block.size != int(np.prod(chunks)):
            raise ValueError(f"{fname}: {block.size} elements, the chunk shape {chunks} needs {int(np.prod(chunks))}")
        block = block.reshape(chunks, order=order)
        sel = tuple(slice(i * c, min((i + 1) * c, n)) for i, c, n in zip(idx, chunks, shape))
        out[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
    return out.astype(dtype.newbyteorder("="), copy=False)


def _resolve_placeholders(tree, root, prefix=""):
    """Replace PLACEHOLDER:// leaves by the arrays tensorstore stored beside the aggregate file (zarr v2 only)."""
    if isinstance(tree, dict):
        return {k: _resolve_placeholders(v, root, f"{prefix}/{k}") for k, v in tree.items()}
    if isinstance(tree, str) and tree.startswith("PLACEHOLDER://"):
        name = tree[len("PLACEHOLDER://"):]
        leaf = os.path.join(root, name) if root else None
        if leaf and os.path.isfile(os.path.join(leaf, ".zarray")):
            return read_zarr_v2(leaf)
        why = "tensorstore / OCDBT parameter files are not decoded here"
        if root and (os.path.exists(os.path.join(root, "manifest.ocdbt")) or
                     any(n.startswith("ocdbt.process_") for n in os.listdir(root))):
            why = "the directory is an OCDBT container (manifest.ocdbt), which is not decoded here"
        elif leaf and os.path.isdir(leaf):
            why = f"{leaf} holds no .zarray (not a zarr v2 array)"
        raise NotImplementedError(f"{prefix} was stored outside the aggregate file ({tree}): {why}")
    return tree


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
    return _resolve_placeholders(tree, path if os.path.isdir(path) else os.path.dirname(os.path.abspath(fname)))


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
