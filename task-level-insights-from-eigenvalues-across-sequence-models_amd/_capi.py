"""ctypes binding of libeigspec_hip.so -- the C ABI declared in include/eigspec.h.

The product path has no CPU fallback: if the shared library is missing or fails to load, every op raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# EGS_LIB_PATH selects another build of the same library (A/B tuning runs); the default is the in-tree build
LIB_PATH = os.environ.get("EGS_LIB_PATH") or os.path.join(_HERE, "libeigspec_hip.so")

# constants mirrored from include/eigspec.h
EGS_ERR_RUNTIME = -1000
EGS_F32, EGS_F64, EGS_C64, EGS_C128 = 0, 1, 2, 3
EGS_DIAG_LRU, EGS_DIAG_S5_ZOH = 0, 1
EGS_NORM = {"exp": 0, "elu": 1, "softplus": 2, "sigmoid": 3}
EGS_HIST_VALUE, EGS_HIST_RADIUS, EGS_HIST_PHASE_DEG, EGS_HIST_ZERO = 0, 1, 2, 3

_c = ctypes
_vp, _i, _i64, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t


class DeviceProps(_c.Structure):
    _fields_ = [("compute_units", _i), ("wavefront_size", _i), ("lds_bytes_per_block", _i), ("clock_khz", _i),
                ("hbm_bytes", _i64), ("arch", _c.c_char * 32)]


# name -> (restype, argtypes); every symbol include/eigspec.h declares
SIGNATURES = {
    "egs_version": (_i, []),
    "egs_last_error_string": (_c.c_char_p, []),
    "egs_device_props_query": (_i, [_i, _c.POINTER(DeviceProps)]),
    "egs_diag_ssm_lambda": (_i, [_i, _vp, _vp, _vp, _i64, _vp, _vp]),
    "egs_s4_dplr_build": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "egs_s4_dplr_build_pq": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "egs_dplr_rank1_eigvals": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "egs_s4_cayley_map": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "egs_eig_workspace_bytes": (_sz, [_i, _i64, _i]),
    "egs_eig_general_batched": (_i, [_vp, _i, _i64, _i, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "egs_hessenberg_general_batched": (_i, [_vp, _i, _i64, _i, _i, _i64, _vp, _vp, _vp, _vp]),
    "egs_hseqr_general_batched": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "egs_hessenberg_batched": (_i, [_vp, _i64, _i, _i, _i64, _vp, _vp]),
    "egs_hseqr_batched": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "egs_mamba2_lambda": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "egs_mamba2_lti_lambda": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "egs_attn_workspace_bytes": (_sz, [_i64, _i64, _i, _i]),
    "egs_linattn_eta": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "egs_softmax_eta": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "egs_normattn_eta": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "egs_threshold_hist": (_i, [_vp, _i, _i, _i64, _i64, _i64, _c.POINTER(_c.c_double), _i, _vp, _vp]),
    "egs_softmax_eta_qk_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "egs_softmax_eta_qk": (_i, [_vp, _i64, _i64, _i, _i, _vp, _vp, _sz, _vp]),
    "egs_linattn_eta_qk_workspace_bytes": (_sz, [_i64, _i64, _i, _i]),
    "egs_linattn_eta_qk": (_i, [_vp, _i64, _i64, _i, _i, _vp, _vp, _sz, _vp]),
    "egs_causal_conv1d_silu": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i, _i, _vp]),
    "egs_layer_norm": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _c.c_float, _vp]),
    "egs_softmax_attention_fwd": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i,
                                       _c.c_float, _vp]),
    "egs_linrec_workspace_bytes": (_sz, [_i64, _i64, _i, _i, _i, _i, _c.c_uint]),
    "egs_linear_recurrence_fwd": (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                       _i, _i, _i, _i, _i, _c.c_float, _i, _vp, _sz, _vp]),
}

_lib = None


class EigspecError(RuntimeError):
    pass


def build(verbose=False):
    """``make`` the library in-tree (hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", _HERE, "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise EigspecError("building libeigspec_hip.so failed (see output above)")
    global _lib
    _lib = None
    return LIB_PATH


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EigspecError(
            f"{LIB_PATH} not found: build it with `make -C {_HERE}` or __graft_entry__.build(). "
            "There is no CPU fallback for the eigenvalue-analysis path.")
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(status, what):
    """Turn a C status into the Python error the reference's call site would have raised."""
    if status == 0:
        return
    msg = lib().egs_last_error_string().decode("utf-8", "replace")
    if status == EGS_ERR_RUNTIME:
        raise EigspecError(f"{what}: {msg}")
    raise ValueError(f"{what}: invalid argument {-status}: {msg}")
