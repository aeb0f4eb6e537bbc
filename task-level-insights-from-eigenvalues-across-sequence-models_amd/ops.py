"""Device ops of the eigenvalue-analysis path: torch tensors in, torch tensors out, compute in HIP.

torch is plumbing only (device memory, current stream); every function calls one entry point of the C ABI
(include/eigspec.h) on torch's current stream and returns without synchronising.  Nothing here falls back to
torch / NumPy arithmetic: without a GPU or without libeigspec_hip.so the ops raise.
"""
import ctypes

import torch

from . import _capi
from ._capi import check, lib

_workspaces = {}
_workspace_caps = {}          # key -> {"asked": full request, "fitted": size that fitted, "calls": requests since then}
_CAP_RETRY_CALLS = 16         # every that many requests the full size is tried again


def _require_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _capi.EigspecError(f"{name} must be a torch tensor on the GPU (got {type(t).__name__}"
                                 f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}); "
                                 "the eigenvalue-analysis path has no CPU fallback")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _workspace(device, nbytes, tag="scratch", floor=None):
    """Grow-only scratch buffer per (device, stream, purpose) -- the C ABI never allocates.  Keyed by the CURRENT
    stream: two streams (or two host threads driving their own streams) never share scratch, and a buffer is only ever
    released to torch's caching allocator from the stream that used it, which keeps the release stream-ordered.

    `floor`: the entry point also works with less than `nbytes` (the real n <= 64 eigenvalue solver then runs its batch
    in several passes, include/eigspec.h): when the device cannot give `nbytes`, the request is halved down to `floor`
    before the out-of-memory error is passed on, and the size that fitted is remembered for this key."""
    nbytes = max(int(nbytes), 256)
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    full = nbytes
    cap = _workspace_caps.get(key)
    if floor is not None and cap is not None:
        # a cap is the size that FITTED after an out-of-memory halving for a request of cap["asked"] bytes.  It holds for
        # requests of that size for _CAP_RETRY_CALLS calls, then the full request is tried again: one transient shortage
        # must not pin the stream to the slower several-pass form for the rest of the process
        cap["calls"] += 1
        if cap["asked"] == full and cap["calls"] % _CAP_RETRY_CALLS != 0:
            nbytes = max(min(nbytes, cap["fitted"]), int(floor))
    buf = _workspaces.get(key)
    if buf is not None and buf.numel() >= nbytes:
        return buf
    _workspaces.pop(key, None)
    del buf
    want = nbytes
    while True:
        try:
            _workspaces[key] = buf = torch.empty(want, dtype=torch.uint8, device=device)
            if want < full:                         # recorded only once the smaller allocation has succeeded
                if key not in _workspace_caps:
                    import warnings
                    warnings.warn("eigspec: %s workspace of %d bytes does not fit on %s, using %d (the n <= 64 solver runs "
                                  "its batch in several passes: same spectra, lower throughput)" % (tag, full, device, want),
                                  RuntimeWarning, stacklevel=3)
                _workspace_caps[key] = {"asked": full, "fitted": want, "calls": 0}
            else:
                _workspace_caps.pop(key, None)
            return buf
        except torch.cuda.OutOfMemoryError:
            if floor is None or want <= int(floor):
                raise
            torch.cuda.empty_cache()
            want = max(int(floor), want // 2)


def release_workspaces():
    _workspaces.clear()
    _workspace_caps.clear()


def device_props(device=0):
    p = _capi.DeviceProps()
    check(lib().egs_device_props_query(int(device), ctypes.byref(p)), "egs_device_props_query")
    return {"compute_units": p.compute_units, "wavefront_size": p.wavefront_size,
            "lds_bytes_per_block": p.lds_bytes_per_block, "clock_khz": p.clock_khz, "hbm_bytes": p.hbm_bytes,
            "arch": p.arch.decode()}


# ---- K5 -------------------------------------------------------------------------------------------------
_EIG_DTYPES = {torch.float32: _capi.EGS_F32, torch.float64: _capi.EGS_F64, torch.complex64: _capi.EGS_C64,
               torch.complex128: _capi.EGS_C128}


def eig_general_batched(a, workspace=None):
    """Eigenvalues of a batch of dense matrices a[..., n, n] (row-major; single-precision input is widened on the
    device and solved in double precision).

    Returns (w complex128 [..., n], info int32 [...]).  Replaces np.linalg.eigvals (analysis/eval_eig.py:296)."""
    _require_cuda(a, "a")
    if a.dtype not in _EIG_DTYPES:
        raise TypeError(f"eig_general_batched takes float32/float64/complex64/complex128 matrices: got {a.dtype}")
    if a.dim() < 2 or a.shape[-1] != a.shape[-2]:
        raise ValueError(f"expected [..., n, n], got {tuple(a.shape)}")
    n = a.shape[-1]
    batch_shape = a.shape[:-2]
    if n == 0 or a.numel() == 0:
        return (torch.empty(tuple(batch_shape) + (n,), dtype=torch.complex128, device=a.device),
                torch.zeros(tuple(batch_shape), dtype=torch.int32, device=a.device))
    a3 = a.reshape(-1, n, n)
    if not a3.is_contiguous():
        a3 = a3.contiguous()
    nmat = a3.shape[0]
    dtype = _EIG_DTYPES[a.dtype]
    w = torch.empty((nmat, n), dtype=torch.complex128, device=a.device)
    info = torch.empty((nmat,), dtype=torch.int32, device=a.device)
    with torch.cuda.device(a.device):
        # query and launch under the SAME current device: the n = 64 sizes depend on its compute-unit count
        if workspace is not None:
            ws = workspace
        else:
            need = lib().egs_eig_workspace_bytes(dtype, nmat, n)
            # real float64 input of order <= 64 is solved in passes over whatever the workspace holds (eigspec.h): under
            # memory pressure take less instead of failing (n = 64 asks for 8.9 KB per matrix of a pass)
            floor = (lib().egs_eig_workspace_bytes(dtype, min(nmat, 4096), n)
                     if a.dtype == torch.float64 and n <= 64 else None)
            ws = _workspace(a.device, need, floor=floor)
        check(lib().egs_eig_general_batched(_ptr(a3), dtype, nmat, n, n, n * n, _ptr(w), _ptr(info), _ptr(ws),
                                            ws.numel() * ws.element_size(), _stream()), "egs_eig_general_batched")
    return w.reshape(*batch_shape, n), info.reshape(batch_shape)


def hessenberg_general_batched(a):
    """Stage 1 of the level-3 path (64 < n <= 1024): a [nmat, n, n] float64 / complex128 -> (h complex128 [nmat, n, n],
    w, info).  Read row-major, h is LOWER Hessenberg with the eigenvalues of a (see include/eigspec.h); w / info are
    the output buffers of stage 2 (info = n and a NaN spectrum already set for non-finite input)."""
    _require_cuda(a, "a")
    if a.dtype not in (torch.float64, torch.complex128):
        raise TypeError(f"hessenberg_general_batched takes float64 / complex128: got {a.dtype}")
    a = a.contiguous()
    nmat, n, _ = a.shape
    h = torch.empty((nmat, n, n), dtype=torch.complex128, device=a.device)
    w = torch.empty((nmat, n), dtype=torch.complex128, device=a.device)
    info = torch.empty((nmat,), dtype=torch.int32, device=a.device)
    with torch.cuda.device(a.device):
        check(lib().egs_hessenberg_general_batched(_ptr(a), _EIG_DTYPES[a.dtype], nmat, n, n, n * n, _ptr(h), _ptr(w),
                                                   _ptr(info), _stream()), "egs_hessenberg_general_batched")
    return h, w, info


def hseqr_general_batched(h, w, info, profile=False):
    """Stage 2 of the level-3 path: multishift QR on the output of hessenberg_general_batched (h is destroyed).
    profile=True also returns the per-matrix phase cycle counts [nmat, 12] (see include/eigspec.h)."""
    _require_cuda(h, "h")
    nmat, n, _ = h.shape
    prof = torch.zeros((nmat, 12), dtype=torch.float64, device=h.device) if profile else None
    with torch.cuda.device(h.device):
        check(lib().egs_hseqr_general_batched(_ptr(h), nmat, n, _ptr(w), _ptr(info), _ptr(prof), _stream()),
              "egs_hseqr_general_batched")
    return (w, info, prof) if profile else (w, info)


def hessenberg_batched(a):
    """Stage 1 of the real LDS solver (n <= 64): a [nmat, n, n] float64 -> Hessenberg h [nmat, n, n]."""
    _require_cuda(a, "a")
    a = a.contiguous()
    nmat, n, _ = a.shape
    h = torch.empty_like(a)
    with torch.cuda.device(a.device):
        check(lib().egs_hessenberg_batched(_ptr(a), nmat, n, n, n * n, _ptr(h), _stream()), "egs_hessenberg_batched")
    return h


def hseqr_batched(h):
    """Stage 2 of the real LDS solver: Hessenberg h [nmat, n, n] float64 -> (w complex128 [nmat, n], info)."""
    _require_cuda(h, "h")
    h = h.contiguous()
    nmat, n, _ = h.shape
    w = torch.empty((nmat, n), dtype=torch.complex128, device=h.device)
    info = torch.empty((nmat,), dtype=torch.int32, device=h.device)
    with torch.cuda.device(h.device):
        queue = _workspace(h.device, 256, tag="hqr-queue")      # its own allocation: never aliases an op's scratch
        check(lib().egs_hseqr_batched(_ptr(h), nmat, n, _ptr(w), _ptr(info), _ptr(queue), _stream()), "egs_hseqr_batched")
    return w, info


# ---- K1 / S4 build ----------------------------------------------------------------------------------------
def diag_ssm_lambda(mode, p0, p1, p2=None):
    """LRU (mode 'lru': nu_log, theta_log) / S5 (mode 's5': Lambda_re, Lambda_im, log_step) closed forms."""
    _require_cuda(p0, "p0")
    m = {"lru": _capi.EGS_DIAG_LRU, "s5": _capi.EGS_DIAG_S5_ZOH}[mode]
    p0 = p0.to(torch.float64).contiguous().reshape(-1)
    p1 = p1.to(torch.float64).contiguous().reshape(-1)
    p2 = p2.to(torch.float64).contiguous().reshape(-1) if p2 is not None else None
    n = p0.numel()
    out = torch.empty((n,), dtype=torch.complex128, device=p0.device)
    with torch.cuda.device(p0.device):
        check(lib().egs_diag_ssm_lambda(m, _ptr(p0), _ptr(p1), _ptr(p2), n, _ptr(out), _stream()),
              "egs_diag_ssm_lambda")
    return out


def s4_dplr_build(lambda_re, lambda_im, p, log_step, q=None, clip=True):
    """Ab [nmat, n, n] complex128 of the bilinear-discretised DPLR SSM, for nmat feature columns.

    lambda_re, lambda_im: [nmat, n] real; p: [nmat, n] complex (or real); log_step: [nmat].
    q: None = the reference's call, Q = P (analysis/eval_eig.py:294), else [nmat, n] -- A = diag(Lambda) - P Q^H;
    clip: Re(Lambda) <= -1e-4 as get_eigvals_ssm clips before the call (:288); False = Lambda as given."""
    _require_cuda(lambda_re, "lambda_re")
    lr = lambda_re.to(torch.float64).contiguous()
    li = lambda_im.to(torch.float64).contiguous()
    pc = p.to(torch.complex128)
    pr = pc.real.contiguous()
    pi = pc.imag.contiguous()
    ls = log_step.to(torch.float64).contiguous().reshape(-1)
    nmat, n = lr.shape
    ab = torch.empty((nmat, n, n), dtype=torch.complex128, device=lr.device)
    with torch.cuda.device(lr.device):
        if q is None and clip:
            check(lib().egs_s4_dplr_build(_ptr(lr), _ptr(li), _ptr(pr), _ptr(pi), _ptr(ls), nmat, n, _ptr(ab), _stream()),
                  "egs_s4_dplr_build")
        else:
            qc = pc if q is None else q.to(torch.complex128)
            if tuple(qc.shape) != (nmat, n):
                raise ValueError(f"q must be [nmat, n] = {(nmat, n)}, got {tuple(qc.shape)}")
            qr = qc.real.contiguous()
            qi = qc.imag.contiguous()
            check(lib().egs_s4_dplr_build_pq(_ptr(lr), _ptr(li), _ptr(pr), _ptr(pi), _ptr(qr), _ptr(qi), _ptr(ls), nmat, n,
                                             1 if clip else 0, _ptr(ab), _stream()), "egs_s4_dplr_build_pq")
    return ab


def s4_dplr_eigs(lambda_re, lambda_im, p, log_step, q=None, clip=True, dedupe=True):
    """Eigenvalues of the bilinear-discretised DPLR state matrix WITHOUT forming it (analysis key `s4_solver: dplr`):
    eig(Ab) = (g + a)/(g - a), a in eig(diag(Lambda) - P Q^H) from the secular equation (egs_dplr_rank1_eigvals), g = 2/step.

    lambda_re, lambda_im: [ncol, n]; p (q): [ncol, n] complex; log_step: [ncol].  Returns (w complex128 [ncol, n],
    info int32 [ncol]).  dedupe: columns whose (Lambda, P, Q) agree bit for bit share one solve (`a` does not depend on the
    step) -- every column of a HiPPO-initialised layer does.  Replaces discrete_DPLR + np.linalg.eigvals
    (analysis/eval_eig.py:294-296); agrees with the dense path to rounding on well-conditioned spectra and by backward
    error on ill-conditioned ones."""
    _require_cuda(lambda_re, "lambda_re")
    lr = lambda_re.to(torch.float64).contiguous()
    li = lambda_im.to(torch.float64).contiguous()
    pc = p.to(torch.complex128)
    qc = pc if q is None else q.to(torch.complex128)
    ls = log_step.to(torch.float64).contiguous().reshape(-1)
    ncol, n = lr.shape
    if tuple(pc.shape) != (ncol, n) or tuple(qc.shape) != (ncol, n) or ls.numel() != ncol:
        raise ValueError(f"expected [ncol, n] = {(ncol, n)} parameters and [ncol] log_step")
    w = torch.empty((ncol, n), dtype=torch.complex128, device=lr.device)
    if ncol == 0 or n == 0:
        return w, torch.zeros((ncol,), dtype=torch.int32, device=lr.device)
    planes = [lr, li, pc.real.contiguous(), pc.imag.contiguous(), qc.real.contiguous(), qc.imag.contiguous()]
    set_index = None
    if dedupe and ncol > 1:
        # index plumbing, not arithmetic: group the columns by the BITS of their parameters.  The case that matters -- a
        # HiPPO-initialised layer, every column equal to the first -- is one comparison pass (a lexicographic
        # torch.unique over 768 x 1536 words measured 25 ms, ten times the solve it was meant to save); anything else is
        # grouped by a 64-bit mix of the row's words and the groups are then verified word for word.
        key = torch.cat([t.view(torch.int64) for t in planes], dim=1)
        same_as_first = (key == key[0:1]).all(dim=1)
        if bool(same_as_first.all().item()):
            planes = [t[0:1].contiguous() for t in planes]
            set_index = torch.zeros((ncol,), dtype=torch.int32, device=lr.device)
        elif dedupe == "all" or dedupe is True:
            mult = (torch.arange(1, key.shape[1] + 1, device=lr.device, dtype=torch.int64) * 0x1E3779B97F4A7C15) | 1   # odd, wraps
            h = (key * mult).sum(dim=1)                               # wrap-around int64 arithmetic: a hash, nothing more
            _, inverse = torch.unique(h, return_inverse=True)
            nuniq = int(inverse.max().item()) + 1
            if nuniq < ncol:
                first = torch.full((nuniq,), ncol, dtype=torch.int64, device=lr.device)
                first.scatter_reduce_(0, inverse, torch.arange(ncol, device=lr.device), reduce="amin")
                if bool((key == key[first[inverse]]).all().item()):   # equal hash AND equal words: a real group
                    planes = [t[first].contiguous() for t in planes]
                    set_index = inverse.to(torch.int32).contiguous()
    nset = planes[0].shape[0]
    a = torch.empty((nset, n), dtype=torch.complex128, device=lr.device)
    info_set = torch.empty((nset,), dtype=torch.int32, device=lr.device)
    with torch.cuda.device(lr.device):
        check(lib().egs_dplr_rank1_eigvals(*[_ptr(t) for t in planes], nset, n, 1 if clip else 0, _ptr(a), _ptr(info_set),
                                           _stream()), "egs_dplr_rank1_eigvals")
        check(lib().egs_s4_cayley_map(_ptr(a), _ptr(set_index), _ptr(ls), ncol, n, _ptr(w), _stream()), "egs_s4_cayley_map")
    info = info_set if set_index is None else info_set[set_index.long()]
    return w, info


# ---- K2 -------------------------------------------------------------------------------------------------
def mamba2_lambda(x, w_dt, b_dt, dt_bias, a_log):
    """x [B, L, D] fp32 -> lambda [B, L, nheads] fp32 (analysis/eval_eig.py:176-190)."""
    _require_cuda(x, "x")
    x = x.to(torch.float32).contiguous()
    bsz, seqlen, d = x.shape
    w = w_dt.to(torch.float32).contiguous()
    nh = w.shape[0]
    b = b_dt.to(torch.float32).contiguous() if b_dt is not None else None
    dtb = dt_bias.to(torch.float32).contiguous()
    al = a_log.to(torch.float32).contiguous()
    out = torch.empty((bsz, seqlen, nh), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().egs_mamba2_lambda(_ptr(x), bsz * seqlen, d, d, _ptr(w), _ptr(b), _ptr(dtb), _ptr(al), nh,
                                      _ptr(out), _stream()), "egs_mamba2_lambda")
    return out


def mamba2_lti_lambda(a_param, beta, batch, seqlen):
    _require_cuda(a_param, "a_param")
    a = a_param.to(torch.float32).contiguous()
    bt = beta.to(torch.float32).contiguous().to(a.device)
    nh = a.numel()
    out = torch.empty((batch, seqlen, nh), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        check(lib().egs_mamba2_lti_lambda(_ptr(a), _ptr(bt), nh, batch * seqlen, _ptr(out), _stream()),
              "egs_mamba2_lti_lambda")
    return out


# ---- K3 / K4 / norm -----------------------------------------------------------------------------------------
def _attn(fn_name, x, w_qk, b_qk, d_qk, num_heads):
    _require_cuda(x, "x")
    x = x.to(torch.float32).contiguous()
    bsz, seqlen, d = x.shape
    w = w_qk.to(torch.float32).contiguous()
    if w.shape != (2 * d_qk, d):
        raise ValueError(f"w_qk must be [2*d_qk, D] = {(2 * d_qk, d)}, got {tuple(w.shape)}")
    b = b_qk.to(torch.float32).contiguous() if b_qk is not None else None
    eta = torch.empty((bsz, max(seqlen - 1, 0), num_heads), dtype=torch.float64, device=x.device)
    need = lib().egs_attn_workspace_bytes(bsz, seqlen, d_qk, num_heads)
    ws = _workspace(x.device, need)
    with torch.cuda.device(x.device):
        check(getattr(lib(), fn_name)(_ptr(x), bsz, seqlen, d, _ptr(w), _ptr(b), d_qk, num_heads, _ptr(eta), _ptr(ws),
                                      ws.numel(), _stream()), fn_name)
    return eta


def linattn_eta(x, w_qk, b_qk, d_qk, num_heads):
    """eta [B, L-1, H] float64 of linear attention (analysis/eval_eig.py:97-135), O(L) streaming form."""
    return _attn("egs_linattn_eta", x, w_qk, b_qk, d_qk, num_heads)


def softmax_eta(x, w_qk, b_qk, d_qk, num_heads):
    """eta [B, L-1, H] float64 of softmax attention (analysis/eval_eig.py:43-95), no (B,L,L,H) tensor."""
    return _attn("egs_softmax_eta", x, w_qk, b_qk, d_qk, num_heads)


def normattn_eta(x, w_n, b_n, offset, norm_fn, num_heads):
    """eta [B, L-1, H] float64 of normalised attention (analysis/eval_eig.py:137-174)."""
    _require_cuda(x, "x")
    if norm_fn not in _capi.EGS_NORM:
        raise RuntimeError("normalization function {0} not implemented!".format(norm_fn))   # eval_eig.py:151
    x = x.to(torch.float32).contiguous()
    bsz, seqlen, d = x.shape
    w = w_n.to(torch.float32).contiguous()
    b = b_n.to(torch.float32).contiguous() if b_n is not None else None
    off = offset.to(torch.float32).contiguous().to(x.device) if offset is not None else None
    eta = torch.empty((bsz, max(seqlen - 1, 0), num_heads), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().egs_normattn_eta(_ptr(x), bsz, seqlen, d, _ptr(w), _ptr(b), _ptr(off), _capi.EGS_NORM[norm_fn],
                                     num_heads, _ptr(eta), _stream()), "egs_normattn_eta")
    return eta


# ---- a9 / a10 ------------------------------------------------------------------------------------------------
_TRANSFORMS = {"value": _capi.EGS_HIST_VALUE, "radius": _capi.EGS_HIST_RADIUS, "phase": _capi.EGS_HIST_PHASE_DEG,
               "zero": _capi.EGS_HIST_ZERO}
_DTYPES = {torch.float32: _capi.EGS_F32, torch.float64: _capi.EGS_F64, torch.complex64: _capi.EGS_C64,
           torch.complex128: _capi.EGS_C128}


def threshold_hist(vals, edges, transform="value"):
    """Inclusive-edge percentages over axis 1 of vals [outer, T, *inner] -> float64 [len(edges)+1, outer, *inner]
    (analysis/eval_eig.py:335-391 with the radius / phase transforms of :605-615 / :726-735 fused in)."""
    _require_cuda(vals, "vals")
    if vals.dtype not in _DTYPES:
        raise TypeError(f"unsupported dtype {vals.dtype}")
    vals = vals.contiguous()
    outer, t_len = vals.shape[0], vals.shape[1]
    inner_shape = tuple(vals.shape[2:])
    inner = 1
    for s in inner_shape:
        inner *= s
    edges = [float(e) for e in edges]
    c_edges = (ctypes.c_double * len(edges))(*edges)
    pct = torch.empty((len(edges) + 1, outer) + inner_shape, dtype=torch.float64, device=vals.device)
    with torch.cuda.device(vals.device):
        check(lib().egs_threshold_hist(_ptr(vals), _DTYPES[vals.dtype], _TRANSFORMS[transform], outer, t_len, inner,
                                       c_edges, len(edges), _ptr(pct), _stream()), "egs_threshold_hist")
    return pct


# ---- block forward ---------------------------------------------------------------
def linear_recurrence_fwd(q, k, v, decay=None, vscale=None, w=None, skip=None, phi=False, k_scale=1.0,
                          normalise=False, cols_per_decay=None):
    """y_t = w_t (q_t^T S_t) + skip * v_t with S_t = decay_t S_{t-1} + k_t (vscale_t v_t)^T, fp32, per batch row.

    q, k: [B, L, G, d] (views with equal strides and a contiguous last dim -- e.g. slices of one projection output --
    are passed through as they are, anything else is made contiguous first); v: [B, L, C], C = G * cols_per_group; decay / vscale: [B, L, C // cols_per_decay];
    w: [B, L, G]; skip: [C].  ``normalise`` computes w_t = 1 / (q_t . sum_{s<=t} k_s) on the device instead.
    Replaces the mixers of models/attention.py:58-83, models/norm_attention.py:61-89 and the
    mamba_chunk_scan_combined calls of models/mamba.py:138-150, :276-288."""
    for name, t in (("q", q), ("k", k), ("v", v)):
        _require_cuda(t, name)
        if t.dtype != torch.float32:
            raise TypeError(f"{name} must be float32, got {t.dtype}")
    if q.dim() != 4 or k.shape != q.shape:
        raise ValueError(f"q and k must both be [B, L, G, d], got {tuple(q.shape)} and {tuple(k.shape)}")
    if q.stride() != k.stride() or (q.shape[-1] > 1 and q.stride(-1) != 1):
        q, k = q.contiguous(), k.contiguous()                          # the ABI takes one stride triple for both
    batch, seqlen, groups, d_state = q.shape
    if v.dim() != 3 or v.shape[:2] != q.shape[:2]:
        raise ValueError(f"v must be [B, L, C] with the batch / sequence shape of q, got {tuple(v.shape)}")
    if v.shape[-1] > 1 and v.stride(-1) != 1:
        v = v.contiguous()
    channels = v.shape[-1]
    if channels % groups != 0:
        raise ValueError(f"channels ({channels}) must be a multiple of the q/k groups ({groups})")
    cpg = channels // groups
    cpd = cpg if cols_per_decay is None else int(cols_per_decay)
    nh = channels // cpd

    def _opt(t, shape, name):
        if t is None:
            return None
        _require_cuda(t, name)
        if tuple(t.shape) != shape:
            raise ValueError(f"{name} must have shape {shape}, got {tuple(t.shape)}")
        return t.to(torch.float32).contiguous()

    decay = _opt(decay, (batch, seqlen, nh), "decay")
    vscale = _opt(vscale, (batch, seqlen, nh), "vscale")
    w = _opt(w, (batch, seqlen, groups), "w")
    skip = _opt(skip, (channels,), "skip")
    y = torch.empty((batch, seqlen, channels), dtype=torch.float32, device=v.device)
    if batch == 0 or seqlen == 0:
        return y
    aligned = (q.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0 and all(st % 4 == 0 for st in q.stride()[:3]))
    if not aligned and d_state % 4 == 0 and d_state <= 128:
        # e.g. slices of a projection whose width is not a multiple of 4 (Wvqkn: D + 2 d_qk + H): one copy buys the
        # matrix-core kernels, which need 16-byte aligned q / k rows
        q, k = q.contiguous(), k.contiguous()
        aligned = True
    flags = ((1 if decay is not None else 0) | (2 if (w is not None and not normalise) else 0) |
             (4 if skip is not None else 0) | (8 if phi else 0) | (16 if normalise else 0) |
             (32 if float(k_scale) != 1.0 else 0) | (64 if aligned else 0))             # EGS_REC_* of eigspec.h
    need = lib().egs_linrec_workspace_bytes(batch, seqlen, channels, cpg, cpd, d_state, flags)
    ws = _workspace(v.device, need)
    with torch.cuda.device(v.device):
        check(lib().egs_linear_recurrence_fwd(
            _ptr(q), _ptr(k), q.stride(0), q.stride(1), q.stride(2), _ptr(v), v.stride(0), v.stride(1),
            _ptr(decay), _ptr(vscale), _ptr(w), _ptr(skip), _ptr(y), batch, seqlen, channels, d_state, cpg, cpd,
            1 if phi else 0, float(k_scale), 1 if normalise else 0, _ptr(ws), ws.numel() * ws.element_size(),
            _stream()), "egs_linear_recurrence_fwd")
    return y


def softmax_attention_fwd(q, k, v, scale=None):
    """Causal softmax attention, fp32: y[b,t,h,:] = sum_{s<=t} softmax_s(scale q_t.k_s) v_s.

    q, k: [B, L, H, hd] views with equal strides and contiguous last dim; v: [B, L, H * dv] with contiguous channels.
    ``scale`` defaults to 1/sqrt(hd).  Replaces SelfAttention.forward (models/attention.py:17-35)."""
    for name, t in (("q", q), ("k", k), ("v", v)):
        _require_cuda(t, name)
        if t.dtype != torch.float32:
            raise TypeError(f"{name} must be float32, got {t.dtype}")
    if q.dim() != 4 or k.shape != q.shape or q.stride() != k.stride() or (q.shape[-1] > 1 and q.stride(-1) != 1):
        raise ValueError("q and k must be [B, L, H, hd] views with identical strides and a contiguous last dim")
    batch, seqlen, heads, hd = q.shape
    if v.dim() != 3 or v.shape[:2] != q.shape[:2] or v.shape[-1] % heads != 0 or (v.shape[-1] > 1 and v.stride(-1) != 1):
        raise ValueError("v must be [B, L, H * dv] with contiguous channels")
    dv = v.shape[-1] // heads
    y = torch.empty((batch, seqlen, heads * dv), dtype=torch.float32, device=v.device)
    if batch == 0 or seqlen == 0:
        return y
    sc = float(scale) if scale is not None else 1.0 / float(hd) ** 0.5
    with torch.cuda.device(v.device):
        check(lib().egs_softmax_attention_fwd(_ptr(q), _ptr(k), q.stride(0), q.stride(1), q.stride(2), _ptr(v),
                                              v.stride(0), v.stride(1), _ptr(y), batch, seqlen, heads, hd, dv, sc,
                                              _stream()), "egs_softmax_attention_fwd")
    return y


# ---- notebook path ------------------------------------------------------------------
def softmax_eta_from_qk(q, k):
    """Softmax-attention eta from existing q, k: q [B, L, H, hd], k [B, L, H_kv, hd] with H % H_kv == 0 (grouped-query
    attention: query head h reads key head h // (H / H_kv), the transformers ``repeat_kv`` convention).  Any float
    dtype (computed in fp32 / fp64 like the eta path); returns float64 [B, L-1, H].
    Replaces get_eig_from_qkv_att_softmax of notebooks/lm_eigvals.ipynb (cell 13)."""
    _require_cuda(q, "q")
    _require_cuda(k, "k")
    if q.dim() != 4 or k.dim() != 4 or q.shape[:2] != k.shape[:2] or q.shape[3] != k.shape[3]:
        raise ValueError(f"expected q [B, L, H, hd] and k [B, L, H_kv, hd], got {tuple(q.shape)} and {tuple(k.shape)}")
    batch, seqlen, heads, hd = q.shape
    kv = k.shape[2]
    if kv == 0 or heads % kv != 0:
        raise ValueError(f"num_heads ({heads}) must be a multiple of num_kv_heads ({kv})")
    if kv != heads:
        k = k.repeat_interleave(heads // kv, dim=2)
    qk = torch.cat([q.reshape(batch, seqlen, heads * hd).float(), k.reshape(batch, seqlen, heads * hd).float()], dim=-1)
    qk = qk.contiguous()
    eta = torch.empty((batch, max(seqlen - 1, 0), heads), dtype=torch.float64, device=q.device)
    if batch == 0 or seqlen <= 1:
        return eta
    need = lib().egs_softmax_eta_qk_workspace_bytes(batch, seqlen, heads)
    ws = _workspace(q.device, need)
    with torch.cuda.device(q.device):
        check(lib().egs_softmax_eta_qk(_ptr(qk), batch, seqlen, heads * hd, heads, _ptr(eta), _ptr(ws),
                                       ws.numel() * ws.element_size(), _stream()), "egs_softmax_eta_qk")
    return eta


def linattn_eta_from_qk(q, k):
    """Linear-attention eta from existing raw projections q, k [B, L, H, hd] (phi = elu + 1 applied on the device):
    float64 [B, L-1, H].  The (q, k)-input form of get_eig_att_linear (analysis/eval_eig.py:113-130)."""
    _require_cuda(q, "q")
    _require_cuda(k, "k")
    if q.dim() != 4 or q.shape != k.shape:
        raise ValueError(f"expected q, k [B, L, H, hd] of equal shape, got {tuple(q.shape)} and {tuple(k.shape)}")
    batch, seqlen, heads, hd = q.shape
    qk = torch.cat([q.reshape(batch, seqlen, heads * hd).float(), k.reshape(batch, seqlen, heads * hd).float()],
                   dim=-1).contiguous()
    eta = torch.empty((batch, max(seqlen - 1, 0), heads), dtype=torch.float64, device=q.device)
    if batch == 0 or seqlen <= 1:
        return eta
    need = lib().egs_linattn_eta_qk_workspace_bytes(batch, seqlen, heads * hd, heads)
    ws = _workspace(q.device, need)
    with torch.cuda.device(q.device):
        check(lib().egs_linattn_eta_qk(_ptr(qk), batch, seqlen, heads * hd, heads, _ptr(eta), _ptr(ws),
                                       ws.numel() * ws.element_size(), _stream()), "egs_linattn_eta_qk")
    return eta


def causal_conv1d_silu(x, weight, bias=None):
    """silu(depthwise causal conv1d) on channel-last activations: x [B, L, C] (contiguous channels), weight [C, 1, K] or
    [C, K] (nn.Conv1d(groups=C).weight), bias [C] or None; K <= 8.  Returns [B, L, C] float32.
    Replaces ``act(conv1d(x.transpose(1, 2))[..., :L]).transpose(1, 2)`` (models/attention.py:152-155 and twins)."""
    _require_cuda(x, "x")
    if x.dim() != 3:
        raise ValueError(f"x must be [B, L, C], got {tuple(x.shape)}")
    x = x if x.dtype == torch.float32 else x.float()
    if x.shape[-1] > 1 and x.stride(-1) != 1:
        x = x.contiguous()
    batch, seqlen, channels = x.shape
    w = weight.detach().reshape(weight.shape[0], -1).to(torch.float32).contiguous()
    if w.shape[0] != channels:
        raise ValueError(f"weight has {w.shape[0]} channels, x has {channels}")
    kw = w.shape[1]
    if kw > 8:
        raise ValueError(f"kernel size {kw} exceeds the supported maximum of 8")
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    y = torch.empty((batch, seqlen, channels), dtype=torch.float32, device=x.device)
    if batch == 0 or seqlen == 0:
        return y
    with torch.cuda.device(x.device):
        check(lib().egs_causal_conv1d_silu(_ptr(x), x.stride(0), x.stride(1), _ptr(w), _ptr(b), _ptr(y), batch, seqlen,
                                           channels, kw, _stream()), "egs_causal_conv1d_silu")
    return y


def layer_norm(x, weight=None, bias=None, eps=1e-5):
    """LayerNorm over the last dimension (fp32, biased variance), x [..., d] with a contiguous last dim and d <= 1024.
    Replaces nn.LayerNorm.forward in the blocks (models/transformer.py:93-99, models/mamba.py:318-328)."""
    _require_cuda(x, "x")
    x = x if x.dtype == torch.float32 else x.float()
    d = x.shape[-1]
    if d > 1024:
        raise ValueError(f"layer_norm supports a normalised dimension up to 1024, got {d}")
    x2 = x.reshape(-1, d)
    if x2.shape[0] > 0 and x2.stride(-1) != 1:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    y = torch.empty((rows, d), dtype=torch.float32, device=x.device)
    if rows == 0:
        return y.reshape(x.shape)
    g = None if weight is None else weight.detach().to(torch.float32).contiguous()
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    with torch.cuda.device(x.device):
        check(lib().egs_layer_norm(_ptr(x2), x2.stride(0), _ptr(g), _ptr(b), _ptr(y), rows, d, float(eps), _stream()),
              "egs_layer_norm")
    return y.reshape(x.shape)
