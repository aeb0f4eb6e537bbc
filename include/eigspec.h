/* eigspec.h -- C ABI of the MI355X-native eigenvalue-analysis hot path (libeigspec_hip.so).
 *
 * Drop-in boundary for the reference's analysis/eval_eig.py (all file:line citations are relative to the
 * reference repository root).  The reference has no FFI of its own: its hot path bottoms out in torch
 * (nn.Linear / einsum on "cuda"), JAX and NumPy/LAPACK calls.  Each entry point below replaces one such
 * call site; the Python host code (the package's analysis/eval_eig.py mirror) binds them with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in the signatures (egs_stream_t is a
 *     hipStream_t passed as void*; NULL = the default stream).
 *   - every function returns an int status: 0 = ok, -k = argument k (1-based) is invalid (LAPACK style),
 *     EGS_ERR_RUNTIME = a HIP runtime call failed (see egs_last_error_string()).
 *   - the caller owns every buffer.  All data pointers are DEVICE pointers unless stated otherwise.  The
 *     library never allocates or frees user-visible memory; scratch space is passed in as `workspace`
 *     (size from the matching *_workspace_bytes query; host-only, no GPU needed).
 *   - calls are asynchronous on `stream` unless stated otherwise; outputs are valid after the stream
 *     is synchronised.  One host thread per stream.
 *   - complex numbers are interleaved (re, im) pairs of the stated real type.
 */
#ifndef EIGSPEC_H
#define EIGSPEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* egs_stream_t;

#define EGS_VERSION 100          /* 0.1.0 */
#define EGS_ERR_RUNTIME (-1000)

/* element types for matrices / value arrays */
#define EGS_F32 0
#define EGS_F64 1
#define EGS_C64 2
#define EGS_C128 3

/* ---- diagnostics ------------------------------------------------------------------------------- */
int egs_version(void);
/* Thread-local, NUL-terminated description of the last failure on the calling thread ("" if none). */
const char* egs_last_error_string(void);

typedef struct egs_device_props {
    int compute_units;
    int wavefront_size;
    int lds_bytes_per_block;      /* max dynamic LDS a block of ours may request */
    int clock_khz;
    int64_t hbm_bytes;
    char arch[32];                /* e.g. "gfx950" */
} egs_device_props;
/* Synchronous; fills *out (HOST pointer) for HIP device `device`. */
int egs_device_props_query(int device, egs_device_props* out);

/* ---- K1: diagonal SSM spectra (closed forms) ---------------------------------------------------
 * Replaces get_eigvals_ssm "lru" (analysis/eval_eig.py:318-329; models/lru.py:88) and "s5"
 * (analysis/eval_eig.py:303-316: always ZOH, Lambda_re never clipped -- quirk Q10).
 *   mode 0 (lru): lambda[i] = exp(-exp(p0[i]) + i*exp(p1[i]))           p0 = nu_log, p1 = theta_log, p2 unused
 *   mode 1 (s5) : lambda[i] = exp((p0[i] + i*p1[i]) * exp(p2[i]))        p0 = Lambda_re, p1 = Lambda_im, p2 = log_step
 * p*: double[n]; out: complex128[n].  (The host mirror down-casts to complex64 when asked to emulate JAX's default.) */
#define EGS_DIAG_LRU 0
#define EGS_DIAG_S5_ZOH 1
int egs_diag_ssm_lambda(int mode, const double* p0, const double* p1, const double* p2, int64_t n,
                        double* out_lambda, egs_stream_t stream);

/* ---- S4: bilinear-discretised DPLR state matrix --------------------------------------------------
 * Replaces discrete_DPLR's Ab (analysis/eval_eig.py:254-273, twin models/s4.py:16-40) with Q = P as
 * called at analysis/eval_eig.py:294, for `nmat` independent feature columns at once.  The discarded
 * Bb/Cb (matrix_power, inv -- quirk Q18) are not computed.  Lambda_re is clipped to <= -1e-4 inside
 * (analysis/eval_eig.py:288, quirk Q11); step = exp(log_step).
 *   lambda_re, lambda_im, p_re, p_im : double[nmat, n]      log_step : double[nmat]
 *   ab : complex128[nmat, n, n] row-major, Ab = A1 * A0. */
int egs_s4_dplr_build(const double* lambda_re, const double* lambda_im, const double* p_re,
                      const double* p_im, const double* log_step, int64_t nmat, int n,
                      double* ab, egs_stream_t stream);
/* The general form of discrete_DPLR (analysis/eval_eig.py:254-273): A = diag(Lambda) - P Q^H with separate P and Q
 * (q_re, q_im : double[nmat, n]).  clip_lambda = 1 clips Re(Lambda) to <= -1e-4 as get_eigvals_ssm does before the call
 * (:288); 0 takes Lambda as given, which is what discrete_DPLR itself does. */
int egs_s4_dplr_build_pq(const double* lambda_re, const double* lambda_im, const double* p_re,
                         const double* p_im, const double* q_re, const double* q_im,
                         const double* log_step, int64_t nmat, int n, int clip_lambda,
                         double* ab, egs_stream_t stream);

/* ---- S4 fast path: eigenvalues from the DPLR structure (optional; the dense solver stays the default) ----------
 * eig(Ab) without forming Ab: Ab = (gI - A)^-1 (gI + A) with A = diag(Lambda) - P Q^H and g = 2/step
 * (analysis/eval_eig.py:260-273), so eig(Ab) = (g + a)/(g - a), a in eig(A) -- and A, diagonal plus rank one, has its
 * eigenvalues at the roots of 1 - sum_k conj(Q_k) P_k / (Lambda_k - a) (Ehrlich-Aberth iteration, O(n^2) per sweep).
 * Replaces the pair discrete_DPLR + np.linalg.eigvals (analysis/eval_eig.py:294-296) when the analysis YAML sets
 * `s4_solver: dplr`.  Root by root backward stable (each value is an exact eigenvalue of A with P perturbed by O(n ulp)
 * elementwise); on ill-conditioned spectra (raw HiPPO, N >= 64) it agrees with the dense path by backward error only.
 *   egs_dplr_rank1_eigvals: lambda_*, p_*, q_* : double[nmat, n];  a : complex128[nmat, n] = eig(diag(Lambda) - P Q^H);
 *       info[m] = number of roots of set m that did not converge (0 = ok).  clip_lambda as in egs_s4_dplr_build_pq.
 *   egs_s4_cayley_map: w[c, :] = (g_c + a[s, :]) / (g_c - a[s, :]), g_c = 2/exp(log_step[c]), s = set_index[c] (or c when
 *       set_index is null): `a` does not depend on the step, so columns sharing (Lambda, P, Q) share ONE solve. */
int egs_dplr_rank1_eigvals(const double* lambda_re, const double* lambda_im, const double* p_re,
                           const double* p_im, const double* q_re, const double* q_im, int64_t nmat, int n,
                           int clip_lambda, double* a, int32_t* info, egs_stream_t stream);
int egs_s4_cayley_map(const double* a, const int32_t* set_index, const double* log_step, int64_t ncol, int n,
                      double* w, egs_stream_t stream);

/* ---- K5: batched dense non-symmetric eigenvalues -------------------------------------------------
 * Replaces np.linalg.eigvals (LAPACK ?geev, analysis/eval_eig.py:296), generalised from one matrix per
 * layer to a batch.  Householder reduction to Hessenberg form + shifted QR (Francis double-shift for real
 * input, single-shift for complex input), eigenvalues only.
 *   a      : [nmat] matrices, element (i,j) of matrix m at a[m*stride + i*lda + j] (elements of `dtype`;
 *            row-major).  Not modified.
 *   w      : complex128[nmat, n] eigenvalues, in the order the deflation produced them (sort to compare).
 *   info   : int32[nmat]; 0 = converged, k > 0 = QR failed to deflate with k eigenvalues outstanding
 *            (their slots hold NaN), as LAPACK's xHSEQR info.
 *   workspace : device scratch of egs_eig_workspace_bytes(dtype, nmat, n) bytes, 256-byte aligned.
 * Real n = 64 (BASELINE config C2) runs a fused kernel: per wavefront the Hessenberg reduction with the matrix in
 * registers, then the QR iteration in LDS, the whole batch behind one work queue; once a matrix's active block ends below
 * row 44 it is handed over (an 8.9 KB record per matrix in the workspace) to a second kernel that finishes it at twice the
 * occupancy -- same arithmetic, bit-identical spectra.  The workspace provides a 32 KB scratch slot per resident wavefront
 * plus those records; a smaller workspace than the query's value is accepted: records for >= min(nmat, 32768) matrices
 * give several passes, less gives the one-kernel form, which only needs the slots (it limits the number of resident
 * wavefronts, at least 4 slots + 256 bytes).  Spectra and info do not depend on the workspace size by a single bit; the
 * trade is memory against throughput (C2's 524288 matrices: 4.8 GB for one pass of the two-phase form; the Python
 * binding halves its request down to 4096 matrices' worth when the allocation fails).  Other n <= 64, real or complex, run LDS-resident in two
 * kernels (one wavefront per matrix, Hessenberg matrices of a chunk in the workspace); larger n (<= 1024; the reference's S4
 * state_dim 128 / 512 and config C5's 256) run the level-3 path: blocked Householder reduction with fp64 matrix-core panel
 * updates, then small-bulge multishift QR (one workgroup of 16 wavefronts per matrix, working copy in the workspace).
 * dtype EGS_F32 / EGS_C64 inputs are widened on load and solved in double precision (w stays complex128). */
size_t egs_eig_workspace_bytes(int dtype, int64_t nmat, int n);
int egs_eig_general_batched(const void* a, int dtype, int64_t nmat, int n, int lda, int64_t stride,
                            double* w, int32_t* info, void* workspace, size_t workspace_bytes,
                            egs_stream_t stream);

/* The two stages of the level-3 path (n > 64 .. 1024), exposed separately like LAPACK's xGEHRD / xHSEQR.
 *   egs_hessenberg_general_batched : a (dtype as above, row-major, not modified) -> h, complex128[nmat, n, n]: the
 *       upper-Hessenberg reduction of the TRANSPOSE of each (balanced) matrix, stored column-major -- i.e. read as a
 *       row-major [n, n] array, h is LOWER Hessenberg.  Each matrix is first scaled by a power of two, 2^-e, to a largest
 *       entry in [1, 2) (exact; what xGEEV does outside [sqrt(safmin), sqrt(safmax)], here always, so that the QR stage
 *       can use absolute negligibility thresholds): h has the eigenvalues of 2^-e a, and e travels as a double in the first
 *       slot of w (w[m][0].re), where egs_hseqr_general_batched picks it up and multiplies the eigenvalues back -- pass
 *       the same w to both calls.  info[m] = n flags NaN / Inf input.
 *   egs_hseqr_general_batched      : h as produced above (destroyed) -> w complex128[nmat, n], info.  Diagnostics:
 *       phase_cycles = NULL, or device double[nmat, 12] receiving per matrix the shader-clock cycles spent in the
 *       deflation scans, shift generator / small blocks, window load + store, bulge chase, deferred matrix-core
 *       updates and single-window sweeps, then the number of sweeps and of windows, then the chase split as wave 0
 *       sees it (row phase, barrier, column phase, barrier). */
int egs_hessenberg_general_batched(const void* a, int dtype, int64_t nmat, int n, int lda, int64_t stride, double* h,
                                   double* w, int32_t* info, egs_stream_t stream);
int egs_hseqr_general_batched(double* h, int64_t nmat, int n, double* w, int32_t* info, double* phase_cycles,
                              egs_stream_t stream);

/* The two stages of the real LDS-resident solver (dtype EGS_F64, n <= 64), exposed separately like LAPACK's
 * xGEHRD / xHSEQR so that callers (and bench.py) can time or reuse them individually.
 *   egs_hessenberg_batched : a -> h, double[nmat, n, n] row-major upper-Hessenberg (exact zeros below the
 *                            subdiagonal); orthogonally similar to a.  Any finite scale: the reduction works on a
 *                            power-of-two multiple of a (so that squares of entries neither under- nor overflow) and
 *                            scales the result back exactly.
 *   egs_hseqr_batched      : h -> w, info.  h is any upper-Hessenberg matrix (the output above or the caller's own),
 *                            at any finite scale: at n = 64 the kernel normalises it to max |entry| in [1, 2) first --
 *                            its deflation and tiny-column thresholds are absolute -- and multiplies the eigenvalues back.  `queue` is NULL or 8 bytes of device scratch: with
 *                            it the kernel runs as one resident set of wavefronts that take matrices from a work
 *                            queue (what egs_eig_general_batched does); without it, one matrix per wavefront. */
int egs_hessenberg_batched(const double* a, int64_t nmat, int n, int lda, int64_t stride, double* h,
                           egs_stream_t stream);
int egs_hseqr_batched(const double* h, int64_t nmat, int n, double* w, int32_t* info, void* queue,
                      egs_stream_t stream);

/* ---- K2: Mamba-2 per-token transition scalars -----------------------------------------------------
 * Replaces get_eig_mamba2 (analysis/eval_eig.py:176-190): only the last `nheads` rows of in_proj.weight
 * are used (models/mamba.py:63-64), fused with softplus(dt + dt_bias) (torch threshold 20) and
 * exp(dt * -exp(A_log)).  fp32 throughout (quirk Q7).
 *   x : float[ntok, ldx] (first D columns used)   w_dt : float[nheads, D]   b_dt : float[nheads] or NULL
 *   dt_bias, a_log : float[nheads]                out : float[ntok, nheads] */
int egs_mamba2_lambda(const float* x, int64_t ntok, int d_model, int64_t ldx, const float* w_dt,
                      const float* b_dt, const float* dt_bias, const float* a_log, int nheads,
                      float* out, egs_stream_t stream);
/* Replaces get_eig_mamba2_LTI (analysis/eval_eig.py:192-205): exp(beta * -softplus(A)) broadcast to
 * [ntok, nheads] (quirk Q19: the reference's dead in_proj GEMM is not run). */
int egs_mamba2_lti_lambda(const float* a_param, const float* beta, int nheads, int64_t ntok,
                          float* out, egs_stream_t stream);

/* ---- K3/K4: attention normaliser ratios -------------------------------------------------------------
 * Replace get_eig_att_linear (analysis/eval_eig.py:97-135) and get_eig_att_softmax (:43-95).  Inputs are
 * the block output x and the q/k rows of Wqkv (models/attention.py:120-122: rows [0, d_qk) = q,
 * [d_qk, 2 d_qk) = k, split "(two h d)"); the v rows are never touched.  Neither materialises the
 * (B, L, L, H) score tensor.
 *   x : float[B, L, D]   w_qk : float[2*d_qk, D]   b_qk : float[2*d_qk] or NULL   eta : double[B, L-1, H]
 *   workspace: egs_attn_workspace_bytes(B, L, d_qk, H) bytes (projected q,k in fp32, nu / row max scratch).
 * linear : nu_t = phi(q_t) . sum_{s<=t} phi(k_s), phi = elu + 1; nu == 0 -> 2e-23 (Q6); eta_t = nu_t / nu_{t+1}.
 * softmax: scores q_t.k_s (no 1/sqrt(d), Q3); m_t = max(max_{s<=t} score, 0 if t < L-1) (Q5);
 *          nu_t = sum_{s<=t} exp(score - m_t) + (L-1-t) (Q4); eta_t = nu_t/nu_{t+1} * exp(m_t - m_{t+1}).
 *          Accuracy: WITHIN THE REFERENCE'S OWN fp32 ERROR, NOT BIT-FAITHFUL to its rounding.  One pass with a running row
 *          maximum: the maximum is subtracted in fp32 as at analysis/eval_eig.py:62-72, exp runs in fp32 with an exact
 *          range reduction (1.5e-7 relative; the reference's np.exp is fp64 on fp32 differences), the row sums are fp64
 *          and are rescaled exactly when the maximum rises.  Against the fp64-exact value of the same formula the result
 *          stays within 2 x the reference's own deviation (tests/test_gpu_parity.py).  Scores that are NaN or +-inf are
 *          OUTSIDE the contract: the running maximum ignores NaN where torch.max propagates it (the reference's row
 *          becomes 0 / NaN), and an infinite score gives inf - inf.  Finite inputs of any magnitude are covered. */
size_t egs_attn_workspace_bytes(int64_t batch, int64_t seqlen, int d_qk, int num_heads);
int egs_linattn_eta(const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_qk,
                    const float* b_qk, int d_qk, int num_heads, double* eta, void* workspace,
                    size_t workspace_bytes, egs_stream_t stream);
int egs_softmax_eta(const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_qk,
                    const float* b_qk, int d_qk, int num_heads, double* eta, void* workspace,
                    size_t workspace_bytes, egs_stream_t stream);

/* Replaces get_eig_att_norm (analysis/eval_eig.py:137-174): n = exp(-norm_fn(Wn x + b + offset)) in fp32,
 * n == 0 -> 2e-23, eta_t = n_{t+1} / n_t.  w_n = the last H rows of Wvqkn (models/norm_attention.py:201-203).
 *   x : float[B, L, D]   w_n : float[H, D]   b_n, offset : float[H] or NULL   eta : double[B, L-1, H] */
#define EGS_NORM_EXP 0
#define EGS_NORM_ELU 1
#define EGS_NORM_SOFTPLUS 2
#define EGS_NORM_SIGMOID 3
int egs_normattn_eta(const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_n,
                     const float* b_n, const float* offset, int norm_fn, int num_heads, double* eta,
                     egs_stream_t stream);

/* ---- a9/a10: radius / phase + inclusive-edge threshold histograms -----------------------------------
 * Replaces threshold_analysis (analysis/eval_eig.py:335-362) and threshold_analysis_ssm (:364-391) with the
 * radius/phase transforms of :605-606, :614-615, :726-727, :734-735 fused in.  Bins (quirk Q12): bin 0 =
 * [0, e0], bin i = [e_{i-1}, e_i] (both inclusive), last = (e_last, inf); negatives / NaN fall in no bin.
 *   vals : [outer, T, inner] of `dtype` (EGS_F32/EGS_F64 real, EGS_C64/EGS_C128 complex)
 *   transform : EGS_HIST_VALUE (real input as is), EGS_HIST_RADIUS, EGS_HIST_PHASE_DEG (complex or real input;
 *               real input is treated as imag = 0), EGS_HIST_ZERO (0*value, quirk Q13)
 *   edges : HOST double[nedges] (nedges <= 15)
 *   pct   : double[nedges+1, outer, inner] = count / T * 100. */
#define EGS_HIST_VALUE 0
#define EGS_HIST_RADIUS 1
#define EGS_HIST_PHASE_DEG 2
#define EGS_HIST_ZERO 3
int egs_threshold_hist(const void* vals, int dtype, int transform, int64_t outer, int64_t t_len,
                       int64_t inner, const double* edges_host, int nedges, double* pct,
                       egs_stream_t stream);

/* ---- 8(f) rank 2: block forward (token mixers that produce the next layer's input, quirk Q1) ----------
 * One fp32 recurrence, evaluated token by token per batch row:
 *     S_t = a_t S_{t-1} + k_t (s_t v_t)^T,      y_t = w_t (q_t^T S_t) + skip * v_t
 * Replaces
 *   SelfLinAttention.forward   (models/attention.py:58-83)      phi = 1, normalise = 1 (w_t = 1 / q_t.sum_{s<=t} k_s)
 *   SelfNormAttention.forward / FlashNormAttention.forward (models/norm_attention.py:61-89, :129-156; the fla
 *       chunk_simple_gla(q, k, v, None, scale) call)            phi = approx_fn == "elu", k_scale = scale, w = n
 *   mamba_chunk_scan_combined(x, dt, A, B, C, D=D, z=None) as called by SSD / SSD_LTI.forward
 *       (models/mamba.py:138-150, :276-288; mamba_ssm 2.x)      q = C, k = B, v = x, decay = exp(dt A), vscale = dt, skip = D
 * Layout (all float32; strides in elements):
 *   q, k   : [B, L, G, d_state] through (qk_batch_stride, qk_token_stride, qk_group_stride), d_state contiguous;
 *            G = channels / cols_per_group (attention: heads; SSD: B/C groups)
 *   v      : [B, L, channels] through (v_batch_stride, v_token_stride), channels contiguous
 *   decay, vscale : [B, L, channels / cols_per_decay] contiguous, or NULL (= 1)
 *   w      : [B, L, G] contiguous or NULL (= 1); ignored when normalise != 0
 *   skip   : [channels] or NULL (= 0)
 *   y      : [B, L, channels] contiguous
 * d_state <= 512.  Three kernels serve it.  With d_state % 4 == 0, d_state <= 128 and 16-byte aligned q / k rows
 * (pointer and strides) the recurrence runs chunk-wise on the matrix cores and needs NO workspace: the un-gated form
 * (decay == NULL, skip == NULL), and the gated form without w / phi / normalise / k_scale when cols_per_group and
 * cols_per_decay are multiples of 32.  Everything else runs token by token and needs the workspace (compact transformed
 * q, k and the normaliser).  `flags` describes the call to the query with the EGS_REC_* bits below. */
#define EGS_REC_GATED 1      /* decay / vscale given */
#define EGS_REC_W 2          /* w given (and normalise == 0) */
#define EGS_REC_SKIP 4       /* skip given */
#define EGS_REC_PHI 8        /* phi != 0 */
#define EGS_REC_NORMALISE 16 /* normalise != 0 */
#define EGS_REC_KSCALE 32    /* k_scale != 1 */
#define EGS_REC_ALIGNED 64   /* q, k pointers 16-byte aligned and their strides multiples of 4 elements */
size_t egs_linrec_workspace_bytes(int64_t batch, int64_t seqlen, int channels, int cols_per_group, int cols_per_decay,
                                  int d_state, unsigned flags);
int egs_linear_recurrence_fwd(const float* q, const float* k, int64_t qk_batch_stride, int64_t qk_token_stride,
                              int64_t qk_group_stride, const float* v, int64_t v_batch_stride,
                              int64_t v_token_stride, const float* decay, const float* vscale, const float* w,
                              const float* skip, float* y, int64_t batch, int64_t seqlen, int channels,
                              int d_state, int cols_per_group, int cols_per_decay, int phi, float k_scale,
                              int normalise, void* workspace, size_t workspace_bytes, egs_stream_t stream);

/* Replaces SelfAttention.forward (models/attention.py:17-35) and, in fp32 instead of fp16, the flash_attn call of
 * FlashAttention.forward (:46-55): y[b,t,h,:] = sum_{s<=t} softmax_s(scale * q_t.k_s) v_s, fp32.
 *   q, k : [B, L, H, head_dim] through (qk_batch_stride, qk_token_stride, qk_head_stride), head_dim contiguous
 *   v    : [B, L, H * v_dim]   through (v_batch_stride, v_token_stride), channels contiguous
 *   y    : [B, L, H * v_dim]   contiguous */
int egs_softmax_attention_fwd(const float* q, const float* k, int64_t qk_batch_stride, int64_t qk_token_stride,
                              int64_t qk_head_stride, const float* v, int64_t v_batch_stride, int64_t v_token_stride,
                              float* y, int64_t batch, int64_t seqlen, int num_heads, int head_dim, int v_dim,
                              float scale, egs_stream_t stream);

/* ---- 8(f) rank 4: softmax eta from q, k that already exist (forward hooks on a language model's q_proj / k_proj) --
 * Replaces get_eig_from_qkv_att_softmax (notebooks/lm_eigvals.ipynb, cell 13): the arithmetic of get_eig_att_softmax
 * without the projection.  qk : float[B, L, 2 d_qk] = per token the H query heads followed by the H key heads (grouped
 * query attention: the caller repeats each key head H / H_kv times); eta : double[B, L-1, H]. */
size_t egs_softmax_eta_qk_workspace_bytes(int64_t batch, int64_t seqlen, int num_heads);
int egs_softmax_eta_qk(const float* qk, int64_t batch, int64_t seqlen, int d_qk, int num_heads, double* eta,
                       void* workspace, size_t workspace_bytes, egs_stream_t stream);

/* The (q, k)-input form of the linear-attention eta (the scan half of get_eig_att_linear,
 * analysis/eval_eig.py:113-130): qk : float[B, L, 2 d_qk] = per token the H query heads followed by the H key heads, RAW
 * projections (phi = elu + 1 is applied inside); eta : double[B, L-1, H].  Same fp64 prefix state as egs_linattn_eta. */
size_t egs_linattn_eta_qk_workspace_bytes(int64_t batch, int64_t seqlen, int d_qk, int num_heads);
int egs_linattn_eta_qk(const float* qk, int64_t batch, int64_t seqlen, int d_qk, int num_heads, double* eta,
                       void* workspace, size_t workspace_bytes, egs_stream_t stream);

/* Depthwise causal conv1d + SiLU on channel-last activations: the conv blocks in front of the token mixers
 * (models/attention.py:130-155, models/norm_attention.py:225-244, models/mamba.py:96-108, :130-134 --
 * nn.Conv1d(groups = C, padding = K-1) truncated to L outputs, then nn.SiLU).
 *   x : float[B, L, C] through (x_batch_stride, x_token_stride), channels contiguous
 *   w : float[C, K] (= conv1d.weight[:, 0, :]), bias : float[C] or NULL, K <= 8;   y : float[B, L, C] contiguous */
int egs_causal_conv1d_silu(const float* x, int64_t x_batch_stride, int64_t x_token_stride, const float* w,
                           const float* bias, float* y, int64_t batch, int64_t seqlen, int channels, int kernel_size,
                           egs_stream_t stream);

/* LayerNorm over the last dimension, fp32, biased variance (nn.LayerNorm of the blocks: models/transformer.py:84-86,
 * :143-144, models/mamba.py:312-313).  x : float[rows, d] through x_row_stride (d <= 1024); gamma, beta : float[d] or
 * NULL; y : float[rows, d] contiguous. */
int egs_layer_norm(const float* x, int64_t x_row_stride, const float* gamma, const float* beta, float* y, int64_t rows,
                   int d, float eps, egs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EIGSPEC_H */
