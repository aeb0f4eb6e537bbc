// extern "C" entry points of libeigspec_hip.so (declared in include/eigspec.h).  Argument validation happens
// here, on the host, before any kernel is launched: a kernel never sees shapes it was not built for.
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include "egs_common.h"

namespace egs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return EGS_ERR_RUNTIME;
}

// launchers implemented in the kernel translation units
int diag_ssm_launch(int, const double*, const double*, const double*, int64_t, double*, hipStream_t);
int mamba2_lambda_launch(const float*, int64_t, int, int64_t, const float*, const float*, const float*, const float*,
                         int, float*, hipStream_t);
int mamba2_lti_launch(const float*, const float*, int, int64_t, float*, hipStream_t);
int normattn_eta_launch(const float*, int64_t, int64_t, int, const float*, const float*, const float*, int, int,
                        double*, hipStream_t);
int s4_dplr_launch(const double*, const double*, const double*, const double*, const double*, const double*,
                   const double*, int64_t, int, int, double*, hipStream_t);
int dplr_secular_launch(const double*, const double*, const double*, const double*, const double*, const double*, int64_t,
                        int, int, double*, int32_t*, hipStream_t);
int s4_cayley_map_launch(const double*, const int32_t*, const double*, int64_t, int, double*, hipStream_t);
size_t attn_workspace_bytes(int64_t, int64_t, int, int);
int attn_eta_launch(bool, const float*, int64_t, int64_t, int, const float*, const float*, int, int, double*, void*,
                    size_t, hipStream_t);
int threshold_hist_launch(const void*, int, int, int64_t, int64_t, int64_t, const double*, int, double*, hipStream_t);
size_t linrec_workspace_bytes(int64_t, int64_t, int, int, int);
int linrec_path(int, int, int, unsigned);
int layer_norm_launch(const float*, int64_t, const float*, const float*, float*, int64_t, int, float, hipStream_t);
int causal_conv1d_silu_launch(const float*, int64_t, int64_t, const float*, const float*, float*, int64_t, int64_t, int, int,
                              hipStream_t);
size_t softmax_eta_qk_workspace_bytes(int64_t, int64_t, int);
size_t linattn_eta_qk_workspace_bytes(int64_t, int64_t, int, int);
int linattn_eta_qk_launch(const float*, int64_t, int64_t, int, int, double*, void*, size_t, hipStream_t);
int softmax_eta_qk_launch(const float*, int64_t, int64_t, int, int, double*, void*, size_t, hipStream_t);
int softmax_attn_fwd_launch(const float*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, int64_t, float*,
                            int64_t, int64_t, int, int, int, float, hipStream_t);
int linrec_launch(const float*, const float*, int64_t, int64_t, int64_t, const float*, int64_t, int64_t, const float*,
                  const float*, const float*, const float*, float*, int64_t, int64_t, int, int, int, int, int, float, int,
                  void*, size_t, hipStream_t);
size_t eig_real_lds_workspace_bytes(int64_t, int);
int eig_real_lds_launch(const double*, int64_t, int, int, int64_t, double*, int32_t*, void*, size_t, hipStream_t);
int hess_real_launch(const double*, int64_t, int, int, int64_t, double*, int32_t*, hipStream_t);
int hqr_real_launch(const double*, int64_t, int, double*, int32_t*, int, hipStream_t, unsigned long long*);
size_t eig_cplx_workspace_bytes(int64_t, int);
int eig_cplx_launch(const void*, bool, int64_t, int, int, int64_t, double*, int32_t*, void*, size_t, hipStream_t);
size_t eig_zlarge_workspace_bytes(int64_t, int);
int zhess_blocked_launch(const void*, bool, int64_t, int, int, int64_t, double*, int32_t*, void*, hipStream_t);
int zmsqr_launch(int64_t, int, double*, int32_t*, void*, double*, hipStream_t);

// float / complex64 -> double / complex128, packed [nmat, n, n] (the single-precision entry of the dense solver)
__global__ void widen_f32_kernel(const float* __restrict__ src, double* __restrict__ dst, int64_t nmat, int n, int lda,
                                 int64_t stride, int comps) {
    const int64_t per = (int64_t)n * n * comps;
    const int64_t total = nmat * per;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / per, e = idx - m * per;
        const int64_t el = e / comps, cpt = e - el * comps;
        const int64_t i = el / n, j = el - i * n;
        dst[idx] = (double)src[(m * stride + i * lda + j) * comps + cpt];
    }
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace egs

using namespace egs;

extern "C" {

int egs_version(void) { return EGS_VERSION; }

const char* egs_last_error_string(void) { return g_err; }

int egs_device_props_query(int device, egs_device_props* out) {
    EGS_ARG(out != nullptr, 2);
    hipDeviceProp_t p;
    EGS_HIP(hipGetDeviceProperties(&p, device));
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->lds_bytes_per_block = (int)p.sharedMemPerBlock;
    out->clock_khz = p.clockRate;
    out->hbm_bytes = (int64_t)p.totalGlobalMem;
    strncpy(out->arch, p.gcnArchName, sizeof(out->arch) - 1);
    out->arch[sizeof(out->arch) - 1] = 0;
    return 0;
}

int egs_diag_ssm_lambda(int mode, const double* p0, const double* p1, const double* p2, int64_t n,
                        double* out_lambda, egs_stream_t stream) {
    EGS_ARG(mode == EGS_DIAG_LRU || mode == EGS_DIAG_S5_ZOH, 1);
    EGS_ARG(n >= 0, 5);
    if (n == 0) return 0;
    EGS_ARG(p0 != nullptr, 2);
    EGS_ARG(p1 != nullptr, 3);
    EGS_ARG(mode == EGS_DIAG_LRU || p2 != nullptr, 4);
    EGS_ARG(out_lambda != nullptr, 6);
    return diag_ssm_launch(mode, p0, p1, p2, n, out_lambda, as_stream(stream));
}

int egs_s4_dplr_build(const double* lambda_re, const double* lambda_im, const double* p_re, const double* p_im,
                      const double* log_step, int64_t nmat, int n, double* ab, egs_stream_t stream) {
    EGS_ARG(nmat >= 0, 6);
    EGS_ARG(n >= 0 && n <= 2048, 7);
    if (nmat == 0 || n == 0) return 0;
    EGS_ARG(lambda_re != nullptr, 1);
    EGS_ARG(lambda_im != nullptr, 2);
    EGS_ARG(p_re != nullptr, 3);
    EGS_ARG(p_im != nullptr, 4);
    EGS_ARG(log_step != nullptr, 5);
    EGS_ARG(ab != nullptr, 8);
    return s4_dplr_launch(lambda_re, lambda_im, p_re, p_im, p_re, p_im, log_step, nmat, n, 1, ab, as_stream(stream));
}

int egs_dplr_rank1_eigvals(const double* lambda_re, const double* lambda_im, const double* p_re, const double* p_im,
                           const double* q_re, const double* q_im, int64_t nmat, int n, int clip_lambda, double* a,
                           int32_t* info, egs_stream_t stream) {
    EGS_ARG(nmat >= 0, 7);
    EGS_ARG(n >= 0 && n <= 2048, 8);
    if (nmat == 0 || n == 0) return 0;
    EGS_ARG(lambda_re != nullptr, 1);
    EGS_ARG(lambda_im != nullptr, 2);
    EGS_ARG(p_re != nullptr, 3);
    EGS_ARG(p_im != nullptr, 4);
    EGS_ARG(q_re != nullptr, 5);
    EGS_ARG(q_im != nullptr, 6);
    EGS_ARG(clip_lambda == 0 || clip_lambda == 1, 9);
    EGS_ARG(a != nullptr, 10);
    EGS_ARG(info != nullptr, 11);
    return dplr_secular_launch(lambda_re, lambda_im, p_re, p_im, q_re, q_im, nmat, n, clip_lambda, a, info,
                               as_stream(stream));
}

int egs_s4_cayley_map(const double* a, const int32_t* set_index, const double* log_step, int64_t ncol, int n, double* w,
                      egs_stream_t stream) {
    EGS_ARG(ncol >= 0, 4);
    EGS_ARG(n >= 0, 5);
    if (ncol == 0 || n == 0) return 0;
    EGS_ARG(a != nullptr, 1);
    EGS_ARG(log_step != nullptr, 3);
    EGS_ARG(w != nullptr, 6);
    return s4_cayley_map_launch(a, set_index, log_step, ncol, n, w, as_stream(stream));
}

int egs_s4_dplr_build_pq(const double* lambda_re, const double* lambda_im, const double* p_re, const double* p_im,
                         const double* q_re, const double* q_im, const double* log_step, int64_t nmat, int n,
                         int clip_lambda, double* ab, egs_stream_t stream) {
    EGS_ARG(nmat >= 0, 8);
    EGS_ARG(n >= 0 && n <= 2048, 9);
    if (nmat == 0 || n == 0) return 0;
    EGS_ARG(lambda_re != nullptr, 1);
    EGS_ARG(lambda_im != nullptr, 2);
    EGS_ARG(p_re != nullptr, 3);
    EGS_ARG(p_im != nullptr, 4);
    EGS_ARG(q_re != nullptr, 5);
    EGS_ARG(q_im != nullptr, 6);
    EGS_ARG(log_step != nullptr, 7);
    EGS_ARG(clip_lambda == 0 || clip_lambda == 1, 10);
    EGS_ARG(ab != nullptr, 11);
    return s4_dplr_launch(lambda_re, lambda_im, p_re, p_im, q_re, q_im, log_step, nmat, n, clip_lambda, ab,
                          as_stream(stream));
}

static size_t eig_core_workspace_bytes(bool real, int64_t nmat, int n) {
    size_t b;
    if (real && n <= 64) b = eig_real_lds_workspace_bytes(nmat, n);
    else b = eig_cplx_workspace_bytes(nmat, n);
    return b < 256 ? 256 : b;
}

size_t egs_eig_workspace_bytes(int dtype, int64_t nmat, int n) {
    if (nmat <= 0 || n <= 0) return 256;
    const bool real = dtype == EGS_F64 || dtype == EGS_F32;
    size_t b = eig_core_workspace_bytes(real, nmat, n);
    if (dtype == EGS_F32 || dtype == EGS_C64) b = align256(b) + (size_t)nmat * n * n * (real ? 8 : 16);
    return b;
}

// widens single-precision input into the tail of the workspace; on return *a / *lda / *stride describe the fp64 copy
static int widen_if_single(const void** a, int dtype, int64_t nmat, int n, int* lda, int64_t* stride, void* workspace,
                           size_t workspace_bytes, hipStream_t stream) {
    if (dtype != EGS_F32 && dtype != EGS_C64) return 0;
    const bool real = dtype == EGS_F32;
    const size_t core = align256(eig_core_workspace_bytes(real, nmat, n));
    const size_t need = core + (size_t)nmat * n * n * (real ? 8 : 16);
    if (workspace_bytes < need) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, need);
        return -9;
    }
    double* dst = reinterpret_cast<double*>(static_cast<char*>(workspace) + core);
    const int64_t total = nmat * (int64_t)n * n * (real ? 1 : 2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(widen_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<const float*>(*a), dst,
                       nmat, n, *lda, *stride, real ? 1 : 2);
    EGS_LAUNCH_CHECK();
    *a = dst;
    *lda = n;
    *stride = (int64_t)n * n;
    return 0;
}

int egs_eig_general_batched(const void* a, int dtype, int64_t nmat, int n, int lda, int64_t stride, double* w,
                            int32_t* info, void* workspace, size_t workspace_bytes, egs_stream_t stream) {
    EGS_ARG(dtype >= EGS_F32 && dtype <= EGS_C128, 2);
    EGS_ARG(nmat >= 0, 3);
    EGS_ARG(n >= 0 && n <= 1024, 4);
    if (nmat == 0 || n == 0) return 0;
    EGS_ARG(a != nullptr, 1);
    EGS_ARG(lda >= n, 5);
    EGS_ARG(stride >= (int64_t)(n - 1) * lda + n || nmat == 1, 6);
    EGS_ARG(w != nullptr, 7);
    EGS_ARG(info != nullptr, 8);
    EGS_ARG(workspace != nullptr, 9);
    const bool real = dtype == EGS_F64 || dtype == EGS_F32;
    const int st = widen_if_single(&a, dtype, nmat, n, &lda, &stride, workspace, workspace_bytes, as_stream(stream));
    if (st) return st;
    // the widened copy lives in the tail of the workspace: the solver may only use the core region in front of it (it
    // sizes its chunk and places its work-queue counter from the byte count it is given)
    if (dtype == EGS_F32 || dtype == EGS_C64) workspace_bytes = align256(eig_core_workspace_bytes(real, nmat, n));
    if (real && n <= 64)
        return eig_real_lds_launch(static_cast<const double*>(a), nmat, n, lda, stride, w, info, workspace,
                                   workspace_bytes, as_stream(stream));
    return eig_cplx_launch(a, real, nmat, n, lda, stride, w, info, workspace, workspace_bytes, as_stream(stream));
}

int egs_hessenberg_general_batched(const void* a, int dtype, int64_t nmat, int n, int lda, int64_t stride, double* h,
                                   double* w, int32_t* info, egs_stream_t stream) {
    EGS_ARG(dtype == EGS_F64 || dtype == EGS_C128, 2);
    EGS_ARG(nmat >= 0, 3);
    EGS_ARG(n > 64 && n <= 1024, 4);
    if (nmat == 0) return 0;
    EGS_ARG(a != nullptr, 1);
    EGS_ARG(lda >= n, 5);
    EGS_ARG(stride >= (int64_t)(n - 1) * lda + n || nmat == 1, 6);
    EGS_ARG(h != nullptr, 7);
    EGS_ARG(w != nullptr, 8);
    EGS_ARG(info != nullptr, 9);
    return zhess_blocked_launch(a, dtype == EGS_F64, nmat, n, lda, stride, w, info, h, as_stream(stream));
}

int egs_hseqr_general_batched(double* h, int64_t nmat, int n, double* w, int32_t* info, double* phase_cycles,
                              egs_stream_t stream) {
    EGS_ARG(nmat >= 0, 2);
    EGS_ARG(n > 64 && n <= 1024, 3);
    if (nmat == 0) return 0;
    EGS_ARG(h != nullptr, 1);
    EGS_ARG(w != nullptr, 4);
    EGS_ARG(info != nullptr, 5);
    return zmsqr_launch(nmat, n, w, info, h, phase_cycles, as_stream(stream));
}

int egs_hessenberg_batched(const double* a, int64_t nmat, int n, int lda, int64_t stride, double* h,
                           egs_stream_t stream) {
    EGS_ARG(nmat >= 0, 2);
    EGS_ARG(n >= 0 && n <= 64, 3);
    if (nmat == 0 || n == 0) return 0;
    EGS_ARG(a != nullptr, 1);
    EGS_ARG(lda >= n, 4);
    EGS_ARG(stride >= (int64_t)(n - 1) * lda + n || nmat == 1, 5);
    EGS_ARG(h != nullptr, 6);
    return hess_real_launch(a, nmat, n, lda, stride, h, nullptr, as_stream(stream));   // no scaling: h ~ a exactly
}

int egs_hseqr_batched(const double* h, int64_t nmat, int n, double* w, int32_t* info, void* queue,
                      egs_stream_t stream) {
    EGS_ARG(nmat >= 0, 2);
    EGS_ARG(n >= 0 && n <= 64, 3);
    if (nmat == 0 || n == 0) return 0;
    EGS_ARG(h != nullptr, 1);
    EGS_ARG(w != nullptr, 4);
    EGS_ARG(info != nullptr, 5);
    return hqr_real_launch(h, nmat, n, w, info, 0, as_stream(stream), static_cast<unsigned long long*>(queue));
}

int egs_mamba2_lambda(const float* x, int64_t ntok, int d_model, int64_t ldx, const float* w_dt, const float* b_dt,
                      const float* dt_bias, const float* a_log, int nheads, float* out, egs_stream_t stream) {
    EGS_ARG(ntok >= 0, 2);
    EGS_ARG(d_model > 0, 3);
    EGS_ARG(ldx >= d_model, 4);
    EGS_ARG(nheads > 0 && nheads <= 1024, 9);
    if (ntok == 0) return 0;
    EGS_ARG(x != nullptr, 1);
    EGS_ARG(w_dt != nullptr, 5);
    EGS_ARG(dt_bias != nullptr, 7);
    EGS_ARG(a_log != nullptr, 8);
    EGS_ARG(out != nullptr, 10);
    return mamba2_lambda_launch(x, ntok, d_model, ldx, w_dt, b_dt, dt_bias, a_log, nheads, out, as_stream(stream));
}

int egs_mamba2_lti_lambda(const float* a_param, const float* beta, int nheads, int64_t ntok, float* out,
                          egs_stream_t stream) {
    EGS_ARG(nheads > 0, 3);
    EGS_ARG(ntok >= 0, 4);
    if (ntok == 0) return 0;
    EGS_ARG(a_param != nullptr, 1);
    EGS_ARG(beta != nullptr, 2);
    EGS_ARG(out != nullptr, 5);
    return mamba2_lti_launch(a_param, beta, nheads, ntok, out, as_stream(stream));
}

size_t egs_attn_workspace_bytes(int64_t batch, int64_t seqlen, int d_qk, int num_heads) {
    if (batch <= 0 || seqlen <= 0 || d_qk <= 0 || num_heads <= 0) return 256;
    return attn_workspace_bytes(batch, seqlen, d_qk, num_heads);
}

static int attn_common(bool softmax, const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_qk,
                       const float* b_qk, int d_qk, int num_heads, double* eta, void* workspace,
                       size_t workspace_bytes, egs_stream_t stream) {
    EGS_ARG(batch >= 0, 2);
    EGS_ARG(seqlen >= 0, 3);
    EGS_ARG(d_model > 0, 4);
    EGS_ARG(d_qk > 0, 7);
    EGS_ARG(num_heads > 0 && d_qk % num_heads == 0 && num_heads <= 65535, 8);
    if (batch == 0 || seqlen <= 1) return 0;
    EGS_ARG(x != nullptr, 1);
    EGS_ARG(w_qk != nullptr, 5);
    EGS_ARG(eta != nullptr, 9);
    EGS_ARG(workspace != nullptr, 10);
    return attn_eta_launch(softmax, x, batch, seqlen, d_model, w_qk, b_qk, d_qk, num_heads, eta, workspace,
                           workspace_bytes, as_stream(stream));
}

int egs_linattn_eta(const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_qk, const float* b_qk,
                    int d_qk, int num_heads, double* eta, void* workspace, size_t workspace_bytes,
                    egs_stream_t stream) {
    return attn_common(false, x, batch, seqlen, d_model, w_qk, b_qk, d_qk, num_heads, eta, workspace, workspace_bytes,
                       stream);
}

int egs_softmax_eta(const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_qk, const float* b_qk,
                    int d_qk, int num_heads, double* eta, void* workspace, size_t workspace_bytes,
                    egs_stream_t stream) {
    return attn_common(true, x, batch, seqlen, d_model, w_qk, b_qk, d_qk, num_heads, eta, workspace, workspace_bytes,
                       stream);
}

int egs_normattn_eta(const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_n, const float* b_n,
                     const float* offset, int norm_fn, int num_heads, double* eta, egs_stream_t stream) {
    EGS_ARG(batch >= 0, 2);
    EGS_ARG(seqlen >= 0, 3);
    EGS_ARG(d_model > 0, 4);
    EGS_ARG(norm_fn >= EGS_NORM_EXP && norm_fn <= EGS_NORM_SIGMOID, 8);
    EGS_ARG(num_heads > 0 && num_heads <= 1024, 9);
    if (batch == 0 || seqlen <= 1) return 0;
    EGS_ARG(x != nullptr, 1);
    EGS_ARG(w_n != nullptr, 5);
    EGS_ARG(eta != nullptr, 10);
    return normattn_eta_launch(x, batch, seqlen, d_model, w_n, b_n, offset, norm_fn, num_heads, eta,
                               as_stream(stream));
}

int egs_threshold_hist(const void* vals, int dtype, int transform, int64_t outer, int64_t t_len, int64_t inner,
                       const double* edges_host, int nedges, double* pct, egs_stream_t stream) {
    EGS_ARG(dtype >= EGS_F32 && dtype <= EGS_C128, 2);
    EGS_ARG(transform >= EGS_HIST_VALUE && transform <= EGS_HIST_ZERO, 3);
    EGS_ARG(!(transform == EGS_HIST_VALUE && (dtype == EGS_C64 || dtype == EGS_C128)), 3);
    EGS_ARG(outer >= 0 && outer <= 65535, 4);
    EGS_ARG(t_len >= 0, 5);
    EGS_ARG(inner >= 0, 6);
    EGS_ARG(edges_host != nullptr, 7);
    EGS_ARG(nedges >= 1 && nedges <= 15, 8);
    if (outer == 0 || inner == 0) return 0;
    EGS_ARG(vals != nullptr || t_len == 0, 1);
    EGS_ARG(pct != nullptr, 9);
    return threshold_hist_launch(vals, dtype, transform, outer, t_len, inner, edges_host, nedges, pct,
                                 as_stream(stream));
}

size_t egs_linrec_workspace_bytes(int64_t batch, int64_t seqlen, int channels, int cols_per_group, int cols_per_decay,
                                  int d_state, unsigned flags) {
    if (batch <= 0 || seqlen <= 0 || channels <= 0 || cols_per_group <= 0 || cols_per_decay <= 0 || d_state <= 0 ||
        channels % cols_per_group != 0)
        return 512;
    if (linrec_path(d_state, cols_per_group, cols_per_decay, flags) != 0) return 512;   // matrix-core paths: no scratch
    return linrec_workspace_bytes(batch, seqlen, channels, cols_per_group, d_state);
}

int egs_linear_recurrence_fwd(const float* q, const float* k, int64_t qk_batch_stride, int64_t qk_token_stride,
                              int64_t qk_group_stride, const float* v, int64_t v_batch_stride,
                              int64_t v_token_stride, const float* decay, const float* vscale, const float* w,
                              const float* skip, float* y, int64_t batch, int64_t seqlen, int channels,
                              int d_state, int cols_per_group, int cols_per_decay, int phi, float k_scale,
                              int normalise, void* workspace, size_t workspace_bytes, egs_stream_t stream) {
    EGS_ARG(batch >= 0 && batch <= 65535, 14);
    EGS_ARG(seqlen >= 0, 15);
    EGS_ARG(channels > 0, 16);
    EGS_ARG(d_state > 0 && d_state <= 512, 17);
    EGS_ARG(cols_per_group > 0 && channels % cols_per_group == 0, 18);
    EGS_ARG(cols_per_decay > 0 && channels % cols_per_decay == 0, 19);
    if (batch == 0 || seqlen == 0) return 0;
    EGS_ARG(q != nullptr, 1);
    EGS_ARG(k != nullptr, 2);
    EGS_ARG(v != nullptr, 6);
    EGS_ARG(y != nullptr, 13);
    EGS_ARG(workspace != nullptr, 23);
    return linrec_launch(q, k, qk_batch_stride, qk_token_stride, qk_group_stride, v, v_batch_stride, v_token_stride,
                         decay, vscale, w, skip, y, batch, seqlen, channels, d_state, cols_per_group, cols_per_decay,
                         phi, k_scale, normalise, workspace, workspace_bytes, as_stream(stream));
}

int egs_softmax_attention_fwd(const float* q, const float* k, int64_t qk_batch_stride, int64_t qk_token_stride,
                              int64_t qk_head_stride, const float* v, int64_t v_batch_stride, int64_t v_token_stride,
                              float* y, int64_t batch, int64_t seqlen, int num_heads, int head_dim, int v_dim,
                              float scale, egs_stream_t stream) {
    EGS_ARG(batch >= 0 && batch <= 65535, 10);
    EGS_ARG(seqlen >= 0, 11);
    EGS_ARG(num_heads > 0, 12);
    EGS_ARG(head_dim > 0, 13);
    EGS_ARG(v_dim > 0 && (int64_t)num_heads * ((v_dim + 127) / 128) <= 65535, 14);
    if (batch == 0 || seqlen == 0) return 0;
    EGS_ARG(q != nullptr, 1);
    EGS_ARG(k != nullptr, 2);
    EGS_ARG(v != nullptr, 6);
    EGS_ARG(y != nullptr, 9);
    return softmax_attn_fwd_launch(q, k, qk_batch_stride, qk_token_stride, qk_head_stride, v, v_batch_stride,
                                   v_token_stride, y, batch, seqlen, num_heads, head_dim, v_dim, scale,
                                   as_stream(stream));
}

size_t egs_softmax_eta_qk_workspace_bytes(int64_t batch, int64_t seqlen, int num_heads) {
    if (batch <= 0 || seqlen <= 0 || num_heads <= 0) return 512;
    return softmax_eta_qk_workspace_bytes(batch, seqlen, num_heads);
}

int egs_softmax_eta_qk(const float* qk, int64_t batch, int64_t seqlen, int d_qk, int num_heads, double* eta,
                       void* workspace, size_t workspace_bytes, egs_stream_t stream) {
    EGS_ARG(batch >= 0 && batch <= 65535, 2);
    EGS_ARG(seqlen >= 0, 3);
    EGS_ARG(d_qk > 0, 4);
    EGS_ARG(num_heads > 0 && d_qk % num_heads == 0 && num_heads <= 65535, 5);
    if (batch == 0 || seqlen <= 1) return 0;
    EGS_ARG(qk != nullptr, 1);
    EGS_ARG(eta != nullptr, 6);
    EGS_ARG(workspace != nullptr, 7);
    return softmax_eta_qk_launch(qk, batch, seqlen, d_qk, num_heads, eta, workspace, workspace_bytes, as_stream(stream));
}

size_t egs_linattn_eta_qk_workspace_bytes(int64_t batch, int64_t seqlen, int d_qk, int num_heads) {
    if (batch <= 0 || seqlen <= 0 || d_qk <= 0 || num_heads <= 0) return 512;
    return linattn_eta_qk_workspace_bytes(batch, seqlen, d_qk, num_heads);
}

int egs_linattn_eta_qk(const float* qk, int64_t batch, int64_t seqlen, int d_qk, int num_heads, double* eta,
                       void* workspace, size_t workspace_bytes, egs_stream_t stream) {
    EGS_ARG(batch >= 0, 2);
    EGS_ARG(seqlen >= 0, 3);
    EGS_ARG(d_qk > 0, 4);
    EGS_ARG(num_heads > 0 && d_qk % num_heads == 0 && num_heads <= 65535, 5);
    if (batch == 0 || seqlen <= 1) return 0;
    EGS_ARG(qk != nullptr, 1);
    EGS_ARG(eta != nullptr, 6);
    EGS_ARG(workspace != nullptr, 7);
    return linattn_eta_qk_launch(qk, batch, seqlen, d_qk, num_heads, eta, workspace, workspace_bytes, as_stream(stream));
}

int egs_causal_conv1d_silu(const float* x, int64_t x_batch_stride, int64_t x_token_stride, const float* w,
                           const float* bias, float* y, int64_t batch, int64_t seqlen, int channels, int kernel_size,
                           egs_stream_t stream) {
    EGS_ARG(batch >= 0 && batch <= 65535, 7);
    EGS_ARG(seqlen >= 0 && (seqlen + 15) / 16 <= 65535, 8);
    EGS_ARG(channels > 0, 9);
    EGS_ARG(kernel_size >= 1 && kernel_size <= 8, 10);
    if (batch == 0 || seqlen == 0) return 0;
    EGS_ARG(x != nullptr, 1);
    EGS_ARG(w != nullptr, 4);
    EGS_ARG(y != nullptr, 6);
    return causal_conv1d_silu_launch(x, x_batch_stride, x_token_stride, w, bias, y, batch, seqlen, channels, kernel_size,
                                     as_stream(stream));
}

int egs_layer_norm(const float* x, int64_t x_row_stride, const float* gamma, const float* beta, float* y, int64_t rows,
                   int d, float eps, egs_stream_t stream) {
    EGS_ARG(rows >= 0 && (rows + 3) / 4 <= 2147483647LL, 6);
    EGS_ARG(d > 0 && d <= 1024, 7);
    EGS_ARG(x_row_stride >= d, 2);
    if (rows == 0) return 0;
    EGS_ARG(x != nullptr, 1);
    EGS_ARG(y != nullptr, 5);
    return layer_norm_launch(x, x_row_stride, gamma, beta, y, rows, d, eps, as_stream(stream));
}

}  // extern "C"
