// HBM-bound streaming kernels of the eigenvalue-analysis path:
//   K1  diagonal SSM spectra (LRU / S5 closed forms)              analysis/eval_eig.py:303-329
//   K2  Mamba-2 per-token lambda_t (projection slice + softplus + exp)   analysis/eval_eig.py:176-190
//       pseudo-LTI broadcast                                        analysis/eval_eig.py:192-205
//   norm-attention ratios n_{t+1}/n_t                               analysis/eval_eig.py:137-174
//   S4 DPLR bilinear discretisation Ab                              analysis/eval_eig.py:254-273
//
// K2 and the norm-attention kernel share one skeleton: a "skinny" projection z[tok, h] = x[tok, :] . W[h, :]
// with only a handful of output rows (nheads / num_heads) per token.  x is read exactly once with 16-byte
// coalesced loads (G lanes per token, each lane a float4 slice of the row), W sits in LDS, the partial dots
// are reduced with cross-lane shuffles, and the family-specific elementwise epilogue is fused in, so the
// HBM traffic is the algorithmic 4*D bytes in + 4*nheads (8*H for eta) bytes out per token.
#include "egs_common.h"

namespace egs {

// ---------------------------------------------------------------------------------------------------
// elementwise helpers reproducing torch's fp32 functions
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f32(float v) {      // F.softplus, beta = 1, threshold = 20
    return (v > 20.0f) ? v : log1pf(expf(v));
}
__device__ __forceinline__ float elu_f32(float v) {           // F.elu, alpha = 1
    return (v > 0.0f) ? v : expm1f(v);
}
__device__ __forceinline__ float sigmoid_f32(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---------------------------------------------------------------------------------------------------
// K1
// ---------------------------------------------------------------------------------------------------
__global__ void diag_ssm_kernel(int mode, const double* __restrict__ p0, const double* __restrict__ p1,
                                const double* __restrict__ p2, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double mag, ang;
    if (mode == EGS_DIAG_LRU) {        // exp(-exp(nu_log) + i exp(theta_log))
        mag = exp(-exp(p0[i]));
        ang = exp(p1[i]);
    } else {                           // exp((re + i im) * exp(log_step))
        const double step = exp(p2[i]);
        mag = exp(p0[i] * step);
        ang = p1[i] * step;
    }
    double s, c;
    sincos(ang, &s, &c);
    out[2 * i] = mag * c;
    out[2 * i + 1] = mag * s;
}

// ---------------------------------------------------------------------------------------------------
// skinny projection: z[tok][h] for a tile of tokens, into LDS
// ---------------------------------------------------------------------------------------------------
constexpr int kProjThreads = 256;
constexpr int kProjHeadsPerPass = 8;

// Computes z[t - tok0][h] = dot(x[t, 0:D], W[h, 0:D]) for t in [tok0, tok1), h in [0, nh) into zs (LDS,
// row stride nh).  ws = W staged in LDS, row stride D.  VEC = 4 needs D % 4 == 0 and 16-byte aligned rows.
template <int VEC>
__device__ __forceinline__ void project_tile(const float* __restrict__ x, int64_t ldx, int d_model,
                                             int64_t tok0, int64_t tok1, const float* ws, int nh,
                                             float* zs, int g_lanes) {
    const int tid = threadIdx.x;
    const int gl = tid & (g_lanes - 1);                 // lane within the token group
    const int grp = tid / g_lanes;                      // token slot of this thread
    const int ngrp = kProjThreads / g_lanes;
    const int nchunk = (d_model + VEC * g_lanes - 1) / (VEC * g_lanes);
    for (int64_t tb = tok0; tb < tok1; tb += ngrp) {
        const int64_t tok = tb + grp;
        const bool live = tok < tok1;
        const float* xr = x + (live ? tok : tok0) * ldx;
        for (int h0 = 0; h0 < nh; h0 += kProjHeadsPerPass) {
            float acc[kProjHeadsPerPass];
#pragma unroll
            for (int q = 0; q < kProjHeadsPerPass; ++q) acc[q] = 0.0f;
            for (int c = 0; c < nchunk; ++c) {
                const int d = (c * g_lanes + gl) * VEC;
                if (d < d_model) {
                    if constexpr (VEC == 4) {
                        const float4 xv = *reinterpret_cast<const float4*>(xr + d);
#pragma unroll
                        for (int q = 0; q < kProjHeadsPerPass; ++q) {
                            if (h0 + q < nh) {
                                const float4 wv = *reinterpret_cast<const float4*>(ws + (h0 + q) * d_model + d);
                                acc[q] = fmaf(xv.x, wv.x, acc[q]);
                                acc[q] = fmaf(xv.y, wv.y, acc[q]);
                                acc[q] = fmaf(xv.z, wv.z, acc[q]);
                                acc[q] = fmaf(xv.w, wv.w, acc[q]);
                            }
                        }
                    } else {
                        const float xv = xr[d];
#pragma unroll
                        for (int q = 0; q < kProjHeadsPerPass; ++q)
                            if (h0 + q < nh) acc[q] = fmaf(xv, ws[(h0 + q) * d_model + d], acc[q]);
                    }
                }
            }
            // reduce over the g_lanes lanes of the group (xor butterflies stay inside aligned groups)
#pragma unroll
            for (int q = 0; q < kProjHeadsPerPass; ++q) {
                float v = acc[q];
                for (int off = g_lanes >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                acc[q] = v;
            }
            if (live) {
#pragma unroll
                for (int q = 0; q < kProjHeadsPerPass; ++q)
                    if (gl == q && h0 + q < nh) zs[(tok - tok0) * nh + h0 + q] = acc[q];
            }
        }
    }
}

__host__ __device__ inline int proj_group_lanes(int d_model, int vec) {
    int g = 1;
    while (g < 64 && g * vec < d_model) g <<= 1;
    if (g < kProjHeadsPerPass) g = kProjHeadsPerPass;   // lane q of a group stores head q
    return g;
}

// family-specific epilogues ---------------------------------------------------------------------------
struct Mamba2Params {
    const float* b_dt;      // [nh] or null
    const float* dt_bias;   // [nh]
    const float* a_log;     // [nh]
    float* out;             // [ntok, nh]
};
struct NormParams {
    const float* b_n;       // [nh] or null
    const float* offset;    // [nh] or null
    int norm_fn;
    int64_t seqlen;
    double* eta;            // [B, L-1, nh]
};

template <int VEC, bool NORM>
__global__ __launch_bounds__(kProjThreads) void skinny_proj_kernel(
    const float* __restrict__ x, int64_t ntok, int d_model, int64_t ldx, const float* __restrict__ w, int nh,
    int tile_tokens, Mamba2Params mp, NormParams np) {
    extern __shared__ float lds_f[];
    float* ws = lds_f;                                     // [nh][D]
    float* zs = lds_f + (size_t)nh * d_model;              // [tile_tokens + 1][nh]
    for (int idx = threadIdx.x; idx < nh * d_model; idx += kProjThreads) ws[idx] = w[idx];
    __syncthreads();
    const int g_lanes = proj_group_lanes(d_model, VEC);
    for (int64_t tile = blockIdx.x; tile * tile_tokens < ntok; tile += gridDim.x) {
        const int64_t tok0 = tile * tile_tokens;
        int64_t tok1 = tok0 + tile_tokens + (NORM ? 1 : 0);     // norm needs the next token too
        if (tok1 > ntok) tok1 = ntok;
        project_tile<VEC>(x, ldx, d_model, tok0, tok1, ws, nh, zs, g_lanes);
        __syncthreads();
        const int ntile = (int)(tok1 - tok0);
        if constexpr (!NORM) {
            // lambda = exp(softplus(z + b + dt_bias) * -exp(A_log)), fp32 (analysis/eval_eig.py:178-184)
            for (int idx = threadIdx.x; idx < ntile * nh; idx += kProjThreads) {
                const int h = idx % nh;
                float z = zs[idx];
                if (mp.b_dt) z += mp.b_dt[h];
                const float dt = softplus_f32(z + mp.dt_bias[h]);
                const float a = -expf(mp.a_log[h]);
                mp.out[tok0 * nh + idx] = expf(dt * a);
            }
        } else {
            // n = exp(-norm_fn(z + b + offset)) in fp32, then float64 ratio n_{t+1}/n_t with 0 -> 2e-23
            // (analysis/eval_eig.py:161-169)
            for (int idx = threadIdx.x; idx < ntile * nh; idx += kProjThreads) {
                const int h = idx % nh;
                float z = zs[idx];
                if (np.b_n) z += np.b_n[h];
                if (np.offset) z += np.offset[h];
                float f;
                switch (np.norm_fn) {
                    case EGS_NORM_EXP: f = expf(z); break;
                    case EGS_NORM_ELU: f = elu_f32(z); break;
                    case EGS_NORM_SOFTPLUS: f = softplus_f32(z); break;
                    default: f = sigmoid_f32(z); break;
                }
                zs[idx] = expf(-f);
            }
            __syncthreads();
            const int nout = (tok0 + tile_tokens <= ntok ? tile_tokens : (int)(ntok - tok0));
            for (int idx = threadIdx.x; idx < nout * nh; idx += kProjThreads) {
                const int tt = idx / nh;
                const int h = idx - tt * nh;
                const int64_t tok = tok0 + tt;
                const int64_t b = tok / np.seqlen;
                const int64_t t = tok - b * np.seqlen;
                if (t < np.seqlen - 1) {
                    double n0 = (double)zs[idx];
                    double n1 = (double)zs[idx + nh];
                    if (n0 == 0.0) n0 = 2e-23;
                    if (n1 == 0.0) n1 = 2e-23;
                    np.eta[(b * (np.seqlen - 1) + t) * nh + h] = n1 / n0;
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// K2 fast path (common Mamba-2 shapes: D = 4*G*C with G in {16,32,64} lanes per token, nheads <= 8).
// Every wave streams blocks of TPW*U consecutive tokens: all U*C 16-byte loads of a lane are issued first
// (memory-level parallelism), W lives in registers, the U*NH partial dots per lane are combined across the G lanes
// of a token with a transpose-reduce (each butterfly step halves the values a lane carries: U*NH + log2 shuffles
// instead of U*NH*log2(G)), and the lane that ends up owning (token u, head h) runs the fused softplus/exp epilogue
// and stores out[tok][h] -- consecutive owners write consecutive addresses.  No LDS, no workgroup barrier.
// ---------------------------------------------------------------------------------------------------
// Transpose-reduce: CNT partial sums per lane, butterfly over xor offsets OFF, OFF/2, ..., 1.  While a lane still
// carries more than one value, a step sends the half it gives up and keeps the other half (upper lanes keep the
// upper half), so after log2(CNT) steps lane gl holds the complete sum of value index (top bits of gl).
template <int CNT, int OFF>
__device__ __forceinline__ void treduce(float* v, int gl) {
    if constexpr (OFF >= 1) {
        const bool upper = (gl & OFF) != 0;
        if constexpr (CNT > 1) {
            constexpr int H = CNT / 2;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const float send = upper ? v[i] : v[i + H];
                const float keep = upper ? v[i + H] : v[i];
                v[i] = keep + __shfl_xor(send, OFF, 64);
            }
            treduce<H, OFF / 2>(v, gl);
        } else {
            v[0] += __shfl_xor(v[0], OFF, 64);
            treduce<1, OFF / 2>(v, gl);
        }
    }
}

template <int G, int C, int NH>
__global__ __launch_bounds__(256) void mamba2_lambda_fast_kernel(const float* __restrict__ x, int64_t ntok, int64_t ldx,
                                                                 const float* __restrict__ w, int nh_real,
                                                                 const float* __restrict__ b_dt,
                                                                 const float* __restrict__ dt_bias,
                                                                 const float* __restrict__ a_log,
                                                                 float* __restrict__ out) {
    constexpr int D = 4 * G * C;
    constexpr int TPW = 64 / G;                       // tokens side by side in one wave
    constexpr int U = (G / NH) < 4 ? (G / NH) : 4;    // tokens per lane group per iteration (U*NH <= G)
    constexpr int V = U * NH;                         // partial sums per lane
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int grp = lane / G;
    // W slice of this lane, zero rows for the padded heads
    float4 wr[NH][C];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int c = 0; c < C; ++c)
            wr[h][c] = (h < nh_real) ? *reinterpret_cast<const float4*>(w + (int64_t)h * D + (c * G + gl) * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    // after the transpose-reduce lane gl owns value index gl / (G / V) = u * NH + h
    constexpr int OWN = G / V;
    const int idx = gl / OWN;
    const int my_u = idx / NH, my_h = idx % NH;
    const bool owner = (gl % OWN) == 0 && my_h < nh_real;
    float bias_h = 0.f, a_h = 0.f;
    if (my_h < nh_real) {
        bias_h = dt_bias[my_h] + (b_dt ? b_dt[my_h] : 0.f);
        a_h = -expf(a_log[my_h]);
    }
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int TOK_PER_IT = TPW * U;
    for (int64_t t0 = wave_id * TOK_PER_IT; t0 < ntok; t0 += nwaves * TOK_PER_IT) {
        float4 xv[U][C];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t tok = t0 + grp * U + u;
            const float* xr = x + (tok < ntok ? tok : 0) * ldx;
#pragma unroll
            for (int c = 0; c < C; ++c) xv[u][c] = *reinterpret_cast<const float4*>(xr + (c * G + gl) * 4);
        }
        float v[V];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    acc = fmaf(xv[u][c].x, wr[h][c].x, acc);
                    acc = fmaf(xv[u][c].y, wr[h][c].y, acc);
                    acc = fmaf(xv[u][c].z, wr[h][c].z, acc);
                    acc = fmaf(xv[u][c].w, wr[h][c].w, acc);
                }
                v[u * NH + h] = acc;
            }
        treduce<V, G / 2>(v, gl);          // transpose-reduce across the G lanes of the token group
        const int64_t tok = t0 + grp * U + my_u;
        if (owner && tok < ntok) {
            const float dt = softplus_f32(v[0] + bias_h);           // analysis/eval_eig.py:182
            out[tok * nh_real + my_h] = expf(dt * a_h);             // :184
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Norm-attention fast path: same streaming skeleton as mamba2_lambda_fast_kernel, with the ratio n_{t+1}/n_t formed
// inside the wave.  Every wave iteration covers T = TPW*U consecutive tokens and advances by T-1, i.e. consecutive
// iterations overlap by one token (1/T extra traffic) so that each ratio's two operands are always in the same
// iteration: after the transpose-reduce the owner of (token u, head h) finds (token u+1, head h) NH*OWN lanes up.
// ---------------------------------------------------------------------------------------------------
template <int G, int C, int NH>
__global__ __launch_bounds__(256) void normattn_fast_kernel(const float* __restrict__ x, int64_t ntok, int64_t ldx,
                                                            const float* __restrict__ w, int nh_real,
                                                            const float* __restrict__ b_n, const float* __restrict__ offset,
                                                            int norm_fn, int64_t seqlen, double* __restrict__ eta) {
    constexpr int D = 4 * G * C;
    constexpr int TPW = 64 / G;
    constexpr int U = (G / NH) < 4 ? (G / NH) : 4;
    constexpr int V = U * NH;
    constexpr int OWN = G / V;
    constexpr int T = TPW * U;                        // tokens per iteration; T-1 ratios are emitted
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int grp = lane / G;
    float4 wr[NH][C];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int c = 0; c < C; ++c)
            wr[h][c] = (h < nh_real) ? *reinterpret_cast<const float4*>(w + (int64_t)h * D + (c * G + gl) * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    const int idx = gl / OWN;
    const int my_u = idx / NH, my_h = idx % NH;
    const int my_slot = grp * U + my_u;               // position of this lane's token inside the iteration
    const bool owner = (gl % OWN) == 0 && my_h < nh_real;
    float bias_h = 0.f;
    if (my_h < nh_real) bias_h = (b_n ? b_n[my_h] : 0.f) + (offset ? offset[my_h] : 0.f);
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t t0 = wave_id * (T - 1); t0 + 1 < ntok; t0 += nwaves * (T - 1)) {
        float4 xv[U][C];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t tok = t0 + grp * U + u;
            const float* xr = x + (tok < ntok ? tok : ntok - 1) * ldx;
#pragma unroll
            for (int c = 0; c < C; ++c) xv[u][c] = *reinterpret_cast<const float4*>(xr + (c * G + gl) * 4);
        }
        float v[V];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    acc = fmaf(xv[u][c].x, wr[h][c].x, acc);
                    acc = fmaf(xv[u][c].y, wr[h][c].y, acc);
                    acc = fmaf(xv[u][c].z, wr[h][c].z, acc);
                    acc = fmaf(xv[u][c].w, wr[h][c].w, acc);
                }
                v[u * NH + h] = acc;
            }
        treduce<V, G / 2>(v, gl);
        // n = exp(-norm_fn(z + b + offset)) in fp32 (analysis/eval_eig.py:161-163)
        const float z = v[0] + bias_h;
        float f;
        switch (norm_fn) {
            case EGS_NORM_EXP: f = expf(z); break;
            case EGS_NORM_ELU: f = elu_f32(z); break;
            case EGS_NORM_SOFTPLUS: f = softplus_f32(z); break;
            default: f = sigmoid_f32(z); break;
        }
        const float n_me = expf(-f);
        const float n_next = __shfl_down(n_me, NH * OWN, 64);      // owner of (next token, same head)
        const int64_t tok = t0 + my_slot;
        if (owner && my_slot < T - 1 && tok + 1 < ntok) {
            const int64_t b = tok / seqlen;
            const int64_t t = tok - b * seqlen;
            if (t < seqlen - 1) {
                double n0 = (double)n_me, n1 = (double)n_next;     // :165-169, exact zeros -> 2e-23 (Q6)
                if (n0 == 0.0) n0 = 2e-23;
                if (n1 == 0.0) n1 = 2e-23;
                eta[(b * (seqlen - 1) + t) * nh_real + my_h] = n1 / n0;
            }
        }
    }
}

__global__ void mamba2_lti_kernel(const float* __restrict__ a_param, const float* __restrict__ beta, int nh,
                                  int64_t ntok, float* __restrict__ out) {
    // exp(beta * -softplus(A)) broadcast over tokens (analysis/eval_eig.py:194-199)
    const int64_t total = ntok * nh;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(idx % nh);
        out[idx] = expf(beta[h] * (-softplus_f32(a_param[h])));
    }
}

// ---------------------------------------------------------------------------------------------------
// S4 DPLR build: Ab = A1 A0,  A0 = g I + diag(L) - P Q^H,  A1 = D - (D P)(Q^H D)/(1 + Q^H D P), D = 1/(g - L)
// (discrete_DPLR takes P and Q separately, analysis/eval_eig.py:254-273; its only caller passes Q = P, :294)
// Expanding the product row i, column j (c = Q^H D P, r_i = D_i P_i, q_j = conj(Q_j)):
//   Ab[i][j] = D_i (g + L_i) [i==j] - r_i q_j - r_i q_j D_j (g + L_j) / (1 + c) + r_i q_j s / (1 + c)
// with s = sum_k conj(P_k) D_k P_k = c.  One workgroup per matrix; O(n^2) outputs, coalesced stores.
// ---------------------------------------------------------------------------------------------------
struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cconj(cplx a) { return {a.re, -a.im}; }
__device__ __forceinline__ cplx cinv(cplx a) {
    const double d = a.re * a.re + a.im * a.im;
    return {a.re / d, -a.im / d};
}

__global__ __launch_bounds__(256) void s4_dplr_kernel(const double* __restrict__ lam_re,
                                                      const double* __restrict__ lam_im,
                                                      const double* __restrict__ p_re,
                                                      const double* __restrict__ p_im,
                                                      const double* __restrict__ q_re,
                                                      const double* __restrict__ q_im,
                                                      const double* __restrict__ log_step, int n, int clip,
                                                      double* __restrict__ ab) {
    extern __shared__ double lds_d[];
    cplx* dl = reinterpret_cast<cplx*>(lds_d);        // D_i
    cplx* gl = dl + n;                                // g + L_i
    cplx* pp = gl + n;                                // P_i
    cplx* qq = pp + n;                                // Q_i
    __shared__ double red[2 * 256];
    const int64_t m = blockIdx.x;
    const double g = 2.0 / exp(log_step[m]);
    double cre = 0.0, cim = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double lre = lam_re[m * n + i];
        const cplx L = {clip ? fmin(lre, -1e-4) : lre, lam_im[m * n + i]};      // clip (Q11): the caller's, eval_eig.py:288
        const cplx P = {p_re[m * n + i], p_im[m * n + i]};
        const cplx Q = {q_re[m * n + i], q_im[m * n + i]};
        const cplx D = cinv({g - L.re, -L.im});
        dl[i] = D;
        gl[i] = {g + L.re, L.im};
        pp[i] = P;
        qq[i] = Q;
        const cplx t = cmul(cmul(cconj(Q), D), P);
        cre += t.re;
        cim += t.im;
    }
    red[threadIdx.x] = cre;
    red[256 + threadIdx.x] = cim;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[threadIdx.x] += red[threadIdx.x + s];
            red[256 + threadIdx.x] += red[256 + threadIdx.x + s];
        }
        __syncthreads();
    }
    const cplx c = {red[0], red[256]};
    const cplx inv1c = cinv({1.0 + c.re, c.im});
    double* out = ab + m * 2 * (int64_t)n * n;
    // A1[i][k] = D_i [i==k] - r_i conj(Q_k) D_k inv1c ;  A0[k][j] = gl_k [k==j] - P_k conj(Q_j)
    // Ab[i][j] = A1[i][j] gl_j - (sum_k A1[i][k] P_k) conj(Q_j)
    //   sum_k A1[i][k] P_k = D_i P_i - r_i inv1c c = r_i (1 - c inv1c) = r_i inv1c
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const int i = idx / n;
        const int j = idx - i * n;
        const cplx ri = cmul(dl[i], pp[i]);
        const cplx qj = cconj(qq[j]);
        cplx a1ij = cmul(cmul(cmul(ri, qj), dl[j]), inv1c);
        a1ij = {-a1ij.re, -a1ij.im};
        if (i == j) a1ij = cadd(a1ij, dl[i]);
        const cplx v = csub(cmul(a1ij, gl[j]), cmul(cmul(ri, inv1c), qj));
        out[2 * (int64_t)idx] = v.re;
        out[2 * (int64_t)idx + 1] = v.im;
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
int diag_ssm_launch(int mode, const double* p0, const double* p1, const double* p2, int64_t n, double* out,
                    hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(diag_ssm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mode, p0, p1,
                       p2, n, out);
    EGS_LAUNCH_CHECK();
    return 0;
}

static int skinny_launch(bool norm, const float* x, int64_t ntok, int d_model, int64_t ldx, const float* w,
                         int nh, Mamba2Params mp, NormParams np, hipStream_t stream) {
    if (ntok == 0) return 0;
    const bool vec4 = (d_model % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    int tile = 8192 / nh;                      // z tile: <= 32 KB of LDS
    if (tile > 1024) tile = 1024;
    if (tile < 32) tile = 32;
    const size_t lds = ((size_t)nh * d_model + (size_t)(tile + 1) * nh) * sizeof(float);
    if (lds > 160 * 1024) {
        set_error("projection slice too large for LDS: nheads*D = %d*%d", nh, d_model);
        return -3;
    }
    int64_t ntiles = (ntok + tile - 1) / tile;
    unsigned grid = (unsigned)(ntiles < 2048 ? ntiles : 2048);
    auto launch = [&](auto kern) -> int {
        if (lds > 48 * 1024)
            EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kProjThreads), lds, stream, x, ntok, d_model, ldx, w, nh, tile,
                           mp, np);
        EGS_LAUNCH_CHECK();
        return 0;
    };
    if (norm) return vec4 ? launch(skinny_proj_kernel<4, true>) : launch(skinny_proj_kernel<1, true>);
    return vec4 ? launch(skinny_proj_kernel<4, false>) : launch(skinny_proj_kernel<1, false>);
}

template <int G, int C>
static int mamba2_fast_dispatch(int nh_pad, const float* x, int64_t ntok, int64_t ldx, const float* w, int nh,
                                const float* b_dt, const float* dt_bias, const float* a_log, float* out,
                                hipStream_t stream) {
    constexpr int tok_per_block = 4 * (64 / G) * 4;
    int64_t blocks = (ntok + tok_per_block - 1) / tok_per_block;
    const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
    switch (nh_pad) {
        case 1: hipLaunchKernelGGL((mamba2_lambda_fast_kernel<G, C, 1>), dim3(grid), dim3(256), 0, stream, x, ntok, ldx, w, nh, b_dt, dt_bias, a_log, out); break;
        case 2: hipLaunchKernelGGL((mamba2_lambda_fast_kernel<G, C, 2>), dim3(grid), dim3(256), 0, stream, x, ntok, ldx, w, nh, b_dt, dt_bias, a_log, out); break;
        case 4: hipLaunchKernelGGL((mamba2_lambda_fast_kernel<G, C, 4>), dim3(grid), dim3(256), 0, stream, x, ntok, ldx, w, nh, b_dt, dt_bias, a_log, out); break;
        default: hipLaunchKernelGGL((mamba2_lambda_fast_kernel<G, C, 8>), dim3(grid), dim3(256), 0, stream, x, ntok, ldx, w, nh, b_dt, dt_bias, a_log, out); break;
    }
    EGS_LAUNCH_CHECK();
    return 0;
}

int mamba2_lambda_launch(const float* x, int64_t ntok, int d_model, int64_t ldx, const float* w_dt,
                         const float* b_dt, const float* dt_bias, const float* a_log, int nheads, float* out,
                         hipStream_t stream) {
    if (ntok == 0) return 0;
    const bool aligned = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(w_dt) & 15) == 0);
    if (aligned && nheads <= 8) {
        int nh_pad = 1;
        while (nh_pad < nheads) nh_pad <<= 1;
        switch (d_model) {
            case 64: return mamba2_fast_dispatch<16, 1>(nh_pad, x, ntok, ldx, w_dt, nheads, b_dt, dt_bias, a_log, out, stream);
            case 128: return mamba2_fast_dispatch<32, 1>(nh_pad, x, ntok, ldx, w_dt, nheads, b_dt, dt_bias, a_log, out, stream);
            case 256: return mamba2_fast_dispatch<64, 1>(nh_pad, x, ntok, ldx, w_dt, nheads, b_dt, dt_bias, a_log, out, stream);
            case 512: return mamba2_fast_dispatch<64, 2>(nh_pad, x, ntok, ldx, w_dt, nheads, b_dt, dt_bias, a_log, out, stream);
            case 1024: return mamba2_fast_dispatch<64, 4>(nh_pad, x, ntok, ldx, w_dt, nheads, b_dt, dt_bias, a_log, out, stream);
            default: break;
        }
    }
    Mamba2Params mp{b_dt, dt_bias, a_log, out};
    NormParams np{nullptr, nullptr, 0, 1, nullptr};
    return skinny_launch(false, x, ntok, d_model, ldx, w_dt, nheads, mp, np, stream);
}

template <int G, int C>
static int normattn_fast_dispatch(int nh_pad, const float* x, int64_t ntok, int64_t ldx, const float* w, int nh,
                                  const float* b_n, const float* offset, int norm_fn, int64_t seqlen, double* eta,
                                  hipStream_t stream) {
    constexpr int tok_per_block = 4 * (64 / G) * 4;
    int64_t blocks = (ntok + tok_per_block - 1) / tok_per_block;
    const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
    switch (nh_pad) {
        case 1: hipLaunchKernelGGL((normattn_fast_kernel<G, C, 1>), dim3(grid), dim3(256), 0, stream, x, ntok, ldx, w, nh, b_n, offset, norm_fn, seqlen, eta); break;
        case 2: hipLaunchKernelGGL((normattn_fast_kernel<G, C, 2>), dim3(grid), dim3(256), 0, stream, x, ntok, ldx, w, nh, b_n, offset, norm_fn, seqlen, eta); break;
        case 4: hipLaunchKernelGGL((normattn_fast_kernel<G, C, 4>), dim3(grid), dim3(256), 0, stream, x, ntok, ldx, w, nh, b_n, offset, norm_fn, seqlen, eta); break;
        default: hipLaunchKernelGGL((normattn_fast_kernel<G, C, 8>), dim3(grid), dim3(256), 0, stream, x, ntok, ldx, w, nh, b_n, offset, norm_fn, seqlen, eta); break;
    }
    EGS_LAUNCH_CHECK();
    return 0;
}

int normattn_eta_launch(const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_n,
                        const float* b_n, const float* offset, int norm_fn, int num_heads, double* eta,
                        hipStream_t stream) {
    const int64_t ntok = batch * seqlen;
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(w_n) & 15) == 0);
    if (aligned && num_heads <= 8 && ntok >= 2) {
        int nh_pad = 1;
        while (nh_pad < num_heads) nh_pad <<= 1;
        // T = TPW*U must be at least 2 for the overlapped iteration scheme: true for every (G, NH <= 8) below
        switch (d_model) {
            case 64: return normattn_fast_dispatch<16, 1>(nh_pad, x, ntok, d_model, w_n, num_heads, b_n, offset, norm_fn, seqlen, eta, stream);
            case 128: return normattn_fast_dispatch<32, 1>(nh_pad, x, ntok, d_model, w_n, num_heads, b_n, offset, norm_fn, seqlen, eta, stream);
            case 256: return normattn_fast_dispatch<64, 1>(nh_pad, x, ntok, d_model, w_n, num_heads, b_n, offset, norm_fn, seqlen, eta, stream);
            case 512: return normattn_fast_dispatch<64, 2>(nh_pad, x, ntok, d_model, w_n, num_heads, b_n, offset, norm_fn, seqlen, eta, stream);
            case 1024: return normattn_fast_dispatch<64, 4>(nh_pad, x, ntok, d_model, w_n, num_heads, b_n, offset, norm_fn, seqlen, eta, stream);
            default: break;
        }
    }
    Mamba2Params mp{nullptr, nullptr, nullptr, nullptr};
    NormParams np{b_n, offset, norm_fn, seqlen, eta};
    return skinny_launch(true, x, ntok, d_model, d_model, w_n, num_heads, mp, np, stream);
}

int mamba2_lti_launch(const float* a_param, const float* beta, int nheads, int64_t ntok, float* out,
                      hipStream_t stream) {
    if (ntok == 0) return 0;
    const int64_t total = ntok * nheads;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(mamba2_lti_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream,
                       a_param, beta, nheads, ntok, out);
    EGS_LAUNCH_CHECK();
    return 0;
}

int s4_dplr_launch(const double* lam_re, const double* lam_im, const double* p_re, const double* p_im,
                   const double* q_re, const double* q_im, const double* log_step, int64_t nmat, int n, int clip,
                   double* ab, hipStream_t stream) {
    if (nmat == 0 || n == 0) return 0;
    const size_t lds = (size_t)4 * n * 2 * sizeof(double);
    hipLaunchKernelGGL(s4_dplr_kernel, dim3((unsigned)nmat), dim3(256), lds, stream, lam_re, lam_im, p_re, p_im,
                       q_re, q_im, log_step, n, clip, ab);
    EGS_LAUNCH_CHECK();
    return 0;
}

}  // namespace egs
