// Attention normaliser-ratio kernels (K3 linear, K4 softmax) -- analysis/eval_eig.py:43-135.
//
// The reference materialises scores[b,t,s,h] (B*L*L*H floats, copies it to the host and reduces it with
// NumPy).  Here:
//   1. qk_proj_kernel   : q,k = x Wqk^T + b (fp32, the v rows of Wqkv are never touched), optionally
//                         phi = elu+1 fused; written once to the workspace as [B*L, 2*d_qk] fp32.
//   2a. linattn_scan    : nu_t = phi(q_t) . sum_{s<=t} phi(k_s)  -- O(L), one wavefront per (b,h), fp64 state.
//   2b. softmax_nu      : flash-style streaming over s-tiles, two passes per t-tile (row max, then
//                         sum exp(fp32(score - m_t)) in fp64) so that the reference's fp32 max-subtraction
//                         (eval_eig.py:62-74) and its masking quirks Q4/Q5 are reproduced exactly.
//   3. eta_finalize     : eta_t = nu_t / nu_{t+1} [* exp(m_t - m_{t+1})], nu == 0 -> 2e-23 (linear, Q6).
#include "egs_common.h"

namespace egs {

__device__ __forceinline__ float elu1_f32(float v) { return ((v > 0.0f) ? v : expm1f(v)) + 1.0f; }

// ---------------------------------------------------------------------------------------------------
// 1. projection GEMM  out[M, N] = act(x[M, D] . W[N, D]^T + b[N])     64x64 tile, 256 threads, 4x4 / thread
// ---------------------------------------------------------------------------------------------------
constexpr int kBM = 64, kBN = 64, kBK = 16;

template <bool PHI>
__global__ __launch_bounds__(256) void qk_proj_kernel(const float* __restrict__ x, int64_t m_rows, int d_model,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      int n_cols, float* __restrict__ out) {
    __shared__ float xs[kBK][kBM + 4];
    __shared__ float wsh[kBK][kBN + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * kBM;
    const int col0 = blockIdx.y * kBN;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

    for (int k0 = 0; k0 < d_model; k0 += kBK) {
        // stage 64 x 16 slices of x and W, transposed to [k][row]; 1024 elements each, 4 per thread
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const int r = idx >> 4, kk = idx & 15;
            const int64_t gr = row0 + r;
            const int gk = k0 + kk;
            xs[kk][r] = (gr < m_rows && gk < d_model) ? x[gr * d_model + gk] : 0.0f;
            const int gc = col0 + r;
            wsh[kk][r] = (gc < n_cols && gk < d_model) ? w[(int64_t)gc * d_model + gk] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kBK; ++kk) {
            const float4 a = *reinterpret_cast<const float4*>(&xs[kk][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&wsh[kk][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t gr = row0 + ty * 4 + i;
        if (gr >= m_rows) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gc = col0 + tx * 4 + j;
            if (gc >= n_cols) continue;
            float v = acc[i][j] + (bias ? bias[gc] : 0.0f);
            if (PHI) v = elu1_f32(v);
            out[gr * n_cols + gc] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 2a. linear attention scan: one wavefront per (b, h); lane d holds the running sum of phi(k)[d] in fp64
// ---------------------------------------------------------------------------------------------------
template <int J>   // head_dim <= 64*J
__global__ __launch_bounds__(256) void linattn_scan_kernel(const float* __restrict__ qk, int64_t nseq, int64_t seqlen,
                                                           int d_qk, int num_heads, double* __restrict__ nu) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (b, h)
    if (pair >= nseq * num_heads) return;
    const int64_t b = pair / num_heads;
    const int h = (int)(pair - b * num_heads);
    const int hd = d_qk / num_heads;
    const float* base = qk + b * seqlen * 2 * d_qk + (int64_t)h * hd;
    double* nu_out = nu + pair * seqlen;                                                // [B, H, L]
    double state[J];
#pragma unroll
    for (int j = 0; j < J; ++j) state[j] = 0.0;
    double keep = 0.0;
    for (int64_t t = 0; t < seqlen; ++t) {
        const float* row = base + t * 2 * d_qk;
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int d = lane + 64 * j;
            if (d < hd) {
                state[j] += (double)row[d_qk + d];
                part = fma((double)row[d], state[j], part);
            }
        }
        const double tot = wave_sum(part);
        if ((t & 63) == lane) keep = tot;
        if ((t & 63) == 63 || t == seqlen - 1) {
            const int64_t t0 = t & ~(int64_t)63;
            if (t0 + lane <= t) nu_out[t0 + lane] = keep;                               // coalesced 512-B store
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 2b. softmax: block per (b, h, 64-row t-tile); streams 64-wide s-tiles up to the diagonal, twice
// ---------------------------------------------------------------------------------------------------
constexpr int kTT = 64, kDK = 64;

__global__ __launch_bounds__(256) void softmax_nu_kernel(const float* __restrict__ qk, int64_t seqlen, int d_qk,
                                                         int num_heads, double* __restrict__ nu,
                                                         float* __restrict__ mrow) {
    __shared__ float qs[kDK][kTT + 4];
    __shared__ float ks[kDK][kTT + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t ttile = blockIdx.x;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int hd = d_qk / num_heads;
    const int64_t t0 = ttile * kTT;
    const float* qbase = qk + b * seqlen * 2 * d_qk + (int64_t)h * hd;
    const float* kbase = qbase + d_qk;

    float m_run[4];
    double sum_run[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m_run[i] = -INFINITY;
        sum_run[i] = 0.0;
    }

    for (int pass = 0; pass < 2; ++pass) {
        for (int64_t s0 = 0; s0 <= t0 + kTT - 1 && s0 < seqlen; s0 += kTT) {
            float acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
            for (int d0 = 0; d0 < hd; d0 += kDK) {
                __syncthreads();
                // stage q[t0.., d0..] and k[s0.., d0..] transposed to [d][row]: 4096 elements each, 16 / thread
                for (int e = 0; e < 16; ++e) {
                    const int idx = tid + e * 256;
                    const int r = idx >> 6, dd = idx & 63;
                    const bool dok = (d0 + dd) < hd;
                    const int64_t tq = t0 + r, sk = s0 + r;
                    qs[dd][r] = (dok && tq < seqlen) ? qbase[tq * 2 * d_qk + d0 + dd] : 0.0f;
                    ks[dd][r] = (dok && sk < seqlen) ? kbase[sk * 2 * d_qk + d0 + dd] : 0.0f;
                }
                __syncthreads();
#pragma unroll 8
                for (int dd = 0; dd < kDK; ++dd) {
                    const float4 a = *reinterpret_cast<const float4*>(&qs[dd][ty * 4]);
                    const float4 c = *reinterpret_cast<const float4*>(&ks[dd][tx * 4]);
                    const float av[4] = {a.x, a.y, a.z, a.w};
                    const float cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], cv[j], acc[i][j]);
                }
            }
            // consume the 4x4 score block: rows t = t0 + ty*4 + i, columns s = s0 + tx*4 + j, valid iff s <= t
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t t = t0 + ty * 4 + i;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t s = s0 + tx * 4 + j;
                    if (s <= t && t < seqlen) {
                        if (pass == 0) {
                            m_run[i] = fmaxf(m_run[i], acc[i][j]);
                        } else {
                            const float diff = __fsub_rn(acc[i][j], m_run[i]);   // fp32, eval_eig.py:72
                            const double e = exp((double)diff);                  // :74-77
                            sum_run[i] += (e == e) ? e : 0.0;                    // nan_to_num
                        }
                    }
                }
            }
        }
        if (pass == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = m_run[i];
                for (int off = 8; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
                const int64_t t = t0 + ty * 4 + i;
                // zero-masked entries s > t take part in the row max (quirk Q5): present unless t == L-1
                if (t < seqlen - 1) v = fmaxf(v, 0.0f);
                m_run[i] = v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double v = sum_run[i];
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        const int64_t t = t0 + ty * 4 + i;
        if (tx == 0 && t < seqlen) {
            const int64_t o = (b * num_heads + h) * seqlen + t;
            nu[o] = v + (double)(seqlen - 1 - t);       // masked entries contribute exp(0) = 1 each (Q4)
            mrow[o] = m_run[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 3. eta[b, t, h] from nu[b, h, t] (and m[b, h, t] for softmax)
// ---------------------------------------------------------------------------------------------------
__global__ void eta_finalize_kernel(const double* __restrict__ nu, const float* __restrict__ mrow, int64_t nseq,
                                    int64_t seqlen, int num_heads, int softmax, double* __restrict__ eta) {
    const int64_t total = nseq * (seqlen - 1) * num_heads;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(idx % num_heads);
        const int64_t bt = idx / num_heads;
        const int64_t t = bt % (seqlen - 1);
        const int64_t b = bt / (seqlen - 1);
        const int64_t o = (b * num_heads + h) * seqlen + t;
        double n0 = nu[o], n1 = nu[o + 1];
        double r;
        if (softmax) {
            const float dm = __fadd_rn(-mrow[o + 1], mrow[o]);     // fp32 difference, eval_eig.py:87
            r = (n0 / n1) * exp((double)dm);                      // :83, :88-90
        } else {
            if (n0 == 0.0) n0 = 2e-23;                            // :127
            if (n1 == 0.0) n1 = 2e-23;
            r = n0 / n1;                                          // :130
        }
        eta[idx] = r;
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static int64_t attn_chunk(int64_t batch, int64_t seqlen, int d_qk) {
    const int64_t per_seq = seqlen * 2 * (int64_t)d_qk * 4;
    int64_t chunk = ((int64_t)1 << 31) / (per_seq > 0 ? per_seq : 1);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if (chunk > 32768) chunk = 32768;          // softmax grid.z and the nu/m scratch stay bounded
    return chunk;
}

size_t attn_workspace_bytes(int64_t batch, int64_t seqlen, int d_qk, int num_heads) {
    const int64_t chunk = attn_chunk(batch, seqlen, d_qk);
    size_t qk = (size_t)chunk * seqlen * 2 * d_qk * sizeof(float);
    size_t nu = (size_t)chunk * num_heads * seqlen * sizeof(double);
    size_t m = (size_t)chunk * num_heads * seqlen * sizeof(float);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(qk) + up(nu) + up(m);
}

int attn_eta_launch(bool softmax, const float* x, int64_t batch, int64_t seqlen, int d_model, const float* w_qk,
                    const float* b_qk, int d_qk, int num_heads, double* eta, void* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
    if (batch == 0 || seqlen <= 1) return 0;
    const int hd = d_qk / num_heads;
    if (!softmax && hd > 512) {
        set_error("linear-attention scan supports head_dim <= 512 (got %d)", hd);
        return -7;
    }
    if (workspace_bytes < attn_workspace_bytes(batch, seqlen, d_qk, num_heads)) {
        set_error("attention workspace too small: %zu < %zu", workspace_bytes,
                  attn_workspace_bytes(batch, seqlen, d_qk, num_heads));
        return -11;
    }
    const int64_t chunk = attn_chunk(batch, seqlen, d_qk);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char* wsp = static_cast<char*>(workspace);
    float* qk = reinterpret_cast<float*>(wsp);
    double* nu = reinterpret_cast<double*>(wsp + up((size_t)chunk * seqlen * 2 * d_qk * sizeof(float)));
    float* mrow = reinterpret_cast<float*>(reinterpret_cast<char*>(nu) +
                                           up((size_t)chunk * num_heads * seqlen * sizeof(double)));
    const int n_cols = 2 * d_qk;
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = (batch - b0 < chunk) ? (batch - b0) : chunk;
        const int64_t rows = nb * seqlen;
        const dim3 pgrid((unsigned)((rows + kBM - 1) / kBM), (unsigned)((n_cols + kBN - 1) / kBN));
        const float* xin = x + b0 * seqlen * d_model;
        if (softmax)
            hipLaunchKernelGGL(qk_proj_kernel<false>, pgrid, dim3(256), 0, stream, xin, rows, d_model, w_qk, b_qk,
                               n_cols, qk);
        else
            hipLaunchKernelGGL(qk_proj_kernel<true>, pgrid, dim3(256), 0, stream, xin, rows, d_model, w_qk, b_qk,
                               n_cols, qk);
        EGS_LAUNCH_CHECK();
        if (softmax) {
            const dim3 sgrid((unsigned)((seqlen + kTT - 1) / kTT), (unsigned)num_heads, (unsigned)nb);
            hipLaunchKernelGGL(softmax_nu_kernel, sgrid, dim3(256), 0, stream, qk, seqlen, d_qk, num_heads, nu, mrow);
        } else {
            const int64_t pairs = nb * num_heads;
            const unsigned grid = (unsigned)((pairs + 3) / 4);
            if (hd <= 64)
                hipLaunchKernelGGL(linattn_scan_kernel<1>, dim3(grid), dim3(256), 0, stream, qk, nb, seqlen, d_qk,
                                   num_heads, nu);
            else if (hd <= 128)
                hipLaunchKernelGGL(linattn_scan_kernel<2>, dim3(grid), dim3(256), 0, stream, qk, nb, seqlen, d_qk,
                                   num_heads, nu);
            else if (hd <= 256)
                hipLaunchKernelGGL(linattn_scan_kernel<4>, dim3(grid), dim3(256), 0, stream, qk, nb, seqlen, d_qk,
                                   num_heads, nu);
            else
                hipLaunchKernelGGL(linattn_scan_kernel<8>, dim3(grid), dim3(256), 0, stream, qk, nb, seqlen, d_qk,
                                   num_heads, nu);
        }
        EGS_LAUNCH_CHECK();
        const int64_t total = nb * (seqlen - 1) * num_heads;
        const int64_t fb = (total + 255) / 256;
        hipLaunchKernelGGL(eta_finalize_kernel, dim3((unsigned)(fb < 8192 ? fb : 8192)), dim3(256), 0, stream, nu, mrow,
                           nb, seqlen, num_heads, softmax ? 1 : 0, eta + b0 * (seqlen - 1) * num_heads);
        EGS_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace egs
