// Attention normaliser-ratio kernels (K3 linear, K4 softmax) -- analysis/eval_eig.py:43-135.
//
// The reference materialises scores[b,t,s,h] (B*L*L*H floats, copies it to the host and reduces it with
// NumPy).  Here:
//   1. qk_proj_kernel   : q,k = x Wqk^T + b (fp32, the v rows of Wqkv are never touched), optionally
//                         phi = elu+1 fused; written once to the workspace as [B*L, 2*d_qk] fp32.
//   2a. linattn_scan    : nu_t = phi(q_t) . sum_{s<=t} phi(k_s)  -- O(L), one wavefront per (b,h), fp64 state.
//   2b. softmax_nu      : flash-style streaming over s-tiles.  The matrix-core kernel (softmax_nu_mfma_kernel) makes ONE
//                         pass: running row maximum (initialised with quirk Q5's zero), subtracted in fp32 like
//                         eval_eig.py:62-72; exp in fp32 with an exact range reduction (1.5e-7); fp64 row sums, rescaled
//                         in fp64 when the maximum rises; masking quirks Q4 / Q5 in closed form.  Within the reference's
//                         own fp32 error, NOT bit-faithful to its rounding (include/eigspec.h states the contract; NaN
//                         and +-inf scores are outside it: fmaxf drops NaN where torch.max propagates it).  The VALU
//                         twin (softmax_nu_kernel, head_dim % 4 != 0 or > 256) keeps the two-pass form: row maxima
//                         first, then sum exp(fp32(score - m_t)) in fp64.
//   3. eta_finalize     : eta_t = nu_t / nu_{t+1} [* exp(m_t - m_{t+1})], nu == 0 -> 2e-23 (linear, Q6).
#include <type_traits>
#include <stdlib.h>

#include "egs_common.h"

namespace egs {

// phi(v) = elu(v) + 1 (analysis/eval_eig.py:109-110); for v <= 0 that is exp(v)
__device__ __forceinline__ float elu1_f32(float v) { return (v > 0.0f) ? v + 1.0f : __expf(v); }

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

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// 1b. the same projection on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32: bit-for-bit a k-ordered fmaf chain,
// MI355X_MICROARCH "FP32-input MFMA").  This is the one genuinely dense contraction of the attention path (AI = d_qk
// flop/B, above the fp32 ridge), so it is the place where MFMA is the right tool.  Operand / accumulator lane maps
// (cdna_hip_programming.md section 3):  A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D: col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// Block tile 128 rows x 256 columns (= all q,k columns when d_qk = 128, so x is streamed from HBM exactly once),
// waves 2 x 2 with 64 x 128 each = 2 x 4 MFMA tiles (128 accumulator VGPRs, 6 LDS operand reads per 8 MFMAs); the
// next K stage is prefetched into registers while the current one is multiplied.  W is consumed pre-transposed (Wt[D][N], a 128 KB one-off transpose into the workspace)
// so that a K-stage of it is copied with conflict-free ds_write_b128; the x stage keeps its row-major shape with an odd
// leading dimension (33), which makes the A-fragment read (lane = row) conflict-free as well.
// ---------------------------------------------------------------------------------------------------
__global__ void wt_transpose_kernel(const float* __restrict__ w, int n_cols, int d_model, float* __restrict__ wt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // over wt[d][n]
    if (idx >= n_cols * d_model) return;
    const int d = idx / n_cols, nn = idx - d * n_cols;
    wt[idx] = w[(int64_t)nn * d_model + d];
}

constexpr int kM2R = 128, kM2C = 256, kM2K = 32, kM2LdA = 33, kM2LdB = kM2C + 4;
constexpr int kPatchLd = 36;   // 32 x 32 epilogue patch per wave, rows 16-B aligned

// When `csum` is given (linear attention, seq_len a multiple of 32) the epilogue also accumulates, in fp64, the column
// sums of phi(k) over each 32-row accumulator tile into csum[(b, h, chunk), d] -- the per-chunk carry-in sums the scan
// needs -- so k is not read a second time for them (one 8-byte atomic per column and tile).
struct ChunkSumArgs {
    double* csum;          // [nseq * H * nchunks, hd] or null
    int64_t seqlen;
    int d_qk, num_heads, nchunks, chunk_tokens;
};

// WN = waves along the columns.  WN = 2: block tile 128 rows x 256 columns (2 x 2 waves, each 64 x 128) -- the form
// rounds 1-3 had for every shape.  WN = 1: 256 rows x 128 columns (4 x 1 waves, each 64 x 128) for projections with at
// most 128 columns (2 d_qk <= 128, e.g. the MQAR configs' d_qk = 64): in the 256-wide tile their second column half was
// all padding, i.e. half of the matrix-core work and of the W staging computed zeros (round 4: 1.35 -> 0.7 ms on the
// MQAR batch).  Same per-wave code either way; only the staging counts change (x: R / 32 float4 per thread, W: C / 32).
template <bool PHI, int WN>
__global__ __launch_bounds__(256, 2) void qk_proj_mfma2_kernel(const float* __restrict__ x, int64_t m_rows, int d_model,
                                                               const float* __restrict__ wt, const float* __restrict__ bias,
                                                               int n_cols, float* __restrict__ out, ChunkSumArgs cs) {
    constexpr int R = 64 * (4 / WN), C = 128 * WN, LdB = C + 4;
    constexpr int NX = R / 32, NW = C / 32;                           // staged float4 per thread and k-stage
    static_assert((WN == 1 ? R * kM2LdA : kM2K * LdB) >= 4 * 32 * kPatchLd, "an operand stage doubles as the epilogue patches");
    __shared__ __attribute__((aligned(16))) float as[R * kM2LdA];
    __shared__ __attribute__((aligned(16))) float bs[kM2K * LdB];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kh = lane >> 5, l31 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    const int col0 = blockIdx.y * C;
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // x stage: thread -> rows (tid>>3) + 32 e, k-quad (tid&7);  Wt stage: thread -> k rows (tid / (C/4)) + (1024/C) e,
    // column quad (tid % (C/4))
    float4 xa[NX], wb[NW];
    unsigned okm = 0u;                 // validity of the staged loads (bits 0..NX-1: x, 8..8+NW-1: W)
    const int xr = tid >> 3, xq = tid & 7;
    constexpr int WQ = C / 4, WKS = 256 / WQ;                         // column quads per k row, k rows per pass
    const int wk = tid / WQ, wq = tid % WQ;
    // Loads always read a valid address and are zeroed when they are STORED to LDS one barrier later: a select right
    // after the load makes the wave wait for the data at once, i.e. turns the prefetch into a synchronous load.
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const int64_t gr = row0 + xr + 32 * e;
            const int gk = k0 + xq * 4;
            const bool ok = gr < m_rows && gk < d_model;
            xa[e] = *reinterpret_cast<const float4*>(x + (ok ? gr * d_model + gk : 0));
            okm = ok ? (okm | (1u << e)) : (okm & ~(1u << e));
        }
#pragma unroll
        for (int e = 0; e < NW; ++e) {
            const int gk = k0 + wk + WKS * e;
            const int gc = col0 + wq * 4;
            const bool ok = gk < d_model && gc < n_cols;
            wb[e] = *reinterpret_cast<const float4*>(wt + (ok ? (int64_t)gk * n_cols + gc : 0));
            okm = ok ? (okm | (256u << e)) : (okm & ~(256u << e));
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const bool ok = (okm >> e) & 1u;
            float* p = &as[(xr + 32 * e) * kM2LdA + xq * 4];
            p[0] = ok ? xa[e].x : 0.0f;
            p[1] = ok ? xa[e].y : 0.0f;
            p[2] = ok ? xa[e].z : 0.0f;
            p[3] = ok ? xa[e].w : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < NW; ++e)
            *reinterpret_cast<float4*>(&bs[(wk + WKS * e) * LdB + wq * 4]) =
                ((okm >> (8 + e)) & 1u) ? wb[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    stage_load(0);
    for (int k0 = 0; k0 < d_model; k0 += kM2K) {
        __syncthreads();
        stage_store();
        __syncthreads();
        if (k0 + kM2K < d_model) stage_load(k0 + kM2K);
        // LDS operands of two k-pairs (16 MFMAs) fetched while the previous two's MFMAs run: left alone, the scheduler sinks
        // each load to just before its consumer and every MFMA group waits an LDS round trip
        float a0[2][2], a1[2][2], bq[2][2][4];
        auto fetch = [&](int buf, int kb) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kk = 2 * (kb + u) + kh;
                a0[buf][u] = as[(wm * 64 + l31) * kM2LdA + kk];
                a1[buf][u] = as[(wm * 64 + 32 + l31) * kM2LdA + kk];
#pragma unroll
                for (int c = 0; c < 4; ++c) bq[buf][u][c] = bs[kk * LdB + wn * 128 + 32 * c + l31];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < kM2K / 4; ++g) {
            if (g + 1 < kM2K / 4) fetch((g + 1) & 1, 2 * (g + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[g & 1][u], bq[g & 1][u][c], acc[0][c], 0, 0, 0);
                    acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[g & 1][u], bq[g & 1][u][c], acc[1][c], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // Epilogue: bias + phi, then each 32x32 accumulator tile is transposed through a per-wave LDS patch so that the
    // global stores are 16 B per lane (8 lanes cover one 128-B row segment) instead of 16 dword stores per tile.
    __syncthreads();                                   // every wave is done with the operand stages
    float* patch = (WN == 1 ? as : bs) + wave * (32 * kPatchLd);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int gc = col0 + wn * 128 + tn * 32 + l31;
            const float bv = (bias && gc < n_cols) ? bias[gc] : 0.0f;
            double colsum = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[tm][tn][r] + bv;
                if (PHI) v = (v > 0.0f) ? v + 1.0f : __expf(v);     // elu(v) + 1 == exp(v) for v <= 0
                patch[((r & 3) + 8 * (r >> 2) + 4 * kh) * kPatchLd + l31] = v;
                colsum += (double)v;
            }
            if (PHI && cs.csum != nullptr) {
                colsum += __shfl_xor(colsum, 32, 64);              // the other 16 rows of this column
                const int64_t grow = row0 + wm * 64 + tm * 32;      // first of the tile's 32 rows (one sequence, one chunk)
                if (kh == 0 && gc >= cs.d_qk && gc < n_cols && grow < m_rows) {
                    const int hd = cs.d_qk / cs.num_heads;
                    const int kc = gc - cs.d_qk;
                    const int64_t bseq = grow / cs.seqlen;
                    const int chunk = (int)((grow - bseq * cs.seqlen) / cs.chunk_tokens);
                    const int64_t task = (bseq * cs.num_heads + kc / hd) * cs.nchunks + chunk;
                    atomicAdd(&cs.csum[task * hd + kc % hd], colsum);
                }
            }
            wave_lds_sync();
            const int pr = lane >> 3, pc = (lane & 7) * 4;
            const int gcol = col0 + wn * 128 + tn * 32 + pc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = pr + 8 * q;
                const float4 v4 = *reinterpret_cast<const float4*>(&patch[rr * kPatchLd + pc]);
                const int64_t gr = row0 + wm * 64 + tm * 32 + rr;
                if (gr < m_rows && gcol < n_cols) *reinterpret_cast<float4*>(out + gr * n_cols + gcol) = v4;
            }
            wave_lds_sync();
        }
}

// ---------------------------------------------------------------------------------------------------
// 2a. linear attention scan: one wavefront per (b, h); lane d holds the running sum of phi(k)[d] in fp64
// ---------------------------------------------------------------------------------------------------
// Transpose-reduce over the 64 lanes (see stream_kernels.hip): CNT values per lane in, lane l ends with the complete
// sum of value index l (for CNT = 64).
template <typename T, int CNT, int OFF>
__device__ __forceinline__ void treduce_t(T* v, int lane) {
    if constexpr (OFF >= 1) {
        const bool upper = (lane & OFF) != 0;
        if constexpr (CNT > 1) {
            constexpr int H = CNT / 2;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const T send = upper ? v[i] : v[i + H];
                const T keep = upper ? v[i + H] : v[i];
                v[i] = keep + __shfl_xor(send, OFF, 64);
            }
            treduce_t<T, H, OFF / 2>(v, lane);
        } else {
            v[0] += __shfl_xor(v[0], OFF, 64);
            treduce_t<T, 1, OFF / 2>(v, lane);
        }
    }
}

// One wavefront per (b, h); lane d carries the fp64 running sum of phi(k)[d].  Tokens are processed 16 at a time:
// the 16 per-lane products q_t[d] * S_t[d] are combined across the 64 lanes with one transpose-reduce (15 + 2 exchanges
// instead of 16 x 6), after which 4-lane groups hold nu for one token each and the store is coalesced.
// The sequence is cut into chunks of kScanChunk tokens so that B*H*L/kScanChunk wavefronts run concurrently:
// linattn_chunksum_kernel produces each chunk's own sum of phi(k) (fp64), the scan kernel adds the sums of the preceding
// chunks as its carry-in.
constexpr int kScanChunk = 256;

// RAW: qk holds raw projections (the (q, k)-input entry point): phi = elu + 1 is applied on load
__device__ __forceinline__ float phi_elu1(float v) { return (v > 0.0f) ? v + 1.0f : __expf(v); }

template <int J, bool RAW = false>
__global__ __launch_bounds__(256) void linattn_chunksum_kernel(const float* __restrict__ qk, int64_t nseq, int64_t seqlen,
                                                               int d_qk, int num_heads, int nchunks,
                                                               double* __restrict__ csum) {
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (b, h, chunk)
    if (task >= nseq * num_heads * nchunks) return;
    const int64_t pair = task / nchunks;
    const int chunk = (int)(task - pair * nchunks);
    const int64_t b = pair / num_heads;
    const int h = (int)(pair - b * num_heads);
    const int hd = d_qk / num_heads;
    const float* base = qk + b * seqlen * 2 * d_qk + d_qk + (int64_t)h * hd;            // k rows
    const int64_t t_begin = (int64_t)chunk * kScanChunk;
    const int64_t t_end = (t_begin + kScanChunk < seqlen) ? t_begin + kScanChunk : seqlen;
    double acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = 0.0;
    for (int64_t t = t_begin; t < t_end; t += 8) {
        float kv[8][J];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t tt = (t + u < t_end) ? t + u : t_end - 1;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int d = lane + 64 * j;
                float kk = base[tt * 2 * d_qk + (d < hd ? d : hd - 1)];            // always-valid address
                if (RAW) kk = phi_elu1(kk);
                kv[u][j] = (d < hd && t + u < t_end) ? kk : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < J; ++j) acc[j] += (double)kv[u][j];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = lane + 64 * j;
        if (d < hd) csum[task * hd + d] = acc[j];
    }
}

template <int J, bool RAW = false>   // head_dim <= 64*J
__global__ __launch_bounds__(256) void linattn_scan_kernel(const float* __restrict__ qk, int64_t nseq, int64_t seqlen,
                                                           int d_qk, int num_heads, int nchunks,
                                                           const double* __restrict__ csum, double* __restrict__ nu) {
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // (b, h, chunk)
    if (task >= nseq * num_heads * nchunks) return;
    const int64_t pair = task / nchunks;
    const int chunk = (int)(task - pair * nchunks);
    const int64_t b = pair / num_heads;
    const int h = (int)(pair - b * num_heads);
    const int hd = d_qk / num_heads;
    const float* base = qk + b * seqlen * 2 * d_qk + (int64_t)h * hd;
    double* nu_out = nu + pair * seqlen;                                                // [B, H, L]
    constexpr int kScanT = (J <= 2) ? 16 : (J == 4 ? 8 : 4);   // tokens per round (register budget)
    constexpr int kGrp = 64 / kScanT;                         // lanes that end up sharing one token's sum
    const int64_t t_begin = (int64_t)chunk * kScanChunk;
    const int64_t t_end = (t_begin + kScanChunk < seqlen) ? t_begin + kScanChunk : seqlen;
    double state[J];
#pragma unroll
    for (int j = 0; j < J; ++j) state[j] = 0.0;
    for (int c = 0; c < chunk; ++c) {                       // carry-in: sums of the preceding chunks
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int d = lane + 64 * j;
            if (d < hd) state[j] += csum[(pair * nchunks + c) * hd + d];
        }
    }
    for (int64_t t0 = t_begin; t0 < t_end; t0 += kScanT) {
        // all 2*J*kScanT loads of the round are issued back to back from always-valid addresses (clamped token and
        // lane); out-of-range lanes / tokens are zeroed by value afterwards, so nothing serialises on a branch
        float qv[kScanT][J], kv[kScanT][J];
#pragma unroll
        for (int u = 0; u < kScanT; ++u) {
            const int64_t t = (t0 + u < t_end) ? (t0 + u) : (t_end - 1);
            const float* row = base + t * 2 * d_qk;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int d = lane + 64 * j;
                const int dc = (d < hd) ? d : hd - 1;
                qv[u][j] = RAW ? phi_elu1(row[dc]) : row[dc];
                kv[u][j] = RAW ? phi_elu1(row[d_qk + dc]) : row[d_qk + dc];
            }
        }
#pragma unroll
        for (int u = 0; u < kScanT; ++u)
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (lane + 64 * j >= hd) {
                    qv[u][j] = 0.0f;
                    kv[u][j] = 0.0f;
                }
        double prod[kScanT];
#pragma unroll
        for (int u = 0; u < kScanT; ++u) {
            double part = 0.0;
            if (t0 + u < t_end) {
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    state[j] += (double)kv[u][j];
                    part = fma((double)qv[u][j], state[j], part);
                }
            }
            prod[u] = part;
        }
        treduce_t<double, kScanT, 32>(prod, lane);
        // value index = lane / kGrp (top bits of the lane), complete in all kGrp lanes of the group
        const int64_t t = t0 + lane / kGrp;
        if ((lane % kGrp) == 0 && t < t_end) nu_out[t] = prod[0];
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
                {
                    // 32 global loads per thread issued back to back from always-valid (clamped) addresses, then
                    // zeroed by value and written to LDS -- a per-element branch would serialise them
                    float qv[16], kv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int idx = tid + e * 256;
                        const int r = idx >> 6, dd = idx & 63;
                        const int dc = (d0 + dd < hd) ? d0 + dd : hd - 1;
                        const int64_t tq = (t0 + r < seqlen) ? t0 + r : seqlen - 1;
                        const int64_t sk = (s0 + r < seqlen) ? s0 + r : seqlen - 1;
                        qv[e] = qbase[tq * 2 * d_qk + dc];
                        kv[e] = kbase[sk * 2 * d_qk + dc];
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int idx = tid + e * 256;
                        const int r = idx >> 6, dd = idx & 63;
                        const bool dok = (d0 + dd) < hd;
                        qs[dd][r] = (dok && t0 + r < seqlen) ? qv[e] : 0.0f;
                        ks[dd][r] = (dok && s0 + r < seqlen) ? kv[e] : 0.0f;
                    }
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
// 2c. the same two-pass row statistics with the score tiles on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32
//     products and sums).  Workgroup = 4 waves = 64 query rows (16 per wave) of one (b, h); key blocks of 64 up to
//     the diagonal, staged once per block in LDS for all four waves.  Each wave computes TRANSPOSED score tiles
//     D[s][t] = K_tile . Q_tile^T: lane l then owns query row t = l % 16 for every tile, so the row maximum / row sum
//     are in-lane reductions over the accumulator registers plus one 2-step combine of the four lane groups at the end.
//       A operand (16 x 4):  lane l -> K[s0 + l % 16][4 j + l / 16]   (LDS, layout [d % 4][s][d / 4]: one 16-byte read
//                                                                      feeds 4 MFMAs, conflict-free with a 4-float pad)
//       B operand (4 x 16):  lane l -> Q[t0 + l % 16][4 j + l / 16]   (J registers, loaded once per workgroup)
//       D (16 x 16, 4 regs): lane l, reg r -> score(s = s0 + 4 (l / 16) + r, t = t0 + l % 16)
// ---------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kNuRows = 64;

// exp(x) in fp64 for the fp32 differences of eval_eig.py:72-77: 2^n * exp(r), r = x - n ln2 in [-0.35, 0.35], degree-11
// Taylor polynomial (truncation 4e-15 relative) -- a quarter of the instructions of the general-purpose routine, which
// matters because pass 2 evaluates it once per score element.  Arguments below -745 give 0 like exp() does.
__device__ __forceinline__ double exp_f64_fast(double x) {
    if (!(x > -745.0)) return (x == x) ? 0.0 : x;                     // underflow; NaN propagates
    const double n = rint(x * 1.4426950408889634);
    const double r = fma(n, -6.93147180369123816490e-01, x) - n * 1.90821492927058770002e-10;   // ln2 hi / lo
    double p = 2.50521083854417187751e-08;                            // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);
    p = fma(p, r, 2.75573192239858906526e-06);
    p = fma(p, r, 2.48015873015873015873e-05);
    p = fma(p, r, 1.98412698412698412698e-04);
    p = fma(p, r, 1.38888888888888888889e-03);
    p = fma(p, r, 8.33333333333333333333e-03);
    p = fma(p, r, 4.16666666666666666667e-02);
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// exp(d) for the fp32 score differences d = score - max <= 0 of one row (eval_eig.py:72-77), in fp32 with an exact
// range reduction: n = rint(d log2 e), f = d log2 e - n from two fused multiply-adds (log2 e split in hi + lo, so f is
// good to half an ulp), 2^f on the hardware unit (v_exp_f32, 1 ulp), scaled by 2^n exactly.  1.5e-7 relative, against the
// 1e-6 ... 2e-4 the reference's own fp32 scores put into these sums (tests/test_gpu_parity.py measures both against the
// float64-exact formula); the row sum itself is accumulated in fp64.  Ten fp32 instructions instead of the ~20 fp64
// ones of exp_f64_fast: the one-pass kernel below evaluates it once per causal score element.
__device__ __forceinline__ float exp_f32_split(float d) {
    d = fmaxf(d, -104.0f);                                            // 2^-150: below the smallest fp32 denormal
    const float n = rintf(d * 1.44269502f);
    float f = fmaf(d, 1.44269502f, -n);
    f = fmaf(d, 1.92596299e-8f, f);
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// One pass over the key blocks (round 4; rounds 1-3 made two: row maxima first, then the sums): every lane keeps a
// running maximum and a running fp64 sum relative to it for its row and its quarter of the keys, and multiplies the sum
// by exp(m_old - m_new) when a block raises the maximum (a wave-uniform branch: after the first blocks of a row it is
// rarely taken).  m_old - m_new is formed in fp64 from two fp32 numbers, i.e. exactly, so the rescaled sum differs from
// the two-pass one only by the rounding of that one exp.  Quirk Q5 (the zero-masked entries s > t take part in the row
// maximum, eval_eig.py:59, :62) is the INITIAL value of the running maximum: 0 for every row but the last.
// Half the matrix-core work of the two-pass form, and the epilogue is fp32 (exp_f32_split).
//
// Scheduling: a (b, h, 64-row tile) costs (tile index + 1) key blocks and lives ~25 us -- as one workgroup each, the chip
// ran at 1.7 of 3 possible waves per SIMD (launch gaps between short-lived workgroups of very different lengths).  Now a
// resident grid walks UNITS of equal cost: tile p of a sequence together with its mirror tile T-1-p (T + 1 blocks), big
// tile first; grid-stride over identical units needs no queue.
template <int J, int KB>
__global__ __launch_bounds__(256) void softmax_nu_mfma_kernel(const float* __restrict__ qk, int64_t nbh, int64_t seqlen,
                                                              int d_qk, int num_heads, double* __restrict__ nu,
                                                              float* __restrict__ mrow) {
    static_assert(KB == 64, "four 16-key tiles per block, one per wave's diagonal position");
    extern __shared__ __attribute__((aligned(16))) float ks2[];      // [4][KB][J + 4]
    constexpr int kLd = J + 4;
    constexpr int kV4 = (KB * J + 255) / 256;                         // float4 of the K block per thread
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int hd = d_qk / num_heads;                                  // multiple of 4 (checked by the launcher)
    const int ntile = (int)((seqlen + kNuRows - 1) / kNuRows);
    const int npair = (ntile + 1) / 2;
    for (int64_t unit = blockIdx.x; unit < nbh * npair; unit += gridDim.x) {
    const int64_t bh = unit / npair;
    const int pr = (int)(unit - bh * npair);
    const int h = (int)(bh % num_heads);
    const int64_t b = bh / num_heads;
    for (int half = 0; half < 2; ++half) {
    const int tile = half == 0 ? ntile - 1 - pr : pr;
    if (half == 1 && tile == ntile - 1 - pr) break;                   // the middle tile of an odd count stands alone
    const int64_t t0 = (int64_t)tile * kNuRows;
    const float* qbase = qk + b * seqlen * 2 * d_qk + (int64_t)h * hd;
    const float* kbase = qbase + d_qk;
    const int64_t t = t0 + 16 * w + c;                                // this lane's query row
    const int64_t tc = t < seqlen ? t : seqlen - 1;

    float qreg[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = 4 * j + g;
        const float v = qbase[tc * 2 * d_qk + (d < hd ? d : hd - 1)];
        qreg[j] = (d < hd && t < seqlen) ? v : 0.0f;
    }
    const int64_t s_end = (t0 + kNuRows < seqlen) ? t0 + kNuRows : seqlen;   // keys [0, s_end) are needed

    // K block of keys s0..s0+KB-1 as kV4 float4 per thread (key sr, dims 4 j..4 j+3), fetched a block ahead
    float4 kpre[kV4];
    auto fetch = [&](int64_t s0) {
#pragma unroll
        for (int x = 0; x < kV4; ++x) {
            const int idx = tid + 256 * x;
            const int sr = (idx / J) % KB, j = idx % J;
            const int64_t sk = (s0 + sr < seqlen) ? s0 + sr : seqlen - 1;
            const int dj = (4 * j < hd) ? 4 * j : hd - 4;
            kpre[x] = *reinterpret_cast<const float4*>(kbase + sk * 2 * d_qk + dj);
        }
    };
    fetch(0);

    float m_run = (t < seqlen - 1) ? 0.0f : -INFINITY;                // Q5: the masked zeros of the row
    double sum_run = 0.0;
    // NT score tiles (16 keys each) of the staged block, independent accumulator chains interleaved; then the online
    // update with them.  DIAG: tile NT-1 straddles the diagonal (keys 16 w + 4 g + r against row 16 w + c).
    auto block = [&](auto nt_tag, auto diag_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr bool DIAG = decltype(diag_tag)::value;
        f32x4 acc[NT];
#pragma unroll
        for (int x = 0; x < NT; ++x) acc[x] = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* kr = ks2 + (g * KB + c) * kLd;
#pragma unroll
        for (int jj = 0; jj < J; jj += 4) {
            float4 k4[NT];
#pragma unroll
            for (int x = 0; x < NT; ++x) k4[x] = *reinterpret_cast<const float4*>(kr + 16 * x * kLd + jj);
#pragma unroll
            for (int x = 0; x < NT; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[x].x, qreg[jj], acc[x], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < NT; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[x].y, qreg[jj + 1], acc[x], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < NT; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[x].z, qreg[jj + 2], acc[x], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < NT; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[x].w, qreg[jj + 3], acc[x], 0, 0, 0);
        }
        float sc[NT][4];
        bool ok[NT][4];
        float bm = m_run;
#pragma unroll
        for (int x = 0; x < NT; ++x) {
            sc[x][0] = acc[x].x, sc[x][1] = acc[x].y, sc[x][2] = acc[x].z, sc[x][3] = acc[x].w;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ok[x][r] = !(DIAG && x == NT - 1) || 4 * g + r <= c;
                bm = ok[x][r] ? fmaxf(bm, sc[x][r]) : bm;
            }
        }
        if (__builtin_amdgcn_ballot_w64(bm > m_run) != 0ULL) {         // some row's maximum moved: rescale (rare later on)
            if (bm > m_run) {
                sum_run *= (m_run == -INFINITY) ? 0.0 : exp_f64_fast((double)m_run - (double)bm);
                m_run = bm;
            }
        }
        // all exponentials first (independent chains the scheduler can interleave), then the fp64 accumulation;
        // branch-free: a per-element `if` serialises sixteen dependent ten-instruction chains behind sixteen branches
        float ev[NT][4];
#pragma unroll
        for (int x = 0; x < NT; ++x) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp_f32_split(__fsub_rn(sc[x][r], m_run));   // fp32 difference, eval_eig.py:72
                ev[x][r] = ok[x][r] ? fmaxf(e, 0.0f) : 0.0f;                 // maxnum drops a NaN: nan_to_num, :74-77
            }
        }
#pragma unroll
        for (int x = 0; x < NT; ++x) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sum_run += (double)ev[x][r];
        }
    };
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;
    using T4 = std::integral_constant<int, 4>;
    for (int64_t s0 = 0; s0 < s_end; s0 += KB) {
        __syncthreads();                                              // previous block fully consumed
#pragma unroll
        for (int x = 0; x < kV4; ++x) {
            const int idx = tid + 256 * x;
            const int sr = idx / J, j = idx - sr * J;
            if (idx < KB * J) {
                const bool okk = (4 * j < hd) && (s0 + sr < seqlen);
                ks2[(0 * KB + sr) * kLd + j] = okk ? kpre[x].x : 0.0f;
                ks2[(1 * KB + sr) * kLd + j] = okk ? kpre[x].y : 0.0f;
                ks2[(2 * KB + sr) * kLd + j] = okk ? kpre[x].z : 0.0f;
                ks2[(3 * KB + sr) * kLd + j] = okk ? kpre[x].w : 0.0f;
            }
        }
        __syncthreads();
        if (s0 + KB < s_end) fetch(s0 + KB);
        if (s0 < t0) {                                                // block below the diagonal: every key <= every row
            block(T4{}, std::false_type{});
        } else {                                                      // the diagonal block: tiles 0..w, tile w masked
            if (w == 0) block(T1{}, std::true_type{});
            else if (w == 1) block(T2{}, std::true_type{});
            else if (w == 2) block(T3{}, std::true_type{});
            else block(T4{}, std::true_type{});
        }
    }
    // the four lane groups of a row hold disjoint key subsets: combine (m, sum) pairs
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const float mo = __shfl_xor(m_run, off, 64);
        const double so = __shfl_xor(sum_run, off, 64);
        const float mn = fmaxf(m_run, mo);
        const double fa = (m_run == mn) ? 1.0 : ((m_run == -INFINITY) ? 0.0 : exp_f64_fast((double)m_run - (double)mn));
        const double fb = (mo == mn) ? 1.0 : ((mo == -INFINITY) ? 0.0 : exp_f64_fast((double)mo - (double)mn));
        sum_run = sum_run * fa + so * fb;
        m_run = mn;
    }
    if (g == 0 && t < seqlen) {
        const int64_t o = (b * num_heads + h) * seqlen + t;
        nu[o] = sum_run + (double)(seqlen - 1 - t);                   // masked entries contribute exp(0) = 1 each (Q4)
        mrow[o] = m_run;
    }
    }   // half
    }   // unit
}

// key block per LDS stage
template <int J>
static int softmax_nu_mfma_launch(const float* qk, int64_t nb, int64_t seqlen, int d_qk, int num_heads, double* nu,
                                  float* mrow, hipStream_t stream) {
    constexpr int KB = 64;   // measured: 128- and 256-key stages are 8-12 % slower (occupancy, loads past the diagonal)
    const size_t dyn = (size_t)4 * KB * (J + 4) * sizeof(float);
    if (dyn > 48 * 1024)
        EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_nu_mfma_kernel<J, KB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    int per_cu = 0;
    EGS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, softmax_nu_mfma_kernel<J, KB>, 256, dyn));
    if (per_cu < 1) per_cu = 1;
    const int64_t ntile = (seqlen + kNuRows - 1) / kNuRows;
    const int64_t units = nb * num_heads * ((ntile + 1) / 2);
    const int64_t resident = (int64_t)current_device_cus() * per_cu;
    const dim3 grid((unsigned)(units < resident ? units : resident));
    hipLaunchKernelGGL((softmax_nu_mfma_kernel<J, KB>), grid, dim3(256), dyn, stream, qk, nb * num_heads, seqlen, d_qk,
                       num_heads, nu, mrow);
    return 0;
}

// row statistics of the causal score matrix for nb sequences of a packed [nb, L, 2 d_qk] (q | k) buffer
static int softmax_nu_dispatch(const float* qk, int64_t nb, int64_t seqlen, int d_qk, int num_heads, double* nu, float* mrow,
                               hipStream_t stream) {
    const int hd = d_qk / num_heads;
    const int hd4 = (hd + 3) / 4;
    if (hd % 4 == 0 && d_qk % 4 == 0 && hd4 <= 64 && (reinterpret_cast<uintptr_t>(qk) & 15) == 0 &&
        !getenv("EGS_SOFTMAX_VALU")) {                                // matrix-core tiles
        if (hd4 <= 4) return softmax_nu_mfma_launch<4>(qk, nb, seqlen, d_qk, num_heads, nu, mrow, stream);
        if (hd4 <= 8) return softmax_nu_mfma_launch<8>(qk, nb, seqlen, d_qk, num_heads, nu, mrow, stream);
        if (hd4 <= 16) return softmax_nu_mfma_launch<16>(qk, nb, seqlen, d_qk, num_heads, nu, mrow, stream);
        if (hd4 <= 32) return softmax_nu_mfma_launch<32>(qk, nb, seqlen, d_qk, num_heads, nu, mrow, stream);
        return softmax_nu_mfma_launch<64>(qk, nb, seqlen, d_qk, num_heads, nu, mrow, stream);
    }
    const dim3 sgrid((unsigned)((seqlen + kTT - 1) / kTT), (unsigned)num_heads, (unsigned)nb);
    hipLaunchKernelGGL(softmax_nu_kernel, sgrid, dim3(256), 0, stream, qk, seqlen, d_qk, num_heads, nu, mrow);
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// 2c. linear attention, FUSED: projection + phi + fp64 prefix + normaliser in one kernel -- q and k never exist in HBM.
// get_eig_att_linear (analysis/eval_eig.py:97-135) needs nu_t = phi(q_t) . sum_{s<=t} phi(k_s) only; the unfused path
// writes q,k (2 d_qk floats per token) and reads them back, 5x the algorithmic bytes.  Here one workgroup walks ONE
// sequence tile by tile (128 tokens x all 2 d_qk = 256 projected columns on v_mfma_f32_32x32x2_f32, same tiling as
// qk_proj_mfma2_kernel), and after each tile's products:
//   * the waves holding the q columns and the k columns of 32 tokens put phi(q), phi(k) into LDS (the operand stages
//     are free by then),
//   * the four waves scan 8 tokens each in fp64: per-wave k sums first (one barrier), then S_t[d] = carry + earlier
//     waves + own tokens, the products phi(q_t)[d] S_t[d] and one transpose-reduce per 8 tokens (lane = head dimension);
//   * the fp64 carry S (one value per head dimension, held redundantly by every wave) moves on to the next round.
// The next tile's first x / W stage is fetched during the epilogue.  Shapes: 2 d_qk == 256, H in {1, 2}, B >= the
// number of resident workgroups that fills the chip; everything else keeps the two-kernel path.
// ---------------------------------------------------------------------------------------------------
constexpr int kFusLd = 129;           // phi(q) / phi(k) slabs [32][129] floats: rows and columns conflict-free
constexpr int kFusT = 512;            // 8 wavefronts: 2 x 4 waves of 64 x 64 (64 accumulator registers per lane)

template <int HEADS>
__global__ __launch_bounds__(kFusT, 1) void linattn_fused_kernel(const float* __restrict__ x, int64_t nseq, int64_t seqlen,
                                                                 int d_model, const float* __restrict__ wt,
                                                                 const float* __restrict__ bias, double* __restrict__ nu) {
    constexpr int n_cols = kM2C;                       // 2 d_qk == 256
    __shared__ float as[kM2R][kM2LdA];
    __shared__ __attribute__((aligned(16))) float bs[kM2K][kM2LdB];
    __shared__ double ksum[8][128];
    extern __shared__ __attribute__((aligned(16))) float fus_dyn[];   // phi(q), phi(k) of the whole 128-token tile
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;           // rows wm*64.., columns wn*64..  (wn 0,1: q   wn 2,3: k)
    const int kh = lane >> 5, l31 = lane & 31;
    float* qslab = fus_dyn;                            // [64][kFusLd]
    float* kslab = qslab + 64 * kFusLd;                // [64][kFusLd]    (one workgroup per CU: 66 KB + the stages)

    // stage loads: x stage 128 x 32 floats = 1024 float4 (2 per thread), W^T stage 32 x 256 floats = 2048 float4 (4 per thread)
    float4 xa0, xa1, wb0, wb1, wb2, wb3;
    unsigned okm = 0u;                                 // validity of the six staged loads (bits 0-1: x, 2-5: W)
    const int xr = tid >> 3, xq = tid & 7;             // rows xr + 64 e, k-quad xq
    const int wk = tid >> 6, wq = tid & 63;            // k rows wk + 8 e, column quad wq
    float bv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) bv[tn] = bias ? bias[wn * 64 + tn * 32 + l31] : 0.0f;

    for (int64_t seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
        const float* xs = x + seq * seqlen * d_model;
        // Loads always read a valid address and are zeroed when they are STORED to LDS one barrier later (the validity bits
        // travel in okm): a select right after the load would make the wave wait for the data at once, i.e. turn the
        // prefetch into a synchronous load in front of every stage's MFMAs.
#define FUS_XLOAD(XA, E, T0, K0)                                                                          \
    {                                                                                                     \
        const int64_t tk_ = (T0) + xr + 64 * (E);                                                         \
        const int gk_ = (K0) + xq * 4;                                                                    \
        const bool ok_ = tk_ < seqlen && gk_ < d_model;                                                   \
        XA = *reinterpret_cast<const float4*>(xs + (ok_ ? tk_ * d_model + gk_ : 0));                      \
        okm = ok_ ? (okm | (1u << (E))) : (okm & ~(1u << (E)));                                           \
    }
#define FUS_WLOAD(WB, E, K0)                                                                              \
    {                                                                                                     \
        const int gk_ = (K0) + wk + 8 * (E);                                                              \
        const bool ok_ = gk_ < d_model;                                                                   \
        WB = *reinterpret_cast<const float4*>(wt + (ok_ ? gk_ * n_cols + wq * 4 : 0));                    \
        okm = ok_ ? (okm | (4u << (E))) : (okm & ~(4u << (E)));                                           \
    }
#define FUS_STAGE_LOAD(T0, K0)                                                                                     \
    FUS_XLOAD(xa0, 0, T0, K0) FUS_XLOAD(xa1, 1, T0, K0)                                                            \
    FUS_WLOAD(wb0, 0, K0) FUS_WLOAD(wb1, 1, K0) FUS_WLOAD(wb2, 2, K0) FUS_WLOAD(wb3, 3, K0)
#define FUS_XSTORE(XA, E)                               \
    {                                                   \
        const bool ok_ = (okm >> (E)) & 1u;             \
        float* p_ = &as[xr + 64 * (E)][xq * 4];         \
        p_[0] = ok_ ? XA.x : 0.0f;                      \
        p_[1] = ok_ ? XA.y : 0.0f;                      \
        p_[2] = ok_ ? XA.z : 0.0f;                      \
        p_[3] = ok_ ? XA.w : 0.0f;                      \
    }
#define FUS_WSTORE(WB, E)                                                                            \
    {                                                                                                \
        const bool ok_ = (okm >> (2 + (E))) & 1u;                                                    \
        *reinterpret_cast<float4*>(&bs[wk + 8 * (E)][wq * 4]) =                                     \
            ok_ ? WB : make_float4(0.f, 0.f, 0.f, 0.f);                                              \
    }
#define FUS_STAGE_STORE()                                                                 \
    FUS_XSTORE(xa0, 0) FUS_XSTORE(xa1, 1)                                                 \
    FUS_WSTORE(wb0, 0) FUS_WSTORE(wb1, 1) FUS_WSTORE(wb2, 2) FUS_WSTORE(wb3, 3)

        // fp64 carry: S[d] for d = lane and lane + 64 (every wave keeps the same copy)
        double carry0 = 0.0, carry1 = 0.0;
        FUS_STAGE_LOAD(0, 0)
        for (int64_t t0 = 0; t0 < seqlen; t0 += kM2R) {
            f32x16 acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
            for (int k0 = 0; k0 < d_model; k0 += kM2K) {
                __syncthreads();
                FUS_STAGE_STORE()
                __syncthreads();
                if (k0 + kM2K < d_model) {
                    FUS_STAGE_LOAD(t0, k0 + kM2K)
                } else if (t0 + kM2R < seqlen) {
                    FUS_STAGE_LOAD(t0 + kM2R, 0)       // next tile's first stage flies during the epilogue
                }
                // The operands of a quarter stage (4 k-pairs = 16 MFMAs) are fetched from LDS while the previous quarter's
                // MFMAs run (two register buffers, scheduling fences so the compiler keeps the order): with each load right
                // in front of its consumer -- where the scheduler puts it on its own -- every MFMA pair waited a full LDS
                // round trip and the matrix pipe idled 45 % of the time (two waves per SIMD do not cover that).
                float a0[2][4], a1[2][4], b0[2][4], b1[2][4];
                auto fetch = [&](int buf, int kb) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int kk = 2 * (kb + u) + kh;
                        a0[buf][u] = as[wm * 32 + l31][kk];            // tokens wm*32 ..      (first half of the tile)
                        a1[buf][u] = as[64 + wm * 32 + l31][kk];       // tokens 64 + wm*32 .. (second half)
                        b0[buf][u] = bs[kk][wn * 64 + l31];
                        b1[buf][u] = bs[kk][wn * 64 + 32 + l31];
                    }
                };
                fetch(0, 0);
#pragma unroll
                for (int g = 0; g < kM2K / 8; ++g) {
                    if (g + 1 < kM2K / 8) fetch((g + 1) & 1, 4 * (g + 1));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[g & 1][u], b0[g & 1][u], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[g & 1][u], b1[g & 1][u], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[g & 1][u], b0[g & 1][u], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[g & 1][u], b1[g & 1][u], acc[1][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- epilogue: two rounds of 64 tokens; in each, all 8 waves put phi of their 32 x 64 tile into LDS, then
            // every wave scans 8 tokens ----
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                if (tm) __syncthreads();                   // the first round's slabs have been read
                {
                    float* slab = (wn >= 2) ? kslab : qslab;
                    const int cbase = (wn & 1) * 64;
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[tm][tn][r] + bv[tn];
                            v = (v > 0.0f) ? v + 1.0f : __expf(v);              // elu(v) + 1
                            slab[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * kFusLd + cbase + tn * 32 + l31] = v;
                        }
                }
                __syncthreads();
                // tokens past the end of the sequence contribute nothing
                const int64_t tok0 = t0 + tm * 64 + wave * 8;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int rr = wave * 8 + u;
                    const bool ok = tok0 + u < seqlen;
                    s0 += ok ? (double)kslab[rr * kFusLd + lane] : 0.0;
                    s1 += ok ? (double)kslab[rr * kFusLd + 64 + lane] : 0.0;
                }
                ksum[wave][lane] = s0;
                ksum[wave][64 + lane] = s1;
                __syncthreads();
                double st0 = carry0, st1 = carry1, tot0 = 0.0, tot1 = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < 8; ++w2) {
                    const double a = ksum[w2][lane], b = ksum[w2][64 + lane];
                    st0 += (w2 < wave) ? a : 0.0;
                    st1 += (w2 < wave) ? b : 0.0;
                    tot0 += a;
                    tot1 += b;
                }
                carry0 += tot0;
                carry1 += tot1;
                double p0[8], p1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int rr = wave * 8 + u;
                    const bool ok = tok0 + u < seqlen;
                    const float k0v = kslab[rr * kFusLd + lane], k1v = kslab[rr * kFusLd + 64 + lane];
                    const float q0v = qslab[rr * kFusLd + lane], q1v = qslab[rr * kFusLd + 64 + lane];
                    st0 += ok ? (double)k0v : 0.0;
                    st1 += ok ? (double)k1v : 0.0;
                    p0[u] = (double)q0v * st0;                                    // dims 0..63   (head 0 when H = 2)
                    p1[u] = (double)q1v * st1;                                    // dims 64..127 (head 1 when H = 2)
                }
                const int64_t t = tok0 + (lane >> 3);                             // value index after the reduce = lane / 8
                if constexpr (HEADS == 1) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) p0[u] += p1[u];
                    treduce_t<double, 8, 32>(p0, lane);
                    if ((lane & 7) == 0 && t < seqlen) nu[seq * seqlen + t] = p0[0];
                } else {
                    treduce_t<double, 8, 32>(p0, lane);
                    treduce_t<double, 8, 32>(p1, lane);
                    if ((lane & 7) == 0 && t < seqlen) {
                        nu[(seq * 2 + 0) * seqlen + t] = p0[0];
                        nu[(seq * 2 + 1) * seqlen + t] = p1[0];
                    }
                }
            }
        }
#undef FUS_XLOAD
#undef FUS_WLOAD
#undef FUS_STAGE_LOAD
#undef FUS_XSTORE
#undef FUS_WSTORE
#undef FUS_STAGE_STORE
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
constexpr int kMaxDModelWt = 4096;   // hidden sizes up to this take the pre-transposed-W MFMA kernel

static int64_t attn_chunk(int64_t batch, int64_t seqlen, int d_qk) {
    const int64_t per_seq = seqlen * 2 * (int64_t)d_qk * 4;
    int64_t chunk = ((int64_t)1 << 31) / (per_seq > 0 ? per_seq : 1);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if (chunk > 32768) chunk = 32768;          // softmax grid.z and the nu/m scratch stay bounded
    if (const char* e = getenv("EGS_ATTN_CHUNK")) {          // test hook: force several batch chunks
        const int64_t v = atoll(e);
        if (v >= 1 && v < chunk) chunk = v;
    }
    return chunk;
}

size_t attn_workspace_bytes(int64_t batch, int64_t seqlen, int d_qk, int num_heads) {
    const int64_t chunk = attn_chunk(batch, seqlen, d_qk);
    size_t qk = (size_t)chunk * seqlen * 2 * d_qk * sizeof(float);
    size_t nu = (size_t)chunk * num_heads * seqlen * sizeof(double);
    size_t m = (size_t)chunk * num_heads * seqlen * sizeof(float);
    const int64_t nchunks = (seqlen + kScanChunk - 1) / kScanChunk;
    size_t cs = (size_t)chunk * nchunks * d_qk * sizeof(double);
    size_t wt = (size_t)2 * d_qk * kMaxDModelWt * sizeof(float);      // transposed q,k rows of Wqkv
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(qk) + up(nu) + up(m) + up(cs) + up(wt);
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
    double* csum = reinterpret_cast<double*>(reinterpret_cast<char*>(mrow) +
                                             up((size_t)chunk * num_heads * seqlen * sizeof(float)));
    const int nchunks = (int)((seqlen + kScanChunk - 1) / kScanChunk);
    float* wt = reinterpret_cast<float*>(reinterpret_cast<char*>(csum) +
                                         up((size_t)chunk * nchunks * d_qk * sizeof(double)));
    const int n_cols = 2 * d_qk;
    const bool mfma_ok = (d_model % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(w_qk) & 15) == 0);
    const bool mfma2_ok = mfma_ok && (n_cols % 4 == 0) && d_model <= kMaxDModelWt;
    if (mfma2_ok) {
        const int tot = n_cols * d_model;
        hipLaunchKernelGGL(wt_transpose_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, w_qk, n_cols,
                           d_model, wt);
        EGS_LAUNCH_CHECK();
    }
    // fused linear-attention path: all projected columns in one 256-wide tile, one workgroup per sequence
    const bool fused_lin = mfma2_ok && !softmax && n_cols == kM2C && (num_heads == 1 || num_heads == 2) &&
                           batch >= 128 && !getenv("EGS_LINATTN_UNFUSED");
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = (batch - b0 < chunk) ? (batch - b0) : chunk;
        const int64_t rows = nb * seqlen;
        if (fused_lin) {
            const float* xin = x + b0 * seqlen * d_model;
            const int cus = current_device_cus();
            const unsigned fgrid = (unsigned)(nb < cus ? nb : cus);           // one resident workgroup per CU walks sequences
            const size_t fdyn = (size_t)2 * 64 * kFusLd * sizeof(float);
            if (num_heads == 1) {
                EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_fused_kernel<1>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)fdyn));
                hipLaunchKernelGGL(linattn_fused_kernel<1>, dim3(fgrid), dim3(kFusT), fdyn, stream, xin, nb, seqlen, d_model,
                                   wt, b_qk, nu);
            } else {
                EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linattn_fused_kernel<2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)fdyn));
                hipLaunchKernelGGL(linattn_fused_kernel<2>, dim3(fgrid), dim3(kFusT), fdyn, stream, xin, nb, seqlen, d_model,
                                   wt, b_qk, nu);
            }
            EGS_LAUNCH_CHECK();
            const int64_t total = nb * (seqlen - 1) * num_heads;
            const int64_t fb = (total + 255) / 256;
            hipLaunchKernelGGL(eta_finalize_kernel, dim3((unsigned)(fb < 8192 ? fb : 8192)), dim3(256), 0, stream, nu,
                               mrow, nb, seqlen, num_heads, 0, eta + b0 * (seqlen - 1) * num_heads);
            EGS_LAUNCH_CHECK();
            continue;
        }
        const dim3 pgrid((unsigned)((rows + kBM - 1) / kBM), (unsigned)((n_cols + kBN - 1) / kBN));
        const float* xin = x + b0 * seqlen * d_model;
        const bool fused_csum = mfma2_ok && !softmax && (seqlen % 32 == 0);
        if (mfma2_ok) {
            const bool narrow = n_cols <= 128;                         // 256 x 128 block tile instead of 128 x 256
            const int tr = narrow ? 256 : 128, tc = narrow ? 128 : 256;
            const dim3 mgrid((unsigned)((rows + tr - 1) / tr), (unsigned)((n_cols + tc - 1) / tc));
            ChunkSumArgs cs{nullptr, seqlen, d_qk, num_heads, nchunks, kScanChunk};
            if (fused_csum) {
                cs.csum = csum;
                EGS_HIP(hipMemsetAsync(csum, 0, (size_t)nb * nchunks * d_qk * sizeof(double), stream));
            }
#define EGS_PROJ_LAUNCH(PHI, WN)                                                                                          \
    hipLaunchKernelGGL((qk_proj_mfma2_kernel<PHI, WN>), mgrid, dim3(256), 0, stream, xin, rows, d_model, wt, b_qk, n_cols, \
                       qk, cs)
            if (softmax) {
                if (narrow) EGS_PROJ_LAUNCH(false, 1);
                else EGS_PROJ_LAUNCH(false, 2);
            } else {
                if (narrow) EGS_PROJ_LAUNCH(true, 1);
                else EGS_PROJ_LAUNCH(true, 2);
            }
#undef EGS_PROJ_LAUNCH
        } else if (softmax)
            hipLaunchKernelGGL(qk_proj_kernel<false>, pgrid, dim3(256), 0, stream, xin, rows, d_model, w_qk, b_qk,
                               n_cols, qk);
        else
            hipLaunchKernelGGL(qk_proj_kernel<true>, pgrid, dim3(256), 0, stream, xin, rows, d_model, w_qk, b_qk,
                               n_cols, qk);
        EGS_LAUNCH_CHECK();
        if (softmax) {
            const int st = softmax_nu_dispatch(qk, nb, seqlen, d_qk, num_heads, nu, mrow, stream);
            if (st) return st;
        } else {
            const int64_t tasks = nb * num_heads * nchunks;
            const unsigned grid = (unsigned)((tasks + 3) / 4);
#define EGS_SCAN_LAUNCH(JJ)                                                                                              \
    if (!fused_csum)                                                                                                     \
        hipLaunchKernelGGL(linattn_chunksum_kernel<JJ>, dim3(grid), dim3(256), 0, stream, qk, nb, seqlen, d_qk,         \
                           num_heads, nchunks, csum);                                                                    \
    hipLaunchKernelGGL(linattn_scan_kernel<JJ>, dim3(grid), dim3(256), 0, stream, qk, nb, seqlen, d_qk, num_heads,      \
                       nchunks, csum, nu);
            if (hd <= 64) { EGS_SCAN_LAUNCH(1) }
            else if (hd <= 128) { EGS_SCAN_LAUNCH(2) }
            else if (hd <= 256) { EGS_SCAN_LAUNCH(4) }
            else { EGS_SCAN_LAUNCH(8) }
#undef EGS_SCAN_LAUNCH
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

// linear-attention eta from q, k that already exist (raw projections, phi = elu + 1 applied on load): the scan half of
// get_eig_att_linear (analysis/eval_eig.py:113-130) -- SURVEY.md 8(b) lists the (q, k) input form next to the x form
size_t linattn_eta_qk_workspace_bytes(int64_t batch, int64_t seqlen, int d_qk, int num_heads) {
    const int64_t nchunks = (seqlen + kScanChunk - 1) / kScanChunk;
    return (size_t)batch * num_heads * seqlen * sizeof(double) + (size_t)batch * nchunks * d_qk * sizeof(double) + 512;
}

int linattn_eta_qk_launch(const float* qk, int64_t batch, int64_t seqlen, int d_qk, int num_heads, double* eta,
                          void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (batch == 0 || seqlen <= 1) return 0;
    const int hd = d_qk / num_heads;
    if (hd > 512) {
        set_error("linear-attention scan supports head_dim <= 512 (got %d)", hd);
        return -4;
    }
    if (workspace_bytes < linattn_eta_qk_workspace_bytes(batch, seqlen, d_qk, num_heads)) {
        set_error("workspace too small: %zu < %zu", workspace_bytes,
                  linattn_eta_qk_workspace_bytes(batch, seqlen, d_qk, num_heads));
        return -7;
    }
    const int nchunks = (int)((seqlen + kScanChunk - 1) / kScanChunk);
    double* nu = static_cast<double*>(workspace);
    double* csum = nu + batch * num_heads * seqlen;
    const int64_t tasks = batch * num_heads * nchunks;
    const unsigned grid = (unsigned)((tasks + 3) / 4);
#define EGS_SCAN_QK(JJ)                                                                                                   \
    hipLaunchKernelGGL((linattn_chunksum_kernel<JJ, true>), dim3(grid), dim3(256), 0, stream, qk, batch, seqlen, d_qk,  \
                       num_heads, nchunks, csum);                                                                        \
    hipLaunchKernelGGL((linattn_scan_kernel<JJ, true>), dim3(grid), dim3(256), 0, stream, qk, batch, seqlen, d_qk,      \
                       num_heads, nchunks, csum, nu);
    if (hd <= 64) { EGS_SCAN_QK(1) }
    else if (hd <= 128) { EGS_SCAN_QK(2) }
    else if (hd <= 256) { EGS_SCAN_QK(4) }
    else { EGS_SCAN_QK(8) }
#undef EGS_SCAN_QK
    EGS_LAUNCH_CHECK();
    const int64_t total = batch * (seqlen - 1) * num_heads;
    const int64_t fb = (total + 255) / 256;
    hipLaunchKernelGGL(eta_finalize_kernel, dim3((unsigned)(fb < 8192 ? fb : 8192)), dim3(256), 0, stream, nu,
                       (const float*)nullptr, batch, seqlen, num_heads, 0, eta);
    EGS_LAUNCH_CHECK();
    return 0;
}

// softmax eta from q, k that already exist (hooked projections of a language model, notebooks/lm_eigvals.ipynb cell 13):
// qk is the packed [B, L, 2 d_qk] buffer; scratch = nu (fp64) + row maxima (fp32) for the whole batch
size_t softmax_eta_qk_workspace_bytes(int64_t batch, int64_t seqlen, int num_heads) {
    return (size_t)batch * num_heads * seqlen * (sizeof(double) + sizeof(float)) + 512;
}

int softmax_eta_qk_launch(const float* qk, int64_t batch, int64_t seqlen, int d_qk, int num_heads, double* eta,
                          void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (batch == 0 || seqlen <= 1) return 0;
    if (workspace_bytes < softmax_eta_qk_workspace_bytes(batch, seqlen, num_heads)) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, softmax_eta_qk_workspace_bytes(batch, seqlen, num_heads));
        return -7;
    }
    double* nu = static_cast<double*>(workspace);
    float* mrow = reinterpret_cast<float*>(nu + batch * num_heads * seqlen);
    const int st = softmax_nu_dispatch(qk, batch, seqlen, d_qk, num_heads, nu, mrow, stream);
    if (st) return st;
    EGS_LAUNCH_CHECK();
    const int64_t total = batch * (seqlen - 1) * num_heads;
    const int64_t fb = (total + 255) / 256;
    hipLaunchKernelGGL(eta_finalize_kernel, dim3((unsigned)(fb < 8192 ? fb : 8192)), dim3(256), 0, stream, nu, mrow, batch,
                       seqlen, num_heads, 1, eta);
    EGS_LAUNCH_CHECK();
    return 0;
}

}  // namespace egs
