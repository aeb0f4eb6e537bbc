// a9/a10: radius / phase transforms fused with the 7-bin inclusive-edge threshold histograms of
// analysis/eval_eig.py:335-391 (quirk Q12), so that only (nbins, outer, inner) percentages leave the GPU
// instead of the full spectra.
//
// vals is [outer, T, inner]; one count per (bin, outer, inner).  Counting is integer work: per-thread
// register counters, one 64-bit global atomic per (thread, bin) at the end, then a finalize pass that turns
// the counts into `count / T * 100` with the same two IEEE operations NumPy performs -- bit-exact.
#include <type_traits>

#include "egs_common.h"

namespace egs {

constexpr int kMaxEdges = 15;
// NE > 0: exactly NE ascending edges with e[0] >= 0 (the reference's two threshold sets have 6 and 5: eval_eig.py:591-592)
// -- a kernel argument of NE doubles; NE == 0: any edge set up to kMaxEdges (generic path).
template <int NE>
struct HistEdges {
    double e[NE > 0 ? NE : kMaxEdges];
    int n;
};

template <int DTYPE>
__device__ __forceinline__ double load_transformed(const void* vals, int64_t idx, int transform) {
    double re, im = 0.0;
    if constexpr (DTYPE == EGS_F32) {
        re = (double)static_cast<const float*>(vals)[idx];
    } else if constexpr (DTYPE == EGS_F64) {
        re = static_cast<const double*>(vals)[idx];
    } else if constexpr (DTYPE == EGS_C64) {
        const float2 v = static_cast<const float2*>(vals)[idx];
        if (transform == EGS_HIST_RADIUS) {            // float32 arithmetic like NumPy on complex64
            const float r2 = __fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y));
            return (double)__fsqrt_rn(r2);
        }
        if (transform == EGS_HIST_PHASE_DEG) {
            const float ph = atan2f(v.y, v.x);
            return (double)__fdiv_rn(__fmul_rn(ph, 180.0f), 3.14159274101257324f);
        }
        re = (double)v.x;
        im = (double)v.y;
    } else {
        const double2 v = static_cast<const double2*>(vals)[idx];
        re = v.x;
        im = v.y;
    }
    switch (transform) {
        case EGS_HIST_RADIUS:       // sqrt(re^2 + im^2), no FMA contraction (eval_eig.py:605, :726)
            return __dsqrt_rn(__dadd_rn(__dmul_rn(re, re), __dmul_rn(im, im)));
        case EGS_HIST_PHASE_DEG:    // atan2(im, re) * 180 / pi (eval_eig.py:614, :734)
            return __ddiv_rn(__dmul_rn(atan2(im, re), 180.0), 3.141592653589793);
        case EGS_HIST_ZERO:         // 0 * value (quirk Q13): NaN / inf stay NaN
            return __dmul_rn(0.0, re);
        default:
            return re;
    }
}

template <int DTYPE, int NE>
__global__ __launch_bounds__(256) void hist_count_kernel(const void* __restrict__ vals, int transform, int64_t outer,
                                                         int64_t t_len, int64_t inner, HistEdges<NE> ed,
                                                         int rows_per_block, unsigned long long* __restrict__ counts) {
    // block (x = T-slab, y = outer).  Threads cover `inner` contiguously; a thread's column is fixed, so its
    // bin counters live in registers.
    const int64_t o = blockIdx.y;
    const int tpc = (inner >= 256) ? 1 : (int)(256 / inner);       // rows handled concurrently
    const int nthreads_used = (inner >= 256) ? 256 : (int)(tpc * inner);
    const bool worker = (int)threadIdx.x < nthreads_used;
    const int64_t t_begin = (int64_t)blockIdx.x * rows_per_block;
    int64_t t_end = t_begin + rows_per_block;
    if (t_end > t_len) t_end = t_len;
    const int nb = ed.n + 1;
    // per-workgroup counters: the threads of a workgroup that share a column meet in LDS first, so that only one
    // global atomic per (bin, column, workgroup) reaches memory (same-address global atomics serialise)
    __shared__ unsigned lcnt[kMaxEdges + 1][256];
    for (int64_t c0 = 0; c0 < inner; c0 += 256) {                  // one pass unless inner > 256
        for (int idx = threadIdx.x; idx < (kMaxEdges + 1) * 256; idx += 256) (&lcnt[0][0])[idx] = 0u;
        __syncthreads();
        int64_t c = 0;
        int r0 = 0, rstep = 1;
        bool have = worker;
        if (inner >= 256) {
            c = c0 + threadIdx.x;
            have = c < inner;
        } else if (worker) {
            c = threadIdx.x % inner;
            r0 = (int)(threadIdx.x / inner);
            rstep = tpc;
        }
        if (have) {
            // NE > 0: cumulative counters -- ge[b] = #{v >= e_b}, gt[b] = #{v > e_b}, pos = #{v >= 0}: 2 NE + 1 compares per
            // value instead of the 4 x 16 predicated operations of the generic loop (which made the kernel VALU-bound at
            // a quarter of the HBM rate).  The inclusive bins follow exactly (ascending edges, e_0 >= 0; NaN and
            // negatives pass no compare):  bin 0 = pos - gt[0],  bin b = ge[b-1] - gt[b],  last = gt[NE-1].
            constexpr int NC = NE > 0 ? NE : 1;
            unsigned cnt[kMaxEdges + 1];
            unsigned ge[NC], gt[NC], pos = 0;
#pragma unroll
            for (int b = 0; b <= kMaxEdges; ++b) cnt[b] = 0;
#pragma unroll
            for (int b = 0; b < NC; ++b) ge[b] = gt[b] = 0;
            auto tally = [&](double v) {
                if constexpr (NE > 0) {
                    pos += (v >= 0.0) ? 1u : 0u;
#pragma unroll
                    for (int b = 0; b < NE; ++b) {
                        ge[b] += (v >= ed.e[b]) ? 1u : 0u;
                        gt[b] += (v > ed.e[b]) ? 1u : 0u;
                    }
                } else {
#pragma unroll
                    for (int b = 0; b <= kMaxEdges; ++b) {
                        if (b < nb) {
                            bool in;
                            if (b == 0) in = (v >= 0.0) && (v <= ed.e[0]);
                            else if (b == nb - 1) in = v > ed.e[b - 1];
                            else in = (v >= ed.e[b - 1]) && (v <= ed.e[b]);
                            cnt[b] += in ? 1u : 0u;
                        }
                    }
                }
            };
            int64_t t = t_begin + r0;
            for (; t + 3 * rstep < t_end; t += 4 * rstep) {        // four independent loads in flight per thread (eight: slower)
                const int64_t base = (o * t_len + t) * inner + c;
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = load_transformed<DTYPE>(vals, base + (int64_t)u * rstep * inner, transform);
#pragma unroll
                for (int u = 0; u < 4; ++u) tally(v[u]);
            }
            for (; t < t_end; t += rstep) tally(load_transformed<DTYPE>(vals, (o * t_len + t) * inner + c, transform));
            if constexpr (NE > 0) {
                cnt[0] = pos - gt[0];
#pragma unroll
                for (int b = 1; b < NE; ++b) cnt[b] = ge[b - 1] - gt[b];
                cnt[NE] = gt[NE - 1];
            }
            const int cl = (int)(c - c0);
#pragma unroll
            for (int b = 0; b <= kMaxEdges; ++b)
                if (b < nb && cnt[b]) atomicAdd(&lcnt[b][cl], cnt[b]);
        }
        __syncthreads();
        const int ncols = (inner - c0 < 256) ? (int)(inner - c0) : 256;
        for (int idx = threadIdx.x; idx < nb * ncols; idx += 256) {
            const int b = idx / ncols, cl = idx - b * ncols;
            const unsigned v = lcnt[b][cl];
            if (v) atomicAdd(&counts[((int64_t)b * outer + o) * inner + c0 + cl], (unsigned long long)v);
        }
        __syncthreads();
    }
}

__global__ void hist_finalize_kernel(double* __restrict__ pct, int64_t n, int64_t t_len) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long c = reinterpret_cast<const unsigned long long*>(pct)[i];
    pct[i] = __dmul_rn(__ddiv_rn((double)c, (double)t_len), 100.0);   // count / T * 100
}

int threshold_hist_launch(const void* vals, int dtype, int transform, int64_t outer, int64_t t_len, int64_t inner,
                          const double* edges_host, int nedges, double* pct, hipStream_t stream) {
    const int64_t nout = (int64_t)(nedges + 1) * outer * inner;
    if (nout == 0) return 0;
    EGS_HIP(hipMemsetAsync(pct, 0, (size_t)nout * sizeof(double), stream));
    if (t_len > 0) {
        // enough T-slabs to fill the chip even when outer is small
        int64_t slabs = (2048 + outer - 1) / outer;
        int64_t rows = (t_len + slabs - 1) / slabs;
        if (rows < 64) rows = 64;
        slabs = (t_len + rows - 1) / rows;
        const dim3 grid((unsigned)slabs, (unsigned)outer);
        auto* counts = reinterpret_cast<unsigned long long*>(pct);
        bool fast = (nedges == 5 || nedges == 6) && edges_host[0] >= 0.0;
        for (int i = 1; i < nedges && fast; ++i) fast = edges_host[i - 1] <= edges_host[i];
        auto launch = [&](auto ne_tag) {
            constexpr int NE = decltype(ne_tag)::value;
            HistEdges<NE> ed;
            ed.n = nedges;
            for (int i = 0; i < (NE > 0 ? NE : kMaxEdges); ++i) ed.e[i] = (i < nedges) ? edges_host[i] : 0.0;
            switch (dtype) {
                case EGS_F32:
                    hipLaunchKernelGGL((hist_count_kernel<EGS_F32, NE>), grid, dim3(256), 0, stream, vals, transform, outer,
                                       t_len, inner, ed, (int)rows, counts);
                    break;
                case EGS_F64:
                    hipLaunchKernelGGL((hist_count_kernel<EGS_F64, NE>), grid, dim3(256), 0, stream, vals, transform, outer,
                                       t_len, inner, ed, (int)rows, counts);
                    break;
                case EGS_C64:
                    hipLaunchKernelGGL((hist_count_kernel<EGS_C64, NE>), grid, dim3(256), 0, stream, vals, transform, outer,
                                       t_len, inner, ed, (int)rows, counts);
                    break;
                default:
                    hipLaunchKernelGGL((hist_count_kernel<EGS_C128, NE>), grid, dim3(256), 0, stream, vals, transform,
                                       outer, t_len, inner, ed, (int)rows, counts);
                    break;
            }
        };
        if (fast && nedges == 6) launch(std::integral_constant<int, 6>{});
        else if (fast && nedges == 5) launch(std::integral_constant<int, 5>{});
        else launch(std::integral_constant<int, 0>{});
        EGS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(hist_finalize_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, stream, pct, nout,
                       t_len);
    EGS_LAUNCH_CHECK();
    return 0;
}

}  // namespace egs
