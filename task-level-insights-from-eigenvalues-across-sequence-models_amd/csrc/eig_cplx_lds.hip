// K5 (complex, n <= 64): single-shift QR on complex128 Hessenberg matrices, ONE WAVEFRONT PER MATRIX, LDS-resident.
//
// This is the complex counterpart of hqr_real_kernel (eig_real_lds.hip) for the sizes the reference's S4 configs use
// (state_dim = 64: cifar / imdb / listops, analysis/eval_eig.py:296): same mapping -- lane = column for the row
// operation, lane = row for the column operation, two Hessenberg matrices packed into one n x (n+5) LDS array (one
// stored in columns >= row+3, its partner transposed in columns <= row+2), bulge column forwarded through v_readlane,
// deflation and start searches lane-parallel + ballot, no workgroup barrier anywhere.  The Hessenberg reduction itself
// is done by zhess64_kernel below (one workgroup of 4 waves per matrix) into the workspace.
#include "egs_common.h"
#include "zq_common.h"

namespace egs {


// ---------------------------------------------------------------------------------------------------
// Hessenberg reduction for n <= 64, one workgroup (4 waves) per matrix, matrix in LDS (ld = n+1: row and column
// sweeps conflict-free for 16-byte elements).  Each reflector is applied from both sides as ONE rank-2 update
//     H^H A H = A - w v^H - v y^H,   p = A v,  q^H = v^H A,  a2 = v^H p,  c = |tau|^2 a2 / 2,
//     w = tau p - c v,   y^H = conj(tau) q^H - c v^H,
// with the streamed index of both passes split over the 4 waves (partials combined through LDS in a fixed order, so
// every wave holds bit-identical scalars).  Output: dense H (exact zeros below the subdiagonal) + info (0, or n for
// NaN / Inf input together with a NaN spectrum, like numpy.linalg.eigvals' finiteness check).
// ---------------------------------------------------------------------------------------------------
constexpr int kZH = 256;

template <bool INREAL>
__global__ __launch_bounds__(kZH) void zhess64_kernel(const double* __restrict__ a, int n, int lda, int64_t stride,
                                                      double* __restrict__ hout, double* __restrict__ w,
                                                      int32_t* __restrict__ info, int balance) {
    extern __shared__ __attribute__((aligned(16))) double dynz[];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t mat = blockIdx.x;
    const int ldh = n + 1;
    zq* H = reinterpret_cast<zq*>(dynz);
    zq* vsh = H + (size_t)n * ldh;            // [64]
    zq* ysh = vsh + 64;                       // [4 waves][64]  private y^H per wave
    zq* part = ysh + 4 * 64;                  // [4 waves][p | q][64]
#define ZH(i, j) H[(i) * ldh + (j)]
    if (tid == 0) s_bad = 0;
    __syncthreads();
    {
        const double* src = a + mat * stride * (INREAL ? 1 : 2);
        for (int idx = tid; idx < n * n; idx += kZH) {
            const int i = idx / n, j = idx - i * n;
            zq v;
            if (INREAL) v = {src[(int64_t)i * lda + j], 0.0};
            else v = {src[2 * ((int64_t)i * lda + j)], src[2 * ((int64_t)i * lda + j) + 1]};
            ZH(i, j) = v;
            if (!(fabs(v.re) <= 1.7976931348623157e308) || !(fabs(v.im) <= 1.7976931348623157e308)) s_bad = 1;
        }
    }
    __syncthreads();
    if (s_bad != 0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        double* wbad = w + mat * 2 * (int64_t)n;
        for (int j = tid; j < 2 * n; j += kZH) wbad[j] = qnan;
        if (tid == 0) info[mat] = n;
        return;
    }
    // xGEEV-style scaling: a matrix whose largest entry is outside [1e-100, 1e100] is scaled by a power of two (exact),
    // the QR kernel multiplies the eigenvalues back.  Without it the squared moduli of the rotations / reflectors
    // underflow (entries of 1e-160 gave wrong spectra without any flag) or overflow.  The exponent travels in the first
    // slot of this matrix's eigenvalue array, which the QR kernel reads before it writes anything.
    {
        double amax = 0.0;
        for (int idx = tid; idx < n * n; idx += kZH) {
            const zq v = H[(idx / n) * ldh + (idx % n)];
            amax = fmax(amax, fmax(fabs(v.re), fabs(v.im)));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
        double* red = reinterpret_cast<double*>(part);
        if (lane == 0) red[wv] = amax;
        __syncthreads();
        amax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        __syncthreads();
        int e = 0;
        if (amax > 0.0 && (amax < 1e-100 || amax > 1e100)) e = ilogb(amax);
        if (e != 0) {
            for (int idx = tid; idx < n * n; idx += kZH) {
                zq& v = H[(idx / n) * ldh + (idx % n)];
                v = {scalbn(v.re, -e), scalbn(v.im, -e)};      // (2^-e itself overflows for denormal input)
            }
            __syncthreads();
        }
        if (tid == 0) w[mat * 2 * (int64_t)n] = (double)e;
    }
    const int lc = (lane < n) ? lane : n - 1;
    const bool inr = lane < n;
    // ---- xGEBAL (permute + scale), by wave 0; same scheme as hess_real_kernel: isolated eigenvalues are moved to the
    // ends of the diagonal (the reduction keeps their exact zeros, the QR stage deflates them at once) and the active
    // block is equilibrated with power-of-two factors ----
    if (balance && wv == 0) {
#define ZNZ(i, j) (ZH(i, j).re != 0.0 || ZH(i, j).im != 0.0)
        int k0 = 0, l0 = n - 1;
        for (int guard = 0; guard < n; ++guard) {
            bool rz = false;
            if (lane <= l0) {
                rz = true;
                for (int j = 0; j <= l0; ++j)
                    if (j != lane && ZNZ(lane, j)) rz = false;
            }
            const unsigned long long m = __ballot(rz);
            if (m == 0ULL || l0 == 0) break;
            const int r = uniform(63 - __builtin_clzll(m));
            if (r != l0) {
                if (lane <= l0) {
                    const zq t = ZH(lane, r);
                    ZH(lane, r) = ZH(lane, l0);
                    ZH(lane, l0) = t;
                }
                wave_lds_sync();
                if (inr) {
                    const zq t = ZH(r, lane);
                    ZH(r, lane) = ZH(l0, lane);
                    ZH(l0, lane) = t;
                }
                wave_lds_sync();
            }
            --l0;
        }
        for (int guard = 0; guard < n; ++guard) {
            bool cz = false;
            if (lane >= k0 && lane <= l0) {
                cz = true;
                for (int i = k0; i <= l0; ++i)
                    if (i != lane && ZNZ(i, lane)) cz = false;
            }
            const unsigned long long m = __ballot(cz);
            if (m == 0ULL || k0 >= l0) break;
            const int c = uniform(__builtin_ctzll(m));
            if (c != k0) {
                if (lane <= l0) {
                    const zq t = ZH(lane, c);
                    ZH(lane, c) = ZH(lane, k0);
                    ZH(lane, k0) = t;
                }
                wave_lds_sync();
                if (inr && lane >= k0) {
                    const zq t = ZH(c, lane);
                    ZH(c, lane) = ZH(k0, lane);
                    ZH(k0, lane) = t;
                }
                wave_lds_sync();
            }
            ++k0;
        }
        for (int pass = 0; pass < 16; ++pass) {
            double f = 1.0;
            if (lane >= k0 && lane <= l0) {
                double c = 0.0, r = 0.0;
                for (int j = k0; j <= l0; ++j) {
                    if (j != lane) {
                        c += fabs(ZH(j, lane).re) + fabs(ZH(j, lane).im);
                        r += fabs(ZH(lane, j).re) + fabs(ZH(lane, j).im);
                    }
                }
                if (c != 0.0 && r != 0.0 && c < 1e300 && r < 1e300) {
                    const double s0 = c + r;
                    double g = r * 0.5;
                    while (c < g) {
                        f *= 2.0;
                        c *= 4.0;
                    }
                    g = r * 2.0;
                    while (c >= g) {
                        f *= 0.5;
                        c *= 0.25;
                    }
                    if ((c + r) / f >= 0.95 * s0) f = 1.0;
                }
            }
            if (__ballot(f != 1.0) == 0ULL) break;
            const double finv = 1.0 / f;                              // powers of two: exact
            for (int r = 0; r < n; ++r) {
                const double sc = f * bcast_lane(finv, r);
                if (inr) ZH(r, lane) = qscale(ZH(r, lane), sc);
            }
            wave_lds_sync();
        }
#undef ZNZ
    }
    __syncthreads();
    for (int k = 0; k < n - 2; ++k) {
        // every wave reads column k (lane = row) and forms the reflector redundantly
        const zq xi = (lane > k && inr) ? ZH(lane, k) : zq{0.0, 0.0};
        const zq alpha = qbcast(xi, k + 1);
        const double xnorm2 = wave_sum((lane > k + 1) ? fma(xi.re, xi.re, xi.im * xi.im) : 0.0);
        // block-uniform.  A column whose squared norm is below 1e-290 is left alone: its squares underflow, the norm would
        // come out as zero and tau as NaN (c (1 + i/2) ones(64, 64) did that); the matrix is scaled so that its largest
        // entry is >= 1e-100, i.e. such a column is below 1e-45 of it.  The QR stage drops what stays under the sub-diagonal.
        const double tot = fma(alpha.re, alpha.re, alpha.im * alpha.im) + xnorm2;
        const bool act = !(xnorm2 == 0.0 && alpha.im == 0.0) && tot >= 1e-290;
        const double nrm = sqrt(tot);
        const double beta = act ? -copysign(nrm, alpha.re) : alpha.re;
        const zq tau = act ? zq{(beta - alpha.re) / beta, -alpha.im / beta} : zq{0.0, 0.0};
        const zq sc = act ? qdiv({1.0, 0.0}, {alpha.re - beta, alpha.im}) : zq{0.0, 0.0};
        const zq v = !act ? zq{0.0, 0.0}
                          : ((lane == k + 1) ? zq{1.0, 0.0} : ((lane > k + 1 && inr) ? qmul(xi, sc) : zq{0.0, 0.0}));
        __syncthreads();                                   // every wave has read column k
        if (wv == 0) {
            vsh[lane] = v;
            if (act && lane > k && inr) ZH(lane, k) = (lane == k + 1) ? zq{beta, 0.0} : zq{0.0, 0.0};
        }
        __syncthreads();
        // pass 1: this wave's quarter of t = k+1..n-1;  p_i (lane = row i),  q_j = (v^H A)_j (lane = column j)
        const int len = n - k - 1;
        const int qlen = (len + 3) >> 2;
        const int t0 = k + 1 + wv * qlen;
        const int t1 = (t0 + qlen < n) ? t0 + qlen : n;
        zq pp = {0.0, 0.0}, qq = {0.0, 0.0};
        {
            int t = t0;
            for (; t + 4 <= t1; t += 4) {
                zq ar[4], ac[4], vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ar[u] = ZH(lc, t + u);
                    ac[u] = ZH(t + u, lc);
                    vv[u] = vsh[t + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pp = qadd(pp, qmul(ar[u], vv[u]));
                    qq = qadd(qq, qcmul(vv[u], ac[u]));
                }
            }
            for (; t < t1; ++t) {
                const zq vt = vsh[t];
                pp = qadd(pp, qmul(ZH(lc, t), vt));
                qq = qadd(qq, qcmul(vt, ZH(t, lc)));
            }
        }
        part[(wv * 2 + 0) * 64 + lane] = pp;
        part[(wv * 2 + 1) * 64 + lane] = qq;
        __syncthreads();
        zq p = {0.0, 0.0}, q = {0.0, 0.0};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            p = qadd(p, part[(x * 2 + 0) * 64 + lane]);
            q = qadd(q, part[(x * 2 + 1) * 64 + lane]);
        }
        if (!inr) p = {0.0, 0.0};
        if (!(lane > k && inr)) q = {0.0, 0.0};            // only columns j > k are updated
        const zq vp = qcmul(v, p);                        // conj(v_i) p_i
        const zq a2 = {wave_sum(vp.re), wave_sum(vp.im)};
        const double t2 = 0.5 * fma(tau.re, tau.re, tau.im * tau.im);
        const zq c = qscale(a2, t2);
        const zq wi = qsub(qmul(tau, p), qmul(c, v));                              // w_i,   lane = row
        ysh[wv * 64 + lane] = qsub(qcmul(tau, q), qmulc(c, v));                    // y^H_j, lane = column
        wave_lds_sync();
        // pass 2: this wave's quarter of the columns, lane = row:  A(i,j) -= w_i conj(v_j) + v_i y^H_j
        {
            int j = t0;
            for (; j + 4 <= t1; j += 4) {
                zq av[4], vv[4], yy[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    av[u] = ZH(lc, j + u);
                    vv[u] = vsh[j + u];
                    yy[u] = ysh[wv * 64 + j + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (inr) ZH(lane, j + u) = qsub(av[u], qadd(qmulc(wi, vv[u]), qmul(v, yy[u])));
            }
            for (; j < t1; ++j)
                if (inr) ZH(lane, j) = qsub(ZH(lane, j), qadd(qmulc(wi, vsh[j]), qmul(v, ysh[wv * 64 + j])));
        }
        __syncthreads();
    }
    zq* dst = reinterpret_cast<zq*>(hout) + mat * (int64_t)n * n;
    for (int idx = tid; idx < n * n; idx += kZH) {
        const int r = idx / n, c = idx - r * n;
        dst[idx] = (r > c + 1) ? zq{0.0, 0.0} : ZH(r, c);
    }
    if (tid == 0) info[mat] = 0;
#undef ZH
}

int zhess64_launch(const void* a, bool in_real, int64_t nmat, int n, int lda, int64_t stride, double* hout, double* w,
                   int32_t* info, hipStream_t stream) {
    if (nmat == 0 || n == 0) return 0;
    const size_t dyn = ((size_t)n * (n + 1) + 64 + 4 * 64 + 8 * 64) * 2 * sizeof(double);
    const double* ap = static_cast<const double*>(a);
    const int balance = getenv("EGS_NO_BALANCE") ? 0 : 1;
    if (in_real) {
        if (dyn > 48 * 1024)
            EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(zhess64_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(zhess64_kernel<true>, dim3((unsigned)nmat), dim3(kZH), dyn, stream, ap, n, lda, stride, hout, w,
                           info, balance);
    } else {
        if (dyn > 48 * 1024)
            EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(zhess64_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(zhess64_kernel<false>, dim3((unsigned)nmat), dim3(kZH), dyn, stream, ap, n, lda, stride, hout,
                           w, info, balance);
    }
    EGS_LAUNCH_CHECK();
    return 0;
}


template <class V>
__device__ __forceinline__ void zhqr_run(const V H, const double* __restrict__ src, int n, int lane,
                                         double* __restrict__ wo, int32_t* __restrict__ info, const double rescale) {
    // Hessenberg part plus the one extra sub-diagonal a single-shift bulge touches; 8 rows of 16-byte global loads in
    // flight at a time from always-valid addresses (clamped lane), zeroed / dropped by value afterwards
    const int lcl = (lane < n) ? lane : n - 1;
    const zq* srcz = reinterpret_cast<const zq*>(src);
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        zq t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = srcz[(int64_t)(i + u) * n + lcl];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (lane < n && i + u <= lane + 2) H.at(i + u, lane) = (i + u <= lane + 1) ? t[u] : zq{0.0, 0.0};
    }
    for (; i < n; ++i) {
        const zq t = srcz[(int64_t)i * n + lcl];
        if (lane < n && i <= lane + 2) H.at(i, lane) = (i <= lane + 1) ? t : zq{0.0, 0.0};
    }
    wave_lds_sync();
    zhqr_wave(H, n, lane, wo, info, rescale);
}

// info[mat] may arrive pre-set to n by the Hessenberg stage (non-finite input): such matrices are skipped.
__global__ __launch_bounds__(256) void zhqr_wave_kernel(const double* __restrict__ hin, int64_t nmat, int n,
                                                        double* __restrict__ w, int32_t* __restrict__ info,
                                                        int waves_per_block) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int64_t mat = (int64_t)blockIdx.x * waves_per_block + wave;
    if (mat >= nmat) return;
    if (info[mat] != 0) return;
    const int ldp = zhqr_pair_ld(n);
    zq* base = reinterpret_cast<zq*>(smem) + (size_t)(wave >> 1) * ((size_t)n * ldp);
    const double* src = hin + mat * 2 * (int64_t)n * n;
    double* wo = w + mat * 2 * (int64_t)n;
    // the Hessenberg stage left the power-of-two exponent it scaled this matrix by in the first eigenvalue slot
    const int sexp = __builtin_amdgcn_readfirstlane((int)wo[0]);
    const double rescale = (sexp != 0) ? scalbn(1.0, sexp) : 1.0;
    if (n == 64) {
        if ((wave & 1) == 0) zhqr_run(ZViewT<69, false>{base + 5, ldp}, src, n, lane, wo, info + mat, rescale);
        else zhqr_run(ZViewT<69, true>{base, ldp}, src, n, lane, wo, info + mat, rescale);
    } else {
        if ((wave & 1) == 0) zhqr_run(ZViewT<0, false>{base + 5, ldp}, src, n, lane, wo, info + mat, rescale);
        else zhqr_run(ZViewT<0, true>{base, ldp}, src, n, lane, wo, info + mat, rescale);
    }
}

int zhqr_wave_launch(const double* h, int64_t nmat, int n, double* w, int32_t* info, hipStream_t stream) {
    if (nmat == 0 || n == 0) return 0;
    const int ldp = zhqr_pair_ld(n);
    const size_t pair_bytes = (size_t)n * ldp * 2 * sizeof(double);
    int pairs = (int)((80 * 1024) / pair_bytes);       // two workgroups per CU when they fit
    if (pairs < 1) pairs = 1;
    if (pairs > 2) pairs = 2;
    const int waves = 2 * pairs;
    const size_t lds = pairs * pair_bytes;
    if (lds > 48 * 1024)
        EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(zhqr_wave_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((nmat + waves - 1) / waves);
    hipLaunchKernelGGL(zhqr_wave_kernel, dim3(grid), dim3(waves * kWave), lds, stream, h, nmat, n, w, info, waves);
    EGS_LAUNCH_CHECK();
    return 0;
}

}  // namespace egs
