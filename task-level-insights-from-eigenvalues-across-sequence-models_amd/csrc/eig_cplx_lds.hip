// K5 (complex, n <= 64): single-shift QR on complex128 Hessenberg matrices, ONE WAVEFRONT PER MATRIX, LDS-resident.
//
// This is the complex counterpart of hqr_real_kernel (eig_real_lds.hip) for the sizes the reference's S4 configs use
// (state_dim = 64: cifar / imdb / listops, analysis/eval_eig.py:296): same mapping -- lane = column for the row
// operation, lane = row for the column operation, two Hessenberg matrices packed into one n x (n+5) LDS array (one
// stored in columns >= row+3, its partner transposed in columns <= row+2), bulge column forwarded through v_readlane,
// deflation and start searches lane-parallel + ballot, no workgroup barrier anywhere.  The Hessenberg reduction itself
// is done by zhess64_kernel below (one workgroup of 4 waves per matrix) into the workspace.
#include "egs_common.h"

namespace egs {

struct zq {
    double re, im;
};
__device__ __forceinline__ zq qadd(zq a, zq b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ zq qsub(zq a, zq b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ zq qmul(zq a, zq b) {
    return {fma(a.re, b.re, -a.im * b.im), fma(a.re, b.im, a.im * b.re)};
}
__device__ __forceinline__ zq qmulc(zq a, zq b) {   // a * conj(b)
    return {fma(a.re, b.re, a.im * b.im), fma(a.im, b.re, -a.re * b.im)};
}
__device__ __forceinline__ zq qcmul(zq a, zq b) {   // conj(a) * b
    return {fma(a.re, b.re, a.im * b.im), fma(a.re, b.im, -a.im * b.re)};
}
__device__ __forceinline__ zq qscale(zq a, double s) { return {a.re * s, a.im * s}; }
__device__ __forceinline__ double qabs1(zq a) { return fabs(a.re) + fabs(a.im); }
__device__ __forceinline__ zq qdiv(zq a, zq b) {
    if (fabs(b.re) >= fabs(b.im)) {
        const double r = b.im / b.re, d = b.re + b.im * r;
        return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
    }
    const double r = b.re / b.im, d = b.im + b.re * r;
    return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
__device__ __forceinline__ zq qsqrt(zq a) {
    const double m = hypot(a.re, a.im);
    if (m == 0.0) return {0.0, 0.0};
    const double sr = sqrt(0.5 * (m + fabs(a.re)));
    const double si = 0.5 * a.im / sr;
    if (a.re >= 0.0) return {sr, si};
    return {fabs(si), copysign(sr, a.im)};
}
__device__ __forceinline__ zq qbcast(zq x, int src) { return {bcast_lane(x.re, src), bcast_lane(x.im, src)}; }

// element (i,j) at hm[i*sr + j*sc] (complex elements); LDP > 0: compile-time pair leading dimension
template <int LDP, bool TR>
struct ZViewT {
    static constexpr bool kFast = LDP > 0;
    zq* hm;
    int ld;
    __device__ __forceinline__ int sr() const { return TR ? 1 : (LDP > 0 ? LDP : ld); }
    __device__ __forceinline__ int sc() const { return TR ? (LDP > 0 ? LDP : ld) : 1; }
    __device__ __forceinline__ zq& at(int i, int j) const { return hm[i * sr() + j * sc()]; }
};

__host__ __device__ inline int zhqr_pair_ld(int n) {
    int ld = n + 5;
    return (ld & 1) ? ld : ld + 1;
}

// xLARFG for (alpha; x), complex, branch-free: H = I - tau u u^H, u = (1; v2), H^H (alpha; x) = (beta; 0), beta real.
// A zero x gives a unitary diagonal H (it only rotates the phase of alpha onto the real axis) instead of LAPACK's
// identity; the fmax guard keeps an all-zero column finite (tau = 0, v2 = 0).
__device__ __forceinline__ void zlarfg2_fast(zq alpha, zq x, double& beta, zq& v2, zq& tau) {
    const double tiny = 1e-300;
    const double an2 = fma(alpha.re, alpha.re, alpha.im * alpha.im);
    const double nrm2 = an2 + fma(x.re, x.re, x.im * x.im);
    const double rn = fast_rsqrt(fmax(nrm2, tiny));
    const double nrm = nrm2 * rn;
    beta = -copysign(nrm, alpha.re);
    const double rb = -copysign(rn, alpha.re);                    // 1 / beta
    tau = {(beta - alpha.re) * rb, -alpha.im * rb};
    const double dr = alpha.re - beta;                            // alpha - beta = (dr, alpha.im), |dr| >= |alpha.re|
    const double d2 = fma(dr, dr, alpha.im * alpha.im);
    const double rd = fast_rcp(fmax(d2, tiny));
    const zq sc = {dr * rd, -alpha.im * rd};                      // 1 / (alpha - beta)
    v2 = qmul(x, sc);
}

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
    extern __shared__ double dynz[];
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
        const bool act = !(xnorm2 == 0.0 && alpha.im == 0.0);          // block-uniform
        const double nrm = sqrt(fma(alpha.re, alpha.re, alpha.im * alpha.im) + xnorm2);
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
__device__ void zhqr_wave(const V H, const int n, const int lane, double* __restrict__ w_out,
                          int32_t* __restrict__ info_out) {
    const double ulp = 2.220446049250313e-16;
    const double safmin = 2.2250738585072014e-308;
    const double smlnum = safmin * ((double)n / ulp);
    const int itmax = 30 * (n > 10 ? n : 10);
    const int kexsh = 10;
    const int SR = H.sr(), SC = H.sc();
    const int lc = (lane < n) ? lane : n - 1;

    int info = 0;
    int i = n - 1;
    int kdefl = 0;
    while (i >= 0) {
        int l = 0;
        bool converged = false;
        for (int its = 0; its <= itmax; ++its) {
            // ---- small-subdiagonal search: lane k tests H(k,k-1), l < k <= i ----
            bool small = false;
            if (lane > l && lane <= i) {
                const int k = lane;
                const double hkk1 = qabs1(H.at(k, k - 1));
                if (hkk1 <= smlnum) {
                    small = true;
                } else {
                    const zq hkk = H.at(k, k), hk1k1 = H.at(k - 1, k - 1);
                    double tst = qabs1(hk1k1) + qabs1(hkk);
                    if (tst == 0.0) {
                        if (k - 2 >= 0) tst += qabs1(H.at(k - 1, k - 2));
                        if (k + 1 <= n - 1) tst += qabs1(H.at(k + 1, k));
                    }
                    if (hkk1 <= ulp * tst) {
                        const double hk1k = qabs1(H.at(k - 1, k));
                        const double ab = fmax(hkk1, hk1k), ba = fmin(hkk1, hk1k);
                        const double df = qabs1(qsub(hk1k1, hkk));
                        const double aa = fmax(qabs1(hkk), df), bb = fmin(qabs1(hkk), df);
                        const double s = aa + ab;
                        if (ba * (ab / s) <= fmax(smlnum, ulp * (bb * (aa / s)))) small = true;
                    }
                }
            }
            const unsigned long long smask = __ballot(small);
            l = uniform(smask ? (63 - __builtin_clzll(smask)) : l);
            if (l > 0 && lane == 0) H.at(l, l - 1) = {0.0, 0.0};
            if (l >= i) {
                converged = true;
                break;
            }
            wave_lds_sync();
            ++kdefl;
            // ---- shift (uniform) ----
            zq t;
            if (kdefl % (2 * kexsh) == 0) {
                t = qadd({0.75 * qabs1(H.at(i, i - 1)), 0.0}, H.at(i, i));
            } else if (kdefl % kexsh == 0) {
                t = qadd({0.75 * qabs1(H.at(l + 1, l)), 0.0}, H.at(l, l));
            } else {
                t = H.at(i, i);
                const zq u = qmul(qsqrt(H.at(i - 1, i)), qsqrt(H.at(i, i - 1)));
                double s = qabs1(u);
                if (s != 0.0) {
                    const zq x = qscale(qsub(H.at(i - 1, i - 1), t), 0.5);
                    const double sx = qabs1(x);
                    s = fmax(s, sx);
                    const zq xs = qscale(x, 1.0 / s), us = qscale(u, 1.0 / s);
                    zq y = qscale(qsqrt(qadd(qmul(xs, xs), qmul(us, us))), s);
                    if (sx > 0.0) {
                        const zq xn = qscale(x, 1.0 / sx);
                        if (xn.re * y.re + xn.im * y.im < 0.0) y = {-y.re, -y.im};
                    }
                    t = qsub(t, qmul(u, qdiv(u, qadd(x, y))));
                }
            }
            // ---- start search: lane m in (l, i-1] tests its start vector; default m = l ----
            bool start_ok = false;
            if (lane > l && lane <= i - 1) {
                const int m = lane;
                const zq h11 = H.at(m, m), h22 = H.at(m + 1, m + 1);
                zq h11s = qsub(h11, t);
                const double h21 = qabs1(H.at(m + 1, m));
                const double s = qabs1(h11s) + h21;
                h11s = qscale(h11s, 1.0 / s);
                const double h21n = h21 / s;
                const double h10 = qabs1(H.at(m, m - 1));
                start_ok = h10 * h21n <= ulp * (qabs1(h11s) * (qabs1(h11) + qabs1(h22)));
            }
            const unsigned long long mmask = __ballot(start_ok);
            const int m = uniform(mmask ? (63 - __builtin_clzll(mmask)) : l);
            zq a0, a1;
            {
                const zq h11s = qsub(H.at(m, m), t);
                const zq h21 = H.at(m + 1, m);
                const double s = 1.0 / (qabs1(h11s) + qabs1(h21));
                a0 = qscale(h11s, s);
                a1 = qscale(h21, s);
            }
            // ---- single-shift QR sweep, bulge forwarded through registers ----
            zq* rowp = &H.at(m, lc);
            zq* colp = &H.at(lc, m);
            for (int k = m; k < i; ++k) {
                const zq r0 = rowp[0];
                const zq r1 = rowp[SR];
                zq hkk1 = {0.0, 0.0};
                if (k == m && m > l) hkk1 = H.at(k, k - 1);
                double beta;
                zq v2, tau;
                zlarfg2_fast(a0, a1, beta, v2, tau);
                if (k > m) {
                    if (lane < 2) H.at(k + lane, k - 1) = {(lane == 0) ? beta : 0.0, 0.0};
                } else if (m > l) {
                    // sweep started below a small subdiagonal: H(m,m-1) <- (1 - conj(tau)) H(m,m-1)
                    if (lane == 0) H.at(k, k - 1) = qmul(hkk1, {1.0 - tau.re, tau.im});
                }
                // rows k, k+1 (lane = column): c <- H^H c,  sum = conj(tau) (a + conj(v2) b)
                const zq rs = qcmul(tau, qadd(r0, qcmul(v2, r1)));
                const zq nr0 = qsub(r0, rs);
                const zq nr1 = qsub(r1, qmul(rs, v2));
                if (lane >= k && lane <= i) {
                    rowp[0] = nr0;
                    rowp[SR] = nr1;
                }
                wave_lds_sync();
                // columns k, k+1 (lane = row), rows l..min(k+2, i):  r <- r H,  sum = tau (a + b v2)
                const int rmax = (k + 2 < i) ? k + 2 : i;
                const zq c0 = colp[0];
                const zq c1 = colp[SC];
                asm volatile("" ::"v"(nr0.re), "v"(nr0.im), "v"(nr1.re), "v"(nr1.im));
                const zq cs = qmul(tau, qadd(c0, qmul(c1, v2)));
                const zq m0 = qsub(c0, cs);
                if (lane >= l && lane <= rmax) {
                    colp[0] = m0;
                    colp[SC] = qsub(c1, qmulc(cs, v2));
                }
                a0 = qbcast(m0, k + 1);
                a1 = qbcast(m0, rmax);                    // row k+2 (unused after the last step)
                rowp += SR;
                colp += SC;
                wave_lds_sync();
            }
        }
        if (!converged) {
            info = i + 1;
            break;
        }
        {
            const zq e = H.at(i, i);
            if (lane == 0) {
                w_out[2 * i] = e.re;
                w_out[2 * i + 1] = e.im;
            }
        }
        kdefl = 0;
        i = l - 1;
        wave_lds_sync();
    }
    if (info != 0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int j = lane; j < info; j += kWave) {
            w_out[2 * j] = qnan;
            w_out[2 * j + 1] = qnan;
        }
    }
    if (lane == 0) *info_out = info;
}

template <class V>
__device__ __forceinline__ void zhqr_run(const V H, const double* __restrict__ src, int n, int lane,
                                         double* __restrict__ wo, int32_t* __restrict__ info) {
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
    zhqr_wave(H, n, lane, wo, info);
}

// info[mat] may arrive pre-set to n by the Hessenberg stage (non-finite input): such matrices are skipped.
__global__ __launch_bounds__(256) void zhqr_wave_kernel(const double* __restrict__ hin, int64_t nmat, int n,
                                                        double* __restrict__ w, int32_t* __restrict__ info,
                                                        int waves_per_block) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int64_t mat = (int64_t)blockIdx.x * waves_per_block + wave;
    if (mat >= nmat) return;
    if (info[mat] != 0) return;
    const int ldp = zhqr_pair_ld(n);
    zq* base = reinterpret_cast<zq*>(smem) + (size_t)(wave >> 1) * ((size_t)n * ldp);
    const double* src = hin + mat * 2 * (int64_t)n * n;
    double* wo = w + mat * 2 * (int64_t)n;
    if (n == 64) {
        if ((wave & 1) == 0) zhqr_run(ZViewT<69, false>{base + 5, ldp}, src, n, lane, wo, info + mat);
        else zhqr_run(ZViewT<69, true>{base, ldp}, src, n, lane, wo, info + mat);
    } else {
        if ((wave & 1) == 0) zhqr_run(ZViewT<0, false>{base + 5, ldp}, src, n, lane, wo, info + mat);
        else zhqr_run(ZViewT<0, true>{base, ldp}, src, n, lane, wo, info + mat);
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
