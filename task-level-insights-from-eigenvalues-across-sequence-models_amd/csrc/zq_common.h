// Complex128 helpers and the wave-per-matrix single-shift QR iteration shared by the complex eigenvalue kernels
// (eig_cplx_lds.hip: n <= 64, one wavefront per matrix; eig_zlarge.hip: shift generator and small deflated blocks).
#pragma once

#include "egs_common.h"

namespace egs {

// 16-byte aligned: one ds_read_b128 / global_load_dwordx4 per element (the 8-byte aligned form costs the LDS pipe twice
// the cycles per load).  Every zq* in the kernels points into LDS declared aligned(16), the 256-byte aligned workspace,
// or the complex128 output.
struct alignas(16) zq {
    double re, im;
};
// select BY VALUE: a ternary on the struct becomes a select between two addresses, which the compiler routes through
// scratch memory (a round trip to HBM inside the hottest loops)
__device__ __forceinline__ zq qsel(bool c, zq a, zq b) { return {c ? a.re : b.re, c ? a.im : b.im}; }
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

// sqrt of a complex number on the hardware reciprocal-square-root seed (no IEEE division / square root / hypot):
// good to a few ulp, for shifts and other quantities that only steer the iteration
__device__ __forceinline__ zq qsqrt_fast(zq a) {
    const double m2 = fma(a.re, a.re, a.im * a.im);
    if (m2 < 1e-300) return {0.0, 0.0};
    const double m = m2 * fast_rsqrt(m2);
    const double t = 0.5 * (m + fabs(a.re));
    const double rs = fast_rsqrt(t);                 // t > 0 because m > 0
    const double sr = t * rs;
    const double si = 0.5 * a.im * rs;
    if (a.re >= 0.0) return {sr, si};
    return {fabs(si), copysign(sr, a.im)};
}
__device__ __forceinline__ zq qdiv_fast(zq a, zq b) {
    const double r = fast_rcp(fma(b.re, b.re, b.im * b.im));
    return {fma(a.re, b.re, a.im * b.im) * r, fma(a.im, b.re, -a.re * b.im) * r};
}

// SHIFTS = true: the eigenvalues are only going to be used as shifts of a multishift sweep -- deflation at 1e-11
// (one iteration fewer per eigenvalue), no Ahues-Tisseur second test, no search for a start below a small subdiagonal,
// seed-based arithmetic in the Wilkinson shift.  SHIFTS = false is the xLAHQR iteration that produces final spectra.
template <class V, bool SHIFTS = false>
__device__ void zhqr_wave(const V H, const int n, const int lane, double* __restrict__ w_out,
                          int32_t* __restrict__ info_out, const double rescale = 1.0) {
    const double ulp = SHIFTS ? 1e-11 : 2.220446049250313e-16;   // (1e-7 / 1e-5 measured in round 4: more sweeps, C5 +2-3 ms)
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
                    if (SHIFTS) {
                        small = hkk1 <= ulp * tst;
                    } else if (hkk1 <= ulp * tst) {
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
            } else if (SHIFTS) {
                t = H.at(i, i);
                const zq u2 = qmul(H.at(i - 1, i), H.at(i, i - 1));   // u^2 (the branch of u does not matter below)
                if (qabs1(u2) != 0.0) {
                    const zq x = qscale(qsub(H.at(i - 1, i - 1), t), 0.5);
                    zq y = qsqrt_fast(qadd(qmul(x, x), u2));
                    if (x.re * y.re + x.im * y.im < 0.0) y = {-y.re, -y.im};
                    t = qsub(t, qdiv_fast(u2, qadd(x, y)));
                }
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
            if (!SHIFTS && lane > l && lane <= i - 1) {
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
                const double ssum = qabs1(h11s) + qabs1(h21);
                const double s = SHIFTS ? fast_rcp(fmax(ssum, 1e-300)) : 1.0 / ssum;
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
                w_out[2 * i] = e.re * rescale;
                w_out[2 * i + 1] = e.im * rescale;
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

}  // namespace egs
