// K5 (complex, and real n > 64): batched eigenvalues of dense complex128 matrices, ONE WORKGROUP PER MATRIX.
//
// This is the generalisation of the reference's only dense eigendecomposition -- np.linalg.eigvals(Ab) on the
// bilinear-discretised S4 DPLR matrix (analysis/eval_eig.py:296; N = state_dim in {64, 128, 512}) -- from one
// matrix per layer to (layers x feature columns) matrices.
//
//   phase 1: Householder reduction to Hessenberg form (unblocked xGEHD2 order: H <- Hk^H H Hk)
//   phase 2: single-shift QR with the xLAHQR control flow (Ahues-Tisseur deflation test on |re|+|im|
//            magnitudes, Wilkinson shift, exceptional shifts every 10 stalled iterations, start at two
//            consecutive small subdiagonals); 2-element reflectors; eigenvalues only (active window only).
// Orders n <= 64 take the wave-per-matrix kernels of eig_cplx_lds.hip instead (see eig_cplx_launch below).
// The working copy of the matrix lives in LDS when it fits (n <= 96), otherwise in the caller's workspace
// (L2 / Infinity-Cache resident: 1 MB per 256 x 256 matrix).  Subdiagonals are NOT kept real (LAPACK rescales
// to keep them real; the general-complex reflector formulas below do not need that).
#include <stdlib.h>

#include "egs_common.h"

namespace egs {

struct zc {
    double re, im;
};
__device__ __forceinline__ zc zmk(double r, double i) { return {r, i}; }
__device__ __forceinline__ zc zadd(zc a, zc b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ zc zsub(zc a, zc b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ zc zmul(zc a, zc b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ zc zmulc(zc a, zc b) {   // a * conj(b)
    return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im};
}
__device__ __forceinline__ zc zscale(zc a, double s) { return {a.re * s, a.im * s}; }
__device__ __forceinline__ zc zconj(zc a) { return {a.re, -a.im}; }
__device__ __forceinline__ double zabs1(zc a) { return fabs(a.re) + fabs(a.im); }
__device__ __forceinline__ double zabs(zc a) { return hypot(a.re, a.im); }
__device__ __forceinline__ zc zdiv(zc a, zc b) {   // Smith's algorithm (xLADIV spirit)
    if (fabs(b.re) >= fabs(b.im)) {
        const double r = b.im / b.re, d = b.re + b.im * r;
        return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
    }
    const double r = b.re / b.im, d = b.im + b.re * r;
    return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
__device__ __forceinline__ zc zsqrt(zc a) {
    const double m = zabs(a);
    if (m == 0.0) return {0.0, 0.0};
    double sr = sqrt(0.5 * (m + fabs(a.re)));
    double si = 0.5 * a.im / sr;
    if (a.re >= 0.0) return {sr, si};
    return {fabs(si), copysign(sr, a.im)};
}

constexpr int kCThreads = 256;

// 2-element reflector (xLARFG, n = 2): on return alpha := beta (real), x := v2, tau
__device__ __forceinline__ void zlarfg2(zc& alpha, zc& x, zc& tau) {
    const double xnorm = zabs(x);
    if (xnorm == 0.0 && alpha.im == 0.0) {
        tau = {0.0, 0.0};
        return;
    }
    const double beta = -copysign(sqrt(alpha.re * alpha.re + alpha.im * alpha.im + xnorm * xnorm), alpha.re);
    tau = {(beta - alpha.re) / beta, -alpha.im / beta};
    const zc sc = zdiv({1.0, 0.0}, {alpha.re - beta, alpha.im});
    x = zmul(x, sc);
    alpha = {beta, 0.0};
}

template <typename T>
__device__ __forceinline__ T block_reduce_sum(T v, T* red) {
    // red: LDS scratch of kCThreads/64 entries
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    T tot = red[0];
    for (int w = 1; w < kCThreads / 64; ++w) tot += red[w];
    return tot;
}

// INREAL: the input matrices are real fp64 (converted on load).  LDS: working copy in dynamic shared memory.
template <bool INREAL, bool LDS>
__global__ __launch_bounds__(kCThreads) void eig_cplx_kernel(const double* __restrict__ a, int n, int lda,
                                                             int64_t stride, double* __restrict__ w,
                                                             int32_t* __restrict__ info_out,
                                                             double* __restrict__ work, int balance) {
    extern __shared__ double dyn[];
    __shared__ double red[kCThreads / 64];
    __shared__ int s_idx;
    const int tid = threadIdx.x;
    const int64_t mat = blockIdx.x;
    const int ldh = LDS ? (n + 1) : n;
    zc* H = LDS ? reinterpret_cast<zc*>(dyn) : reinterpret_cast<zc*>(work) + mat * (int64_t)n * n;
    zc* vsh = LDS ? reinterpret_cast<zc*>(dyn) + (size_t)n * ldh : reinterpret_cast<zc*>(dyn);   // n entries
#define HM(i, j) H[(int64_t)(i) * ldh + (j)]

    // ---- load ----
    if (tid == 0) s_idx = 0;
    __syncthreads();
    {
        const double* src = a + mat * stride * (INREAL ? 1 : 2);
        for (int idx = tid; idx < n * n; idx += kCThreads) {
            const int i = idx / n, j = idx - i * n;
            zc v;
            if (INREAL) v = {src[(int64_t)i * lda + j], 0.0};
            else v = {src[2 * ((int64_t)i * lda + j)], src[2 * ((int64_t)i * lda + j) + 1]};
            HM(i, j) = v;
            if (!(fabs(v.re) <= 1.7976931348623157e308) || !(fabs(v.im) <= 1.7976931348623157e308)) s_idx = 1;
        }
    }
    __syncthreads();
    if (s_idx != 0) {                    // NaN / Inf input: an error like numpy.linalg.eigvals, not a spectrum
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        double* wbad = w + mat * 2 * (int64_t)n;
        for (int j = tid; j < 2 * n; j += kCThreads) wbad[j] = qnan;
        if (tid == 0) info_out[mat] = n;
        return;
    }

    // ---- phase 0: xGEBAL (what xGEEV does before the reduction) ----
    // Permutation: rows / columns whose off-diagonal part inside the active window is zero isolate an eigenvalue and
    // are swapped to the ends (their exact zeros survive the reflectors below, so the QR phase deflates them at once,
    // bit-exactly).  Candidate search keeps per-row / per-column nonzero counts up to date, O(n) per isolated
    // eigenvalue.  Scaling: Jacobi-form Parlett-Reinsch passes with power-of-two factors on the window [k0, l0].
    if (balance) {
        double* cn = reinterpret_cast<double*>(vsh + n);
        double* rn = cn + n;
        double* fs = rn + n;
        int* cnt = reinterpret_cast<int*>(fs + n);
        const int lane = tid & 63, wv = tid >> 6;
#define ZNZ(i, j) (HM(i, j).re != 0.0 || HM(i, j).im != 0.0)
#define ZSWAP(x, y)        \
    {                      \
        const zc t_ = (x); \
        (x) = (y);         \
        (y) = t_;          \
    }
        int k0 = 0, l0 = n - 1;
        for (int i = wv; i < n; i += kCThreads / 64) {              // wave per row, lanes over columns
            double c = 0.0;
            for (int j = lane; j < n; j += 64) c += (j != i && ZNZ(i, j)) ? 1.0 : 0.0;
            c = wave_sum(c);
            if (lane == 0) cnt[i] = (int)c;
        }
        __syncthreads();
        while (l0 > 0) {
            if (tid == 0) s_idx = -1;
            __syncthreads();
            for (int i = tid; i <= l0; i += kCThreads)
                if (cnt[i] == 0) atomicMax(&s_idx, i);
            __syncthreads();
            const int r = s_idx;
            __syncthreads();
            if (r < 0) break;
            if (r != l0) {
                for (int t = tid; t <= l0; t += kCThreads) ZSWAP(HM(t, r), HM(t, l0));
                __syncthreads();
                for (int t = tid; t < n; t += kCThreads) ZSWAP(HM(r, t), HM(l0, t));
                if (tid == 0) {
                    const int c = cnt[r];
                    cnt[r] = cnt[l0];
                    cnt[l0] = c;
                }
                __syncthreads();
            }
            for (int t = tid; t < l0; t += kCThreads)                // column l0 leaves the window
                if (ZNZ(t, l0)) cnt[t] -= 1;
            --l0;
            __syncthreads();
        }
        for (int j = tid; j <= l0; j += kCThreads) {                // thread per column
            int c = 0;
            for (int i = 0; i <= l0; ++i) c += (i != j && ZNZ(i, j)) ? 1 : 0;
            cnt[j] = c;
        }
        __syncthreads();
        while (k0 < l0) {
            if (tid == 0) s_idx = n;
            __syncthreads();
            for (int j = k0 + tid; j <= l0; j += kCThreads)
                if (cnt[j] == 0) atomicMin(&s_idx, j);
            __syncthreads();
            const int c = s_idx;
            __syncthreads();
            if (c >= n) break;
            if (c != k0) {
                for (int t = tid; t <= l0; t += kCThreads) ZSWAP(HM(t, c), HM(t, k0));
                __syncthreads();
                for (int t = k0 + tid; t < n; t += kCThreads) ZSWAP(HM(c, t), HM(k0, t));
                if (tid == 0) {
                    const int q = cnt[c];
                    cnt[c] = cnt[k0];
                    cnt[k0] = q;
                }
                __syncthreads();
            }
            for (int t = k0 + 1 + tid; t <= l0; t += kCThreads)     // row k0 leaves the window
                if (ZNZ(k0, t)) cnt[t] -= 1;
            ++k0;
            __syncthreads();
        }
        for (int pass = 0; pass < 16; ++pass) {
            for (int j = k0 + tid; j <= l0; j += kCThreads) {
                double c = 0.0;
                for (int i = k0; i <= l0; ++i)
                    if (i != j) c += zabs1(HM(i, j));
                cn[j] = c;
            }
            for (int i = k0 + wv; i <= l0; i += kCThreads / 64) {
                double r = 0.0;
                for (int j = k0 + lane; j <= l0; j += 64)
                    if (j != i) r += zabs1(HM(i, j));
                r = wave_sum(r);
                if (lane == 0) rn[i] = r;
            }
            if (tid == 0) s_idx = 0;
            __syncthreads();
            for (int i = tid; i < n; i += kCThreads) {
                double f = 1.0;
                if (i >= k0 && i <= l0) {
                    double c = cn[i];
                    const double r = rn[i];
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
                fs[i] = f;
                cn[i] = 1.0 / f;                                      // reuse: reciprocal factors (exact, powers of two)
                if (f != 1.0) s_idx = 1;
            }
            __syncthreads();
            const bool any = s_idx != 0;
            __syncthreads();
            if (!any) break;
            for (int idx = tid; idx < n * n; idx += kCThreads) {
                const int i = idx / n, j = idx - i * n;
                HM(i, j) = zscale(HM(i, j), fs[j] * cn[i]);
            }
            __syncthreads();
        }
        if (tid == 0) s_idx = 0;
        __syncthreads();
#undef ZNZ
#undef ZSWAP
    }

    // ---- phase 1: Hessenberg reduction ----
    for (int k = 0; k < n - 2; ++k) {
        double part = 0.0;
        for (int i = k + 2 + tid; i < n; i += kCThreads) {
            const zc x = HM(i, k);
            part += x.re * x.re + x.im * x.im;
        }
        const double xnorm2 = block_reduce_sum(part, red);
        zc alpha = HM(k + 1, k);
        if (xnorm2 == 0.0 && alpha.im == 0.0) continue;          // block-uniform
        const double beta = -copysign(sqrt(alpha.re * alpha.re + alpha.im * alpha.im + xnorm2), alpha.re);
        const zc tau = {(beta - alpha.re) / beta, -alpha.im / beta};
        const zc sc = zdiv({1.0, 0.0}, {alpha.re - beta, alpha.im});
        __syncthreads();                                          // everyone has read alpha
        for (int i = k + 1 + tid; i < n; i += kCThreads) {
            if (i == k + 1) {
                vsh[i] = {1.0, 0.0};
                HM(i, k) = {beta, 0.0};
            } else {
                vsh[i] = zmul(HM(i, k), sc);
                HM(i, k) = {0.0, 0.0};
            }
        }
        __syncthreads();
        // right: A(0:n, k+1:n) -= tau (A v) v^H      one wave per row, lanes over columns (coalesced)
        {
            const int lane = tid & 63, wv = tid >> 6;
            for (int i = wv; i < n; i += kCThreads / 64) {
                zc acc = {0.0, 0.0};
                for (int j = k + 1 + lane; j < n; j += 64) acc = zadd(acc, zmul(HM(i, j), vsh[j]));
                acc.re = wave_sum(acc.re);
                acc.im = wave_sum(acc.im);
                const zc tw = zmul(tau, acc);
                for (int j = k + 1 + lane; j < n; j += 64) HM(i, j) = zsub(HM(i, j), zmulc(tw, vsh[j]));
            }
        }
        __syncthreads();
        // left: A(k+1:n, k+1:n) -= conj(tau) v (v^H A)   thread per column (coalesced across threads)
        for (int j = k + 1 + tid; j < n; j += kCThreads) {
            zc acc = {0.0, 0.0};
            for (int i = k + 1; i < n; ++i) acc = zadd(acc, zmul(zconj(vsh[i]), HM(i, j)));
            const zc tw = zmul(zconj(tau), acc);
            for (int i = k + 1; i < n; ++i) HM(i, j) = zsub(HM(i, j), zmul(vsh[i], tw));
        }
        __syncthreads();
    }

    // ---- phase 2: single-shift QR, eigenvalues only ----
    const double ulp = 2.220446049250313e-16;
    const double safmin = 2.2250738585072014e-308;
    const double smlnum = safmin * ((double)n / ulp);
    const int itmax = 30 * (n > 10 ? n : 10);
    const int kexsh = 10;
    double* wout = w + mat * 2 * (int64_t)n;
    int info = 0;
    int i = n - 1;
    int kdefl = 0;
    while (i >= 0) {
        int l = 0;
        bool converged = false;
        for (int its = 0; its <= itmax; ++its) {
            // small-subdiagonal search: highest k in (l, i] passing the test
            if (tid == 0) s_idx = l;
            __syncthreads();
            for (int k = l + 1 + tid; k <= i; k += kCThreads) {
                bool small = false;
                const double hkk1 = zabs1(HM(k, k - 1));
                if (hkk1 <= smlnum) {
                    small = true;
                } else {
                    const zc hkk = HM(k, k), hk1k1 = HM(k - 1, k - 1);
                    double tst = zabs1(hk1k1) + zabs1(hkk);
                    if (tst == 0.0) {
                        if (k - 2 >= 0) tst += zabs1(HM(k - 1, k - 2));
                        if (k + 1 <= n - 1) tst += zabs1(HM(k + 1, k));
                    }
                    if (hkk1 <= ulp * tst) {
                        const double hk1k = zabs1(HM(k - 1, k));
                        const double ab = fmax(hkk1, hk1k), ba = fmin(hkk1, hk1k);
                        const double df = zabs1(zsub(hk1k1, hkk));
                        const double aa = fmax(zabs1(hkk), df), bb = fmin(zabs1(hkk), df);
                        const double s = aa + ab;
                        if (ba * (ab / s) <= fmax(smlnum, ulp * (bb * (aa / s)))) small = true;
                    }
                }
                if (small) atomicMax(&s_idx, k);
            }
            __syncthreads();
            l = s_idx;
            if (l > 0 && tid == 0) HM(l, l - 1) = {0.0, 0.0};
            if (l >= i) {
                converged = true;
                break;
            }
            ++kdefl;
            // shift (block-uniform, every thread computes it from the same elements)
            zc t;
            if (kdefl % (2 * kexsh) == 0) {
                t = zadd({0.75 * zabs1(HM(i, i - 1)), 0.0}, HM(i, i));
            } else if (kdefl % kexsh == 0) {
                t = zadd({0.75 * zabs1(HM(l + 1, l)), 0.0}, HM(l, l));
            } else {
                t = HM(i, i);
                const zc u = zmul(zsqrt(HM(i - 1, i)), zsqrt(HM(i, i - 1)));
                double s = zabs1(u);
                if (s != 0.0) {
                    const zc x = zscale(zsub(HM(i - 1, i - 1), t), 0.5);
                    const double sx = zabs1(x);
                    s = fmax(s, sx);
                    const zc xs = zscale(x, 1.0 / s), us = zscale(u, 1.0 / s);
                    zc y = zscale(zsqrt(zadd(zmul(xs, xs), zmul(us, us))), s);
                    if (sx > 0.0) {
                        const zc xn = zscale(x, 1.0 / sx);
                        if (xn.re * y.re + xn.im * y.im < 0.0) y = {-y.re, -y.im};
                    }
                    t = zsub(t, zmul(u, zdiv(u, zadd(x, y))));
                }
            }
            __syncthreads();      // the H(l,l-1) = 0 store above is ordered before the sweep's accesses
            // start search: highest m in (l, i-1] whose start vector passes; default m = l
            if (tid == 0) s_idx = l;
            __syncthreads();
            for (int m = l + 1 + tid; m <= i - 1; m += kCThreads) {
                const zc h11 = HM(m, m), h22 = HM(m + 1, m + 1);
                zc h11s = zsub(h11, t);
                const double h21 = zabs1(HM(m + 1, m));
                const double s = zabs1(h11s) + h21;
                h11s = zscale(h11s, 1.0 / s);
                const double h21n = h21 / s;
                const double h10 = zabs1(HM(m, m - 1));
                if (h10 * h21n <= ulp * (zabs1(h11s) * (zabs1(h11) + zabs1(h22)))) atomicMax(&s_idx, m);
            }
            __syncthreads();
            const int m = s_idx;
            zc v0, v1;
            {
                zc h11s = zsub(HM(m, m), t);
                zc h21 = HM(m + 1, m);
                const double s = zabs1(h11s) + zabs1(h21);
                v0 = zscale(h11s, 1.0 / s);
                v1 = zscale(h21, 1.0 / s);
            }
            __syncthreads();
            // single-shift QR sweep
            for (int k = m; k < i; ++k) {
                zc a0, a1;
                if (k > m) {
                    a0 = HM(k, k - 1);
                    a1 = HM(k + 1, k - 1);
                } else {
                    a0 = v0;
                    a1 = v1;
                }
                zc tau;
                zlarfg2(a0, a1, tau);           // a0 := beta, a1 := v2
                const zc v2 = a1;
                // rows k, k+1, columns k..i:  c <- H^H c,  sum = conj(tau) (a + conj(v2) b)
                for (int j = k + tid; j <= i; j += kCThreads) {
                    const zc ra = HM(k, j), rb = HM(k + 1, j);
                    const zc sum = zmul(zconj(tau), zadd(ra, zmul(zconj(v2), rb)));
                    HM(k, j) = zsub(ra, sum);
                    HM(k + 1, j) = zsub(rb, zmul(sum, v2));
                }
                __syncthreads();                // also: every thread has read the bulge column by now
                if (tid == 0) {
                    if (k > m) {
                        HM(k, k - 1) = a0;
                        HM(k + 1, k - 1) = {0.0, 0.0};
                    } else if (m > l) {
                        // sweep started below a small (not yet negligible) subdiagonal: row m of column m-1 sees
                        // the reflector too, H(m,m-1) <- (1 - conj(tau)) H(m,m-1); the fill-in below it is the
                        // product of two small subdiagonals and is dropped (the xLAHQR start criterion)
                        HM(k, k - 1) = zmul(HM(k, k - 1), {1.0 - tau.re, tau.im});
                    }
                }
                // columns k, k+1, rows l..min(k+2,i):  r <- r H,  sum = tau (a + b v2)
                const int rmax = (k + 2 < i) ? k + 2 : i;
                for (int r = l + tid; r <= rmax; r += kCThreads) {
                    const zc ca = HM(r, k), cb = HM(r, k + 1);
                    const zc sum = zmul(tau, zadd(ca, zmul(cb, v2)));
                    HM(r, k) = zsub(ca, sum);
                    HM(r, k + 1) = zsub(cb, zmulc(sum, v2));
                }
                __syncthreads();
            }
        }
        if (!converged) {
            info = i + 1;
            break;
        }
        if (tid == 0) {
            const zc e = HM(i, i);
            wout[2 * i] = e.re;
            wout[2 * i + 1] = e.im;
        }
        kdefl = 0;
        i = l - 1;
        __syncthreads();
    }
    if (info != 0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int j = tid; j < info; j += kCThreads) {
            wout[2 * j] = qnan;
            wout[2 * j + 1] = qnan;
        }
    }
    if (tid == 0) info_out[mat] = info;
#undef HM
}

constexpr int kCplxLdsMaxN = 96;
constexpr int kCplxWaveMaxN = 64;     // wave-per-matrix QR (eig_cplx_lds.hip) below this order

int zhqr_wave_launch(const double* h, int64_t nmat, int n, double* w, int32_t* info, hipStream_t stream);
int zhess64_launch(const void* a, bool in_real, int64_t nmat, int n, int lda, int64_t stride, double* hout, double* w,
                   int32_t* info, hipStream_t stream);

size_t eig_cplx_workspace_bytes(int64_t nmat, int n) {
    return (size_t)nmat * n * n * 2 * sizeof(double);          // working copy (n > 64: either path may be chosen)
}

int eig_zlarge_launch(const void* a, bool in_real, int64_t nmat, int n, int lda, int64_t stride, double* w,
                      int32_t* info, void* workspace, size_t workspace_bytes, hipStream_t stream);

int eig_cplx_launch(const void* a, bool in_real, int64_t nmat, int n, int lda, int64_t stride, double* w,
                    int32_t* info, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (nmat == 0 || n == 0) return 0;
    // n > 64: level-3 path (eig_zlarge.hip); EGS_ZLARGE_OFF / EGS_ZLARGE_MIN keep the unblocked workgroup kernel below for
    // A/B runs
    // every order above the wave-per-matrix kernels takes the level-3 path: measured 2.6-2.8x faster than the LDS-resident
    // unblocked kernel below even at n = 72 ... 96 (768 matrices: 26.8 / 33.0 / 46.2 ms -> 10.3 / 12.3 / 16.5 ms)
    int zl_min = kCplxWaveMaxN + 1;
    if (const char* e = getenv("EGS_ZLARGE_MIN")) {       // tuning knob: smallest order that takes the level-3 path (> 64)
        const int v = atoi(e);
        if (v > kCplxWaveMaxN) zl_min = v;
    }
    if (n >= zl_min && n <= 1024 && !getenv("EGS_ZLARGE_OFF"))
        return eig_zlarge_launch(a, in_real, nmat, n, lda, stride, w, info, workspace, workspace_bytes, stream);
    if (n > 1024) {
        set_error("n = %d exceeds the supported maximum of 1024", n);
        return -4;
    }
    const bool lds = n <= kCplxLdsMaxN;
    const bool wave_qr = n <= kCplxWaveMaxN;
    if ((!lds || wave_qr) && workspace_bytes < eig_cplx_workspace_bytes(nmat, n)) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, eig_cplx_workspace_bytes(nmat, n));
        return -9;
    }
    double* wk = static_cast<double*>(workspace);
    if (wave_qr) {        // n <= 64: workgroup-per-matrix Hessenberg + wave-per-matrix QR (eig_cplx_lds.hip)
        const int st = zhess64_launch(a, in_real, nmat, n, lda, stride, wk, w, info, stream);
        if (st) return st;
        return zhqr_wave_launch(wk, nmat, n, w, info, stream);
    }
    // working copy (LDS path) + Householder vector (n complex) + balancing scratch (4 n doubles)
    const size_t dyn = (lds ? ((size_t)n * (n + 1) + n) * 2 : (size_t)n * 2) * sizeof(double) + (size_t)n * 4 * sizeof(double);
    const int balance = getenv("EGS_NO_BALANCE") ? 0 : 1;
    const double* ap = static_cast<const double*>(a);
    auto launch = [&](auto kern) -> int {
        if (dyn > 48 * 1024)
            EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(kern, dim3((unsigned)nmat), dim3(kCThreads), dyn, stream, ap, n, lda, stride, w, info, wk,
                           balance);
        EGS_LAUNCH_CHECK();
        return 0;
    };
    if (in_real) return lds ? launch(eig_cplx_kernel<true, true>) : launch(eig_cplx_kernel<true, false>);
    return lds ? launch(eig_cplx_kernel<false, true>) : launch(eig_cplx_kernel<false, false>);
}

}  // namespace egs
