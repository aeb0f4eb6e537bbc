// K5 (real, n <= 64): batched eigenvalues of dense real fp64 matrices, LDS-resident, ONE WAVEFRONT PER
// MATRIX (lane = column index for row operations, lane = row index for column operations; both access
// patterns are bank-conflict-free with an odd leading dimension).
//
//   hess_real_kernel : Householder reduction to upper-Hessenberg form, global A -> global H workspace
//   hqr_real_kernel  : Francis double-shift QR on the Hessenberg matrix (LAPACK xLAHQR control flow:
//                      Ahues-Tisseur deflation test, Wilkinson double shift, exceptional shifts every 10
//                      stalled iterations, start at two consecutive small subdiagonals), eigenvalues only.
//
// Replaces np.linalg.eigvals at analysis/eval_eig.py:296 for real batches (BASELINE config C2).
// The QR kernel packs TWO Hessenberg matrices into one (n) x (n+7) LDS array -- one stored normally in
// columns >= row+4, the other transposed in columns <= row+3 -- so 8 matrices fit in 145 KB of a CU's LDS.
#include <stdlib.h>

#include "egs_common.h"
#include "hess_reg.h"

namespace egs {

// Element (i,j) of the wave's matrix lives at hm[i*sr + j*sc].  LDP > 0 fixes the pair-array leading dimension at
// compile time (n = 64 -> 71) so that row / column neighbours become DS-instruction immediate offsets; LDP == 0 keeps
// it in a register (any n <= 64).  TR selects the transposed member of a packed pair.  N = rows of the array (the fast
// layouts clamp never-used lane indices to N - 1 so that every address stays inside the wave's own array).
template <int LDP, bool TR, int N = 64>
struct MatViewT {
    static constexpr bool kFast = LDP > 0;
    static constexpr int kN = N;
    static constexpr int kSR = TR ? 1 : LDP, kSC = TR ? LDP : 1;   // compile-time strides of the fast layouts (LDP > 0)
    double* hm;
    int ld;
    __device__ __forceinline__ int sr() const { return TR ? 1 : (LDP > 0 ? LDP : ld); }
    __device__ __forceinline__ int sc() const { return TR ? (LDP > 0 ? LDP : ld) : 1; }
    __device__ __forceinline__ double& at(int i, int j) const { return hm[i * sr() + j * sc()]; }
};

// ---------------------------------------------------------------------------------------------------
// Hessenberg reduction (unblocked, xGEHD2-style), one wave per matrix, full n x ld layout in LDS.
// ---------------------------------------------------------------------------------------------------
constexpr int kHessLd = 65;            // odd leading dimension: row and column sweeps conflict-free
constexpr int kNonFinite = -2147483647;   // scale-slot sentinel: the input matrix holds NaN / Inf
#ifndef EGS_LD64
#define EGS_LD64 73
#endif
// pair-array leading dimension at n = 64 (>= n + 7, odd).  71 is the minimum; with 73 the DIAGONAL stride (ld + 1 = 74
// doubles = 148 dwords, 20 mod 64) spreads the per-sweep diagonal reads of the deflation / start searches over 16 banks
// instead of 4 (72 doubles = 144 dwords, 16 mod 64: 8-way conflicts), rows and columns stay conflict-free (146 dwords).
constexpr int kLd64 = EGS_LD64;
constexpr int kHessWaves = 4;          // matrices per workgroup (4 x 33,792 B = 135 KB LDS)
constexpr int kHessStride = kWave * kHessLd + 2 * kWave;   // matrix + Householder vector v + update vector y, in doubles
// the Householder entry v_i is wave-uniform: either a broadcast LDS read or two v_readlane of the lane-held copy
#ifdef EGS_HESS_V_READLANE
#define HESS_V(idx) bcast_lane(v, (idx))
#else
#define HESS_V(idx) vs[(idx)]
#endif

__global__ __launch_bounds__(kHessWaves* kWave) void hess_real_kernel(
    const double* __restrict__ a, int64_t nmat, int n, int lda, int64_t stride, double* __restrict__ hout,
    int32_t* __restrict__ scale_exp, int balance) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int64_t mat = (int64_t)blockIdx.x * kHessWaves + wave;
    if (mat >= nmat) return;                       // whole wave exits; no workgroup barriers are used
    double* s = smem + (size_t)wave * kHessStride;
    double* vs = s + kWave * kHessLd;              // Householder vector, broadcast-read by the update loops
    double* ys = vs + kWave;                       // y of the rank-2 update
    const double* src = a + mat * stride;
    const int lc = (lane < n) ? lane : (n - 1);    // clamped lane: loads always hit a valid element
    const bool inr = lane < n;

    // coalesced load: consecutive lanes read consecutive elements of a row; 8 rows in flight
    double amax = 0.0;
    bool finite = true;
    {
        int i = 0;
        for (; i + 8 <= n; i += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(i + u) * lda + lc];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (inr) s[(i + u) * kHessLd + lane] = t[u];
                amax = fmax(amax, fabs(t[u]));
                finite = finite && (fabs(t[u]) <= 1.7976931348623157e308);   // false for NaN and Inf
            }
        }
        for (; i < n; ++i) {
            const double t = src[(int64_t)i * lda + lc];
            if (inr) s[i * kHessLd + lane] = t;
            amax = fmax(amax, fabs(t));
            finite = finite && (fabs(t) <= 1.7976931348623157e308);
        }
    }
    wave_lds_sync();
    // xGEEV-style scaling: matrices whose largest entry is outside [1e-100, 1e100] are scaled by a power of two so
    // that the unscaled reflector arithmetic (squares of entries) cannot overflow / underflow; the QR stage multiplies
    // the eigenvalues back.  The exponent travels in the info slot between the two kernels.
    // Without an exponent output (egs_hessenberg_batched: the caller wants a matrix similar to its input) the same
    // scaling is applied and undone when the result is stored -- entries of 1e-220 otherwise gave a wrong reduction
    // (their squares underflow), entries of 1e+220 an overflow.
    int e_scale = 0;
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
        int e = 0;
        if (__ballot(!finite) != 0ULL) e = kNonFinite;           // NaN / Inf input: the QR stage reports failure
        else if (amax > 0.0 && (amax < 1e-100 || amax > 1e100)) e = ilogb(amax);
        if (e != 0 && e != kNonFinite) {
            // (scalbn per entry, not a multiplication by 2^-e: for denormal input 2^-e itself is out of range)
            for (int i = 0; i < n; ++i)
                if (inr) s[i * kHessLd + lane] = scalbn(s[i * kHessLd + lane], -e);
            wave_lds_sync();
        }
        if (scale_exp != nullptr) {
            if (lane == 0) scale_exp[mat] = e;
        } else if (e != kNonFinite) {
            e_scale = e;
        }
    }

    // ---- xGEBAL: permute rows/columns that isolate an eigenvalue to the ends, then scale the remaining block ----
    // (what numpy.linalg.eigvals -> xGEEV does before the reduction).  Eigenvalues are invariant; the permuted matrix is
    // block upper triangular with exact zeros that the reflectors below preserve (their tails are exactly zero
    // there), so the QR stage deflates those eigenvalues immediately and exactly; the power-of-two diagonal scaling
    // equilibrates row and column norms of the active block, which is what makes graded / badly scaled inputs accurate.
    if (balance) {
        int k0 = 0, l0 = n - 1;
        // one batched sweep over row `lane` and column `lane` (8 loads of each in flight): off-diagonal maxima decide
        // whether any row / column isolates an eigenvalue, the 1-norms are the first scaling pass of the common case
        // (no permutation).  A branch per element would serialise the LDS reads (see the lesson in DESIGN.md).
        double rmax = 0.0, cmax = 0.0, rsum = 0.0, csum = 0.0;
        {
            int j = 0;
            for (; j + 8 <= n; j += 8) {
                double xr[8], xc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    xr[u] = s[lc * kHessLd + j + u];
                    xc[u] = s[(j + u) * kHessLd + lc];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double ar = (j + u != lane) ? fabs(xr[u]) : 0.0, ac = (j + u != lane) ? fabs(xc[u]) : 0.0;
                    rmax = fmax(rmax, ar);
                    cmax = fmax(cmax, ac);
                    rsum += ar;
                    csum += ac;
                }
            }
            for (; j < n; ++j) {
                const double ar = (j != lane) ? fabs(s[lc * kHessLd + j]) : 0.0;
                const double ac = (j != lane) ? fabs(s[j * kHessLd + lc]) : 0.0;
                rmax = fmax(rmax, ar);
                cmax = fmax(cmax, ac);
                rsum += ar;
                csum += ac;
            }
        }
        const bool any_iso = __ballot(inr && (rmax == 0.0 || cmax == 0.0)) != 0ULL;
        if (any_iso && n > 1) {
            // ---- permutation (rare path): rows, then columns, one candidate per round ----
            for (int guard = 0; guard < n; ++guard) {               // rows whose off-diagonal part in columns 0..l0 is zero
                bool rz = false;
                if (lane <= l0) {
                    rz = true;
                    for (int j = 0; j <= l0; ++j)
                        if (j != lane && s[lane * kHessLd + j] != 0.0) rz = false;
                }
                const unsigned long long m = __ballot(rz);
                if (m == 0ULL || l0 == 0) break;
                const int r = uniform(63 - __builtin_clzll(m));
                if (r != l0) {
                    if (lane <= l0) {                               // swap columns r <-> l0 over rows 0..l0
                        const double t = s[lane * kHessLd + r];
                        s[lane * kHessLd + r] = s[lane * kHessLd + l0];
                        s[lane * kHessLd + l0] = t;
                    }
                    wave_lds_sync();
                    if (inr) {                                      // swap rows r <-> l0 over all columns
                        const double t = s[r * kHessLd + lane];
                        s[r * kHessLd + lane] = s[l0 * kHessLd + lane];
                        s[l0 * kHessLd + lane] = t;
                    }
                    wave_lds_sync();
                }
                --l0;
            }
            for (int guard = 0; guard < n; ++guard) {               // columns whose off-diagonal part in rows k0..l0 is zero
                bool cz = false;
                if (lane >= k0 && lane <= l0) {
                    cz = true;
                    for (int i = k0; i <= l0; ++i)
                        if (i != lane && s[i * kHessLd + lane] != 0.0) cz = false;
                }
                const unsigned long long m = __ballot(cz);
                if (m == 0ULL || k0 >= l0) break;
                const int c = uniform(__builtin_ctzll(m));
                if (c != k0) {
                    if (lane <= l0) {
                        const double t = s[lane * kHessLd + c];
                        s[lane * kHessLd + c] = s[lane * kHessLd + k0];
                        s[lane * kHessLd + k0] = t;
                    }
                    wave_lds_sync();
                    if (inr && lane >= k0) {
                        const double t = s[c * kHessLd + lane];
                        s[c * kHessLd + lane] = s[k0 * kHessLd + lane];
                        s[k0 * kHessLd + lane] = t;
                    }
                    wave_lds_sync();
                }
                ++k0;
            }
        }
        // scaling of the block [k0, l0]: all factors of a pass are computed from the same matrix (Jacobi form of the
        // Parlett-Reinsch iteration; powers of two, so the similarity is exact in floating point)
        const bool win = lane >= k0 && lane <= l0;
        for (int pass = 0; pass < 16; ++pass) {
            if (pass > 0 || any_iso) {                              // norms over the window, 8 loads of each in flight
                rsum = 0.0;
                csum = 0.0;
                int j = k0;
                for (; j + 8 <= l0 + 1; j += 8) {
                    double xr[8], xc[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        xr[u] = s[lc * kHessLd + j + u];
                        xc[u] = s[(j + u) * kHessLd + lc];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        rsum += (j + u != lane) ? fabs(xr[u]) : 0.0;
                        csum += (j + u != lane) ? fabs(xc[u]) : 0.0;
                    }
                }
                for (; j <= l0; ++j) {
                    rsum += (j != lane) ? fabs(s[lc * kHessLd + j]) : 0.0;
                    csum += (j != lane) ? fabs(s[j * kHessLd + lc]) : 0.0;
                }
            }
            double f = 1.0;
            if (win) {
                double c = csum;
                const double r = rsum;
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
            // A(r, c) <- A(r, c) * f_c / f_r   (lane = column c; 1 / f_r, exact for a power of two, broadcast from lane r)
            const double finv = 1.0 / f;
            {
                int r = 0;
                for (; r + 8 <= n; r += 8) {
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = s[(r + u) * kHessLd + lc];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double fri = bcast_lane(finv, r + u);
                        if (inr) s[(r + u) * kHessLd + lane] = (x[u] * f) * fri;
                    }
                }
                for (; r < n; ++r) {
                    const double fri = bcast_lane(finv, r);
                    if (inr) s[r * kHessLd + lane] = (s[r * kHessLd + lc] * f) * fri;
                }
            }
            wave_lds_sync();
        }
    }

    for (int k = 0; k < n - 2; ++k) {
        // lane i holds x_i = A(i,k), i > k
        const bool below = (lane > k) && inr;
        const double xi = below ? s[lane * kHessLd + k] : 0.0;
        const double alpha = bcast_lane(xi, k + 1);
        const double xnorm2 = wave_sum((lane > k + 1) ? xi * xi : 0.0);
        if (xnorm2 == 0.0) continue;               // wave-uniform: nothing to annihilate
        // squares in or below the denormal range: the norm would be inaccurate and the reflector not orthogonal; with the
        // scaling above such a column is below 1e-45 of the matrix and is left to the QR stage, which drops it
        if (alpha * alpha + xnorm2 < 1e-290) continue;
        const double nrm = sqrt(alpha * alpha + xnorm2);
        const double beta = -copysign(nrm, alpha);
        const double tau = (beta - alpha) / beta;
        const double scal = 1.0 / (alpha - beta);
        const double v = (lane == k + 1) ? 1.0 : ((lane > k + 1 && inr) ? xi * scal : 0.0);
        if (below) s[lane * kHessLd + k] = (lane == k + 1) ? beta : 0.0;
        vs[lane] = v;
        wave_lds_sync();

        // Two-sided update as ONE rank-2 pass:  H A H = A - w v^T - v y^T  with  p = A v,  q = A^T v,
        // a2 = v^T p,  w = tau p - (tau^2 a2 / 2) v,  y = tau q - (tau^2 a2 / 2) v.
        // Pass 1 (read-only) streams index t = k+1..n-1 once for both products: lane = row i reads A(i,t) for p_i,
        // lane = column j reads A(t,j) for q_j.  Pass 2 updates A(i, k+1:) with lane = row.
        double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
        {
            int t = k + 1;
            for (; t + 8 <= n; t += 8) {
                double ar[8], ac[8], vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ar[u] = s[lc * kHessLd + t + u];        // A(lane, t+u)
                    ac[u] = s[(t + u) * kHessLd + lc];      // A(t+u, lane)
                    vv[u] = vs[t + u];
                }
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    p0 = fma(ar[u], vv[u], p0);
                    q0 = fma(ac[u], vv[u], q0);
                    p1 = fma(ar[u + 1], vv[u + 1], p1);
                    q1 = fma(ac[u + 1], vv[u + 1], q1);
                }
            }
            for (; t < n; ++t) {
                const double vt = vs[t];
                p0 = fma(s[lc * kHessLd + t], vt, p0);
                q0 = fma(s[t * kHessLd + lc], vt, q0);
            }
        }
        const double pi = inr ? (p0 + p1) : 0.0;
        const double qj = below ? (q0 + q1) : 0.0;                 // only columns j > k are updated
        const double a2 = wave_sum(v * pi);
        const double corr = 0.5 * tau * tau * a2;
        const double wv = fma(tau, pi, -corr * v);                  // w_i, lane = row i
        ys[lane] = fma(tau, qj, -corr * v);                         // y_j, lane = column j
        wave_lds_sync();
        {
            int j = k + 1;
            for (; j + 8 <= n; j += 8) {
                double av[8], vv[8], yy[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av[u] = s[lc * kHessLd + j + u];
                    vv[u] = vs[j + u];
                    yy[u] = ys[j + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (inr) s[lane * kHessLd + j + u] = fma(-v, yy[u], fma(-wv, vv[u], av[u]));
            }
            for (; j < n; ++j)
                if (inr) s[lane * kHessLd + j] = fma(-v, ys[j], fma(-wv, vs[j], s[lane * kHessLd + j]));
        }
        wave_lds_sync();
    }

    // store the Hessenberg matrix densely (exact zeros below the subdiagonal)
    double* dst = hout + mat * (int64_t)n * n;
    for (int i = 0; i < n; ++i)
        if (inr) dst[(int64_t)i * n + lane] = (i > lane + 1) ? 0.0 : (e_scale ? scalbn(s[i * kHessLd + lane], e_scale)
                                                                               : s[i * kHessLd + lane]);
}

// ---------------------------------------------------------------------------------------------------
// 2x2 standardisation (xLANV2): eigenvalues of [[a,b],[c,d]]
// ---------------------------------------------------------------------------------------------------
__device__ inline void lanv2_eig(double a, double b, double c, double d, double& rt1r, double& rt1i,
                                 double& rt2r, double& rt2i) {
    const double eps = 2.220446049250313e-16;
    const double multpl = 4.0;
    if (c == 0.0) {
    } else if (b == 0.0) {
        const double t = d;
        d = a;
        a = t;
        b = -c;
        c = 0.0;
    } else if ((a - d) == 0.0 && ((b > 0.0) != (c > 0.0))) {
    } else {
        double temp = a - d;
        double p = 0.5 * temp;
        const double bcmax = fmax(fabs(b), fabs(c));
        const double bcmis = fmin(fabs(b), fabs(c)) * copysign(1.0, b) * copysign(1.0, c);
        const double scale = fmax(fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= multpl * eps) {
            z = p + copysign(sqrt(scale) * sqrt(z), p);
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            b = b - c;
            c = 0.0;
        } else {
            const double sigma = b + c;
            const double tau = hypot(sigma, temp);
            const double cs = sqrt(0.5 * (1.0 + fabs(sigma) / tau));
            const double sn = -(p / (tau * cs)) * copysign(1.0, sigma);
            const double aa = a * cs + b * sn;
            const double bb = -a * sn + b * cs;
            const double cc = c * cs + d * sn;
            const double dd = -c * sn + d * cs;
            a = aa * cs + cc * sn;
            b = bb * cs + dd * sn;
            c = -aa * sn + cc * cs;
            d = -bb * sn + dd * cs;
            temp = 0.5 * (a + d);
            a = temp;
            d = temp;
            if (c != 0.0) {
                if (b != 0.0) {
                    if ((b > 0.0) == (c > 0.0)) {
                        const double sab = sqrt(fabs(b));
                        const double sac = sqrt(fabs(c));
                        p = copysign(sab * sac, c);
                        a = temp + p;
                        d = temp - p;
                        b = b - c;
                        c = 0.0;
                    }
                } else {
                    b = -c;
                    c = 0.0;
                }
            }
        }
    }
    rt1r = a;
    rt2r = d;
    if (c == 0.0) {
        rt1i = 0.0;
        rt2i = 0.0;
    } else {
        rt1i = sqrt(fabs(b)) * sqrt(fabs(c));
        rt2i = -rt1i;
    }
}

// ---------------------------------------------------------------------------------------------------
// Francis double-shift QR, eigenvalues only, one wave per matrix.  All control flow is wave-uniform: the
// scalar quantities (shifts, reflectors, deflation decisions) are computed redundantly by every lane from
// broadcast LDS reads; the O(n) row / column updates of a bulge step are spread over the lanes.
// ---------------------------------------------------------------------------------------------------
// xLARFG on (a0; a1, a2), branch- and select-free.  A zero tail gives H = diag(-1, 1, 1) (t1 = 2, u = 0) instead of
// LAPACK's identity -- equally orthogonal, it only flips the sign of row/column k.  The two fmax guards keep an
// all-zero column finite (t1 = 0, u = 0).  rsqrt / rcp seeds + Newton steps keep the reflector orthogonal to ~1 ulp.
__device__ __forceinline__ void larfg3(double a0, double a1, double a2, double& beta, double& t1, double& u1,
                                       double& u2) {
    const double tiny = 1e-300;
    const double xs = fma(a1, a1, a2 * a2);
    const double nrm2 = fma(a0, a0, xs);
    const double rn = fast_rsqrt(fmax(nrm2, tiny));       // 1/||a||
    const double nrm = nrm2 * rn;
    const double apn = fabs(a0) + nrm;                    // |a0 - beta|
    const double scal = copysign(fast_rcp(fmax(apn, tiny)), a0);   // 1/(a0 - beta)
    beta = -copysign(nrm, a0);
    t1 = apn * rn;                                        // (beta - a0)/beta
    u1 = a1 * scal;
    u2 = a2 * scal;
}

// One order-3 bulge-chase step at row k (k <= i-2).  On entry (a0,a1,a2) is the bulge column H(k..k+2, k-1) (or the
// start vector when FIRST); on exit it is the next bulge column H(k+1..k+3, k), forwarded through registers so the
// critical path never waits for an LDS write -> read round trip.  rowp = &H(k, lane) and colp = &H(lane, k) advance
// by one row / column per step.  In the fast (n = 64) layout every lane owns a real column / row, loads need no
// clamping and the three row / column neighbours are immediate offsets; the generic layout clamps the lane index so
// that loads always hit valid elements.  Only the stores are predicated.
template <bool FIRST, class V>
__device__ __forceinline__ void qr_step3(const V& H, double*& rowp, double*& colp, int k, int l, int i, int lane,
                                         bool m_gt_l, double& a0, double& a1, double& a2) {
    const int SR = H.sr(), SC = H.sc();
    double r0, r1, r2;
    if constexpr (V::kFast) {
        r0 = rowp[0];
        r1 = rowp[SR];
        r2 = rowp[2 * SR];
    } else {
        const int jc = min(max(lane, k), i);              // row op: columns k..i
        r0 = H.at(k, jc);
        r1 = H.at(k + 1, jc);
        r2 = H.at(k + 2, jc);
    }
    double hkk1 = 0.0;
    if (FIRST) hkk1 = m_gt_l ? H.at(k, k - 1) : 0.0;
    double beta, t1, u1, u2;
    larfg3(a0, a1, a2, beta, t1, u1, u2);
    const double t2 = t1 * u1;
    const double t3 = t1 * u2;
    if (!FIRST) {
        if (lane < 3) H.at(k + lane, k - 1) = (lane == 0) ? beta : 0.0;
    } else if (m_gt_l) {
        if (lane == 0) H.at(k, k - 1) = hkk1 * (1.0 - t1);
    }
    const double rsum = fma(u2, r2, fma(u1, r1, r0));
    const double nr0 = fma(-rsum, t1, r0), nr1 = fma(-rsum, t2, r1), nr2 = fma(-rsum, t3, r2);
    if (lane >= k && lane <= i) {
        double* wp = V::kFast ? rowp : &H.at(k, lane);
        wp[0] = nr0;
        wp[SR] = nr1;
        wp[2 * SR] = nr2;
    }
    wave_lds_sync();
    const int rmax = (k + 3 < i) ? k + 3 : i;             // column op: rows l..min(k+3, i)
    double c0, c1, c2;
    if constexpr (V::kFast) {
        c0 = colp[0];
        c1 = colp[SC];
        c2 = colp[2 * SC];
        // keep the just-stored row values live across the column loads: otherwise the allocator lands the loads in
        // the stores' source registers and the compiler has to drain the LDS queue (s_waitcnt) in between
        asm volatile("" ::"v"(nr0), "v"(nr1), "v"(nr2));
    } else {
        const int rc = min(max(lane, l), rmax);
        c0 = H.at(rc, k);
        c1 = H.at(rc, k + 1);
        c2 = H.at(rc, k + 2);
    }
    const double sum = fma(u2, c2, fma(u1, c1, c0));
    const double n0 = fma(-sum, t1, c0);
    if (lane >= l && lane <= rmax) {
        double* wp = V::kFast ? colp : &H.at(lane, k);
        wp[0] = n0;
        wp[SC] = fma(-sum, t2, c1);
        wp[2 * SC] = fma(-sum, t3, c2);
    }
    a0 = bcast_lane(n0, k + 1);
    a1 = bcast_lane(n0, k + 2);
    a2 = bcast_lane(n0, rmax);                            // row k+3; unused by the order-2 step when k+3 > i
    rowp += SR;
    colp += SC;
    wave_lds_sync();
}

// The final order-2 step of a sweep at row k = i-1: reflector from (a0, a1) = H(i-1..i, i-2).
template <class V>
__device__ __forceinline__ void qr_step2(const V& H, int k, int l, int i, int lane, double a0, double a1) {
    const int jc = min(max(lane, k), i);
    const double r0 = H.at(k, jc);
    const double r1 = H.at(k + 1, jc);
    double beta, t1, u1, u2;
    larfg3(a0, a1, 0.0, beta, t1, u1, u2);
    const double t2 = t1 * u1;
    if (lane < 2) H.at(k + lane, k - 1) = (lane == 0) ? beta : 0.0;
    {
        const double sum = fma(u1, r1, r0);
        if (lane >= k && lane <= i) {
            H.at(k, lane) = fma(-sum, t1, r0);
            H.at(k + 1, lane) = fma(-sum, t2, r1);
        }
    }
    wave_lds_sync();
    const int rc = min(max(lane, l), i);                  // rows l..i
    const double c0 = H.at(rc, k);
    const double c1 = H.at(rc, k + 1);
    const double sum = fma(u1, c1, c0);
    if (lane >= l && lane <= i) {
        H.at(lane, k) = fma(-sum, t1, c0);
        H.at(lane, k + 1) = fma(-sum, t2, c1);
    }
    wave_lds_sync();
}


// ---------------------------------------------------------------------------------------------------
// Two bulges per sweep, one HALF-WAVE each (the n = 64 layouts).
//
// A bulge step is ~22 vector instructions of wave-uniform reflector arithmetic plus two O(n) passes that use the lanes;
// chasing ONE bulge per wavefront therefore spends most of its issue slots on work that 63 lanes repeat.  Here a sweep
// chases TWO double-shift bulges, the second kBulgeDist = 3 rows behind the first, and the same instructions serve both:
// lanes 0-31 (group 0) work for the leading bulge, lanes 32-63 (group 1) for the trailing one -- every "uniform" quantity
// (bulge column, reflector, row / column indices) is uniform per half-wave, the row and column operations run as
// sub-passes of 32 columns / rows per group.  At distance >= 3 the two row operations touch disjoint rows (k..k+2 and
// k-3..k-1) and the two column operations disjoint columns, so within a pass every element has one writer; an element
// that receives a row operation of one bulge and a column operation of the other gets them in successive passes, and
// left and right multiplications commute.  An iteration is (row passes of both) -> (column passes of both) and the result
// is that of two consecutive sweeps with the same double shift (oracle/hqr2_proto.py restates the iteration,
// tools/qr2_lane_model.py the lane / pointer / validity bookkeeping below with write-overlap assertions; both were checked
// against LAPACK before the kernel ran: 0.66 iterations per single-bulge step on the benchmark's matrices).
//
// Other differences from qr_step3: (1) the bulge column is read back from LDS at the top of an iteration (three
// broadcast reads per group) instead of six v_readlane; (2) it is not overwritten with (beta, 0, 0): column k-1 simply
// takes part in the row operation, which leaves (beta~, eps, eps) -- rounding-level entries below the sub-diagonal
// that later sweeps treat as data (a backward error of the size every other column already carries; exact zeros of a
// reducible matrix stay exact because a zero tail gives u = 0); (3) lanes own FIXED columns (i - 31 + j | i - 63 + j)
// and rows (l + j | l + 32 + j) for a whole sweep, so the address arithmetic of an iteration is three pointer
// increments and validity is one compare of a per-lane constant with a wave-uniform counter per sub-pass.
// Sweeps whose start row m lies below the top of the window (m > l, ~7 %) chase a single bulge (group 1 stays parked).
// ---------------------------------------------------------------------------------------------------
#ifndef EGS_QR2_DIST
#define EGS_QR2_DIST 3
#endif
constexpr int kBulgeDist = EGS_QR2_DIST;   // rows between the two bulges (>= 3: see below)
#ifdef EGS_QR_PROF
// development only: shader-clock cycles per part of the QR iteration, accumulated per wave and printed by lane 0
struct QrProf { long long scan, setup, gen, fast, ngen, nfast, nsweep; };
__device__ __forceinline__ long long qr_now() { return (long long)__builtin_readcyclecounter(); }
#define QRP(...) __VA_ARGS__
#else
#define QRP(...)
#endif
constexpr int kNever = 1 << 20;               // validity constant of a lane that must not store

// Reflector of the bulge column (a0; a1, a2) without beta.
// The fast path squares the components as they are.  Below |a| ~ 1e-154 the squares underflow: the norm comes out too
// small (or zero) while the components are not, and with an exactly zero a0 (skew-symmetric input has them)
// u = a / (|a0| + norm) explodes -- the first form of this routine returned eigenvalues of 1e+125, without a flag, for
// 1.4 % of random skew-symmetric matrices, whose zero diagonal keeps LAPACK's deflation test waiting until the
// sub-diagonal itself underflows; a constant matrix (Hessenberg form: a cascade down to a trailing block of 1e-190) never
// converged.  So: a column with |a|^2 < 2^-800 takes a rare branch that rescales it by a power of two first (u and t are
// invariant under scaling of a -- this is xLARFG's rescaling), and an exactly zero column gets the reflection
// diag(-1, 1, 1).  The test is one compare and a never-taken branch in place of the clamp the reciprocal square root
// needed anyway.  (The n = 64 drivers scale every matrix to max |entry| in [1, 2), hess_reg.h, and deflate sub-diagonal
// entries below 1e-200 absolutely, so that nothing iterates on blocks near the underflow threshold itself.)
constexpr double kTinyCol = 1.4996968138956309e-241;   // 2^-800

__device__ __forceinline__ void larfg3_coeffs(double a0, double a1, double a2, double& t1, double& t2, double& t3,
                                              double& u1, double& u2) {
    double nrm2 = fma(a0, a0, fma(a2, a2, a1 * a1));
    if (__builtin_expect(__ballot(nrm2 < kTinyCol) != 0ULL, 0)) {  // rare: some group's bulge column is tiny
        const bool z = nrm2 < kTinyCol;
        const double big = fmax(fabs(a0), fmax(fabs(a1), fabs(a2)));
        const bool nz = big > 0.0;
        const int e = nz ? ilogb(big) : 0;
        const double b0 = nz ? scalbn(a0, -e) : 1.0, b1 = nz ? scalbn(a1, -e) : 0.0, b2 = nz ? scalbn(a2, -e) : 0.0;
        a0 = z ? b0 : a0;
        a1 = z ? b1 : a1;
        a2 = z ? b2 : a2;
        nrm2 = z ? fma(b0, b0, fma(b2, b2, b1 * b1)) : nrm2;
    }
    const double rn = fast_rsqrt(nrm2);
    const double apn = fabs(a0) + nrm2 * rn;                       // |a0 - beta|
    const double scal = copysign(fast_rcp(apn), a0);               // 1 / (a0 - beta)
    t1 = apn * rn;                                                 // (beta - a0) / beta
    u1 = a1 * scal;
    u2 = a2 * scal;
    t2 = t1 * u1;
    t3 = t1 * u2;
}

// Single-width LDS reads, hand-issued.  The compiler fuses neighbouring 8-byte reads into ds_read2_b64, which occupies
// the LDS for 8 cycles against 2 + 2 for two ds_read_b64 (MI355X_MICROARCH.md, LDS table), and in its steady state the
// sweep is bound by exactly that pipe (eight waves per CU: iteration time = 8 x LDS cycles per iteration / ~0.93).
// lds_issue3*: issue the reads and wait only for the FIRST three (the bulge column, needed by the reflector at once);
// the other operands stay in flight until lds_land() -- no instruction may touch them in between (their registers are
// outputs of the issuing statement only so that the allocator reserves them).
__device__ __forceinline__ unsigned lds_addr(const double* p) { return (unsigned)(uintptr_t)p; }

template <int S1>
__device__ __forceinline__ void lds_issue3x2(const double* a, const double* p, double& a0, double& a1, double& a2,
                                             double& p0, double& p1, double& p2) {
    asm volatile(
        "ds_read_b64 %0, %6\n\t"
        "ds_read_b64 %1, %6 offset:%8\n\t"
        "ds_read_b64 %2, %6 offset:%9\n\t"
        "ds_read_b64 %3, %7\n\t"
        "ds_read_b64 %4, %7 offset:%8\n\t"
        "ds_read_b64 %5, %7 offset:%9\n\t"
        "s_waitcnt lgkmcnt(3)"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(p0), "=&v"(p1), "=&v"(p2)
        : "v"(lds_addr(a)), "v"(lds_addr(p)), "n"(S1 * 8), "n"(S1 * 16)
        : "memory");
}

template <int S1>
__device__ __forceinline__ void lds_issue3x3(const double* a, const double* p, const double* q, double& a0, double& a1,
                                             double& a2, double& p0, double& p1, double& p2, double& q0, double& q1,
                                             double& q2) {
    asm volatile(
        "ds_read_b64 %0, %9\n\t"
        "ds_read_b64 %1, %9 offset:%12\n\t"
        "ds_read_b64 %2, %9 offset:%13\n\t"
        "ds_read_b64 %3, %10\n\t"
        "ds_read_b64 %4, %10 offset:%12\n\t"
        "ds_read_b64 %5, %10 offset:%13\n\t"
        "ds_read_b64 %6, %11\n\t"
        "ds_read_b64 %7, %11 offset:%12\n\t"
        "ds_read_b64 %8, %11 offset:%13\n\t"
        "s_waitcnt lgkmcnt(6)"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(q0), "=&v"(q1), "=&v"(q2)
        : "v"(lds_addr(a)), "v"(lds_addr(p)), "v"(lds_addr(q)), "n"(S1 * 8), "n"(S1 * 16)
        : "memory");
}

// column phase: all reads, complete on return
template <int S1>
__device__ __forceinline__ void lds_read3(const double* p, double& p0, double& p1, double& p2) {
    asm volatile(
        "ds_read_b64 %0, %3\n\t"
        "ds_read_b64 %1, %3 offset:%4\n\t"
        "ds_read_b64 %2, %3 offset:%5\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(p0), "=&v"(p1), "=&v"(p2)
        : "v"(lds_addr(p)), "n"(S1 * 8), "n"(S1 * 16)
        : "memory");
}

template <int S1>
__device__ __forceinline__ void lds_read3x2(const double* p, const double* q, double& p0, double& p1, double& p2,
                                            double& q0, double& q1, double& q2) {
    asm volatile(
        "ds_read_b64 %0, %6\n\t"
        "ds_read_b64 %1, %6 offset:%8\n\t"
        "ds_read_b64 %2, %6 offset:%9\n\t"
        "ds_read_b64 %3, %7\n\t"
        "ds_read_b64 %4, %7 offset:%8\n\t"
        "ds_read_b64 %5, %7 offset:%9\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(q0), "=&v"(q1), "=&v"(q2)
        : "v"(lds_addr(p)), "v"(lds_addr(q)), "n"(S1 * 8), "n"(S1 * 16)
        : "memory");
}

// Predicated store of three doubles p[0], p[S1], p[2 S1] by the lanes with e <= thr (GE: e >= thr; thr wave-uniform): v_cmpx writes
// EXEC directly, the stores run under it (three ds_write_b64: 3 x 6 store-path cycles against 13 + 6 for ds_write2_b64 +
// ds_write_b64 -- measured 2.1 % of the C2 step, round 4; -DEGS_STORE3_WRITE2 restores the pair form), EXEC goes back
// to `full` -- two instructions of overhead instead of the
// compare + s_and_saveexec + s_cbranch_execz + s_or the compiler emits for `if (e <= thr) { ... }`, and no branch.
// `full` is the EXEC mask the enclosing steady-state loop was ENTERED with (read once in front of the loop, qr2_segment:
// no instruction per iteration), so the statement stays correct if a compiler ever runs that loop under a predicate;
// tests/test_isa_invariants.py checks on the emitted code that nothing else inside such a loop writes EXEC.
// (Saving and restoring EXEC inside the statement measured 0.9 % of the C2 step, `s_mov_b64 exec, -1` assumed all 64
// lanes live without telling anybody.)
template <int S1, bool GE>
__device__ __forceinline__ void lds_store3_where(int e, int thr, double* p, double v0, double v1, double v2,
                                                 unsigned long long full) {
    static_assert(2 * S1 < 256, "ds_write2_b64 offsets are 8-bit, in units of 8 bytes");
    if (GE)
        asm volatile(
#ifndef EGS_STORE3_WRITE2
            "v_cmpx_le_i32_e32 vcc, %[thr], %[e]\n\t"
            "ds_write_b64 %[p], %[v0]\n\t"
            "ds_write_b64 %[p], %[v1] offset:%[o1b]\n\t"
            "ds_write_b64 %[p], %[v2] offset:%[o2]\n\t"
            "s_mov_b64 exec, %[full]"
#else
            "v_cmpx_le_i32_e32 vcc, %[thr], %[e]\n\t"
            "ds_write2_b64 %[p], %[v0], %[v1] offset1:%[o1]\n\t"
            "ds_write_b64 %[p], %[v2] offset:%[o2]\n\t"
            "s_mov_b64 exec, %[full]"
#endif
            :
            : [e] "v"(e), [thr] "s"(thr), [p] "v"((unsigned)(uintptr_t)p), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2),
              [o1] "n"(S1), [o1b] "n"(S1 * 8), [o2] "n"(2 * S1 * 8), [full] "s"(full)
            : "vcc", "memory");
    else
        asm volatile(
#ifndef EGS_STORE3_WRITE2
            "v_cmpx_ge_i32_e32 vcc, %[thr], %[e]\n\t"
            "ds_write_b64 %[p], %[v0]\n\t"
            "ds_write_b64 %[p], %[v1] offset:%[o1b]\n\t"
            "ds_write_b64 %[p], %[v2] offset:%[o2]\n\t"
            "s_mov_b64 exec, %[full]"
#else
            "v_cmpx_ge_i32_e32 vcc, %[thr], %[e]\n\t"
            "ds_write2_b64 %[p], %[v0], %[v1] offset1:%[o1]\n\t"
            "ds_write_b64 %[p], %[v2] offset:%[o2]\n\t"
            "s_mov_b64 exec, %[full]"
#endif
            :
            : [e] "v"(e), [thr] "s"(thr), [p] "v"((unsigned)(uintptr_t)p), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2),
              [o1] "n"(S1), [o1b] "n"(S1 * 8), [o2] "n"(2 * S1 * 8), [full] "s"(full)
            : "vcc", "memory");
}

struct Qr2Lane {
    double* rowp;      // &H(k, c1): this lane's primary column of the row operation, at the group's bulge row k
    double* colp;      // &H(r0, k): this lane's primary row of the column operation
    double* bp;        // &H(k, k-1): the bulge column
    int inc_r, inc_c, inc_b;   // advance per iteration in doubles (0 while the group is parked)
    int d_r, d_c;      // element offset of the secondary column / row
    // validity in iteration t of the sweep -- every test compares a per-lane constant with the iteration counter itself,
    // so the steady-state loop carries one scalar (the window thresholds L = Lbase - t, M = Mbase + t are folded in)
    int er0, er1;      // row pass:    primary valid <=> t <= er0, secondary <=> t <= er1  (er1 = er0 - 32)
    int ec0, ec1;      // column pass: primary valid <=> t >= ec0, secondary <=> t >= ec1
};

// One iteration: both groups advance their bulge by one row.
//   GEN : an iteration in which some group takes its first step (bulge column = start vector; at the top of the window
//         column l-1 is excluded) or its last one (order-2 reflector: third row / column untouched); `first`, `last`
//         are per-lane (uniform per group).  At most four iterations of a sweep.
//   R2 / C2 : some lane's secondary column lies inside [k-1, i] / secondary row inside [l, k+3].  Both are monotone in
//         the iteration count, so a sweep is a handful of segments of constant (R2, C2) and the steady-state loop of
//         a segment carries no uniform branch at all.
// An iteration is two LDS round trips and the reflector chain: every load of the row phase (bulge column, primary and
// secondary row operands -- none depends on the reflector) is issued in one batch at the top, every load of the column
// phase in one batch behind the row stores.  (Left alone, the compiler sinks each sub-pass's loads into its predicated
// store block: five serialised round trips per iteration.)  The empty asm statements pin the batches.
template <bool GEN, bool R2, bool C2, class V>
__device__ __forceinline__ void qr2_iter(Qr2Lane& S, const int t, const bool first, const bool last,
                                         const bool top, const double v0, const double v1, const double v2,
                                         const unsigned long long full = ~0ULL) {
    constexpr int SR = V::kSR, SC = V::kSC;
    double a0, a1, a2, x0, x1, x2;
    double* const rp2 = S.rowp + S.d_r;
    double y0, y1, y2;
#ifndef EGS_QR2_COMPILER_READS
    if (R2) lds_issue3x3<SR>(S.bp, S.rowp, rp2, a0, a1, a2, x0, x1, x2, y0, y1, y2);
    else lds_issue3x2<SR>(S.bp, S.rowp, a0, a1, a2, x0, x1, x2);
#else
    a0 = S.bp[0], a1 = S.bp[SR], a2 = S.bp[2 * SR];
    x0 = S.rowp[0], x1 = S.rowp[SR], x2 = S.rowp[2 * SR];
    if (R2) {
        y0 = rp2[0];
        y1 = rp2[SR];
        y2 = rp2[2 * SR];
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(y0), "+v"(y1), "+v"(y2));
    } else {
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
    }
#endif
#ifndef EGS_QR2_COMPILER_READS
    if (GEN) {                      // the special iterations touch x2 / y2 before the reflector: land everything now
        if (R2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(y0), "+v"(y1), "+v"(y2));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2));
    }
#endif
    int tg = t;
    if (GEN) {
        a0 = first ? v0 : a0;
        a1 = first ? v1 : a1;
        a2 = first ? v2 : (last ? 0.0 : a2);
        x2 = last ? 0.0 : x2;
        if (R2) y2 = last ? 0.0 : y2;
        // a first step at the top of the window (m == l): column l-1 is outside it (H(l, l-1) = 0).  Below the top
        // (m > l) column m-1 takes part like any bulge column, which scales H(m, m-1) by (1 - t1) as xLAHQR does
        tg = (first && top) ? t + 1 : t;
    }
    double t1, t2, t3, u1, u2;
    larfg3_coeffs(a0, a1, a2, t1, t2, t3, u1, u2);
#ifndef EGS_QR2_COMPILER_READS
    if (!GEN) {                     // the row operands have been in flight beside the reflector arithmetic
        if (R2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(t1));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(t1));
    }
#endif
    {
        const double sum = fma(u2, x2, fma(u1, x1, x0));
        const double n0 = fma(-sum, t1, x0), n1 = fma(-sum, t2, x1), n2 = fma(-sum, t3, x2);
        if (!GEN) {
            lds_store3_where<SR, true>(S.er0, t, S.rowp, n0, n1, n2, full);
        } else if (S.er0 >= tg) {
            S.rowp[0] = n0;
            S.rowp[SR] = n1;
            if (!last) S.rowp[2 * SR] = n2;
        }
    }
    if (R2) {
        const double sum = fma(u2, y2, fma(u1, y1, y0));
        const double n0 = fma(-sum, t1, y0), n1 = fma(-sum, t2, y1), n2 = fma(-sum, t3, y2);
        if (!GEN) {
            lds_store3_where<SR, true>(S.er1, t, rp2, n0, n1, n2, full);
        } else if (S.er1 >= tg) {
            rp2[0] = n0;
            rp2[SR] = n1;
            if (!last) rp2[2 * SR] = n2;
        }
    }
    wave_lds_sync();
    double c0, c1, c2;
    double* const cp2 = S.colp + S.d_c;
    double d0, d1, d2;
#ifndef EGS_QR2_COMPILER_READS
    if (C2) lds_read3x2<SC>(S.colp, cp2, c0, c1, c2, d0, d1, d2);
    else lds_read3<SC>(S.colp, c0, c1, c2);
#else
    c0 = S.colp[0], c1 = S.colp[SC], c2 = S.colp[2 * SC];
    if (C2) {
        d0 = cp2[0];
        d1 = cp2[SC];
        d2 = cp2[2 * SC];
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(d0), "+v"(d1), "+v"(d2));
    } else {
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2));
    }
#endif
    if (GEN) {
        c2 = last ? 0.0 : c2;
        if (C2) d2 = last ? 0.0 : d2;
    }
    {
        const double sum = fma(u2, c2, fma(u1, c1, c0));
        const double n0 = fma(-sum, t1, c0), n1 = fma(-sum, t2, c1), n2 = fma(-sum, t3, c2);
        if (!GEN) {
            lds_store3_where<SC, false>(S.ec0, t, S.colp, n0, n1, n2, full);
        } else if (S.ec0 <= t) {
            S.colp[0] = n0;
            S.colp[SC] = n1;
            if (!last) S.colp[2 * SC] = n2;
        }
    }
    if (C2) {
        const double sum = fma(u2, d2, fma(u1, d1, d0));
        const double n0 = fma(-sum, t1, d0), n1 = fma(-sum, t2, d1), n2 = fma(-sum, t3, d2);
        if (!GEN) {
            lds_store3_where<SC, false>(S.ec1, t, cp2, n0, n1, n2, full);
        } else if (S.ec1 <= t) {
            cp2[0] = n0;
            cp2[SC] = n1;
            if (!last) cp2[2 * SC] = n2;
        }
    }
    wave_lds_sync();
    S.rowp += (GEN && last) ? 0 : S.inc_r;         // a group parks where its last step left it (addresses stay valid)
    S.colp += (GEN && last) ? 0 : S.inc_c;
    S.bp += (GEN && (first || last)) ? 0 : S.inc_b;   // bp starts at &H(m+1, m): the first step does not read the bulge
}

// steady-state iterations t0 <= t < t1 of one (R2, C2) segment
template <bool R2, bool C2, class V>
__device__ __forceinline__ void qr2_segment(Qr2Lane& S, int t0, const int t1) {
    const unsigned long long full = __builtin_amdgcn_read_exec();    // the lanes this loop runs with (lds_store3_where)
    for (int t = t0; t < t1; ++t) qr2_iter<false, R2, C2, V>(S, t, false, false, false, 0.0, 0.0, 0.0, full);
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the MERGED iteration -- two sub-passes instead of three for windows of 33 .. 59 rows.
//
// Row and column operations apply the same formula n_j = x_j - (x_0 + u_1 x_1 + u_2 x_2) t_j to a triple of elements
// that differ only in base address and stride, and a bulge needs R + C = (i - k + 2) + (k + 4 - l) = window + 6 <= 64
// triples per iteration: two passes of a 32-lane group hold them.  What the fixed lane ownership of qr2_iter cannot give
// is the one ordering constraint: the corner (rows k..k+3 x columns k-1..k+2) gets the row operation first, so all its
// COLUMN lanes must sit in the earlier pass and all its ROW lanes in the later one.  Moving primaries do:
//   pass A, lane j:  the MOVING column k + 30 - j  (rows k..k+2; valid while <= i; lanes 28..31 are the corner columns)
//                    or, once that column has left the window, the FIXED top row l + j  (columns k..k+2; valid once
//                    <= k - 29);
//   pass B, lane j:  the FIXED right column i - j  (valid while >= k + 31, i.e. not covered by pass A)
//                    or the MOVING bottom row k - 28 + j  (valid while l <= row <= i; lanes 28..31 are the corner rows).
// For a window of <= 59 rows the two roles of a lane never overlap in time (brute-forced for every (l, i, m, t):
// coverage, disjointness, corner order), every element receives the same operations in the same order as in
// qr2_iter -- the spectra are bit-identical -- and a lane switches role at most once per sweep, by a compare of the
// iteration counter with a per-lane constant.  Per pass: two address selects, one threshold select, two address adds
// (per-lane strides: no immediate offsets), pointer bumps; one sub-pass (3 reads, 3 stores, 6 FMAs) less per iteration.
// ---------------------------------------------------------------------------------------------------
struct Qr2M {
    unsigned pa1, pa2, pb1, pb2;   // LDS byte addresses of the four roles' first elements in the current iteration
    unsigned bp;                   // &H(k, k-1): the bulge column
    int hiA1, loA2;                // pass A: moving column valid <=> t <= hiA1; fixed top row valid <=> t >= loA2
    int hiB1, loB2, hiB2;          // pass B: fixed right column valid <=> t <= hiB1; moving bottom row <=> loB2 <= t <= hiB2
    unsigned live;                 // 1 while the lane's group is chasing (pointers advance), 0 while parked
};

template <int S1>
__device__ __forceinline__ void lds_issue_b3_a3(unsigned b, unsigned p0, unsigned p1, unsigned p2, double& a0, double& a1,
                                                double& a2, double& x0, double& x1, double& x2) {
    asm volatile(
        "ds_read_b64 %0, %6\n\t"
        "ds_read_b64 %1, %6 offset:%10\n\t"
        "ds_read_b64 %2, %6 offset:%11\n\t"
        "ds_read_b64 %3, %7\n\t"
        "ds_read_b64 %4, %8\n\t"
        "ds_read_b64 %5, %9\n\t"
        "s_waitcnt lgkmcnt(3)"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(x0), "=&v"(x1), "=&v"(x2)
        : "v"(b), "v"(p0), "v"(p1), "v"(p2), "n"(S1 * 8), "n"(S1 * 16)
        : "memory");
}

__device__ __forceinline__ void lds_read3_a3(unsigned p0, unsigned p1, unsigned p2, double& x0, double& x1, double& x2) {
    asm volatile(
        "ds_read_b64 %0, %3\n\t"
        "ds_read_b64 %1, %4\n\t"
        "ds_read_b64 %2, %5\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x0), "=&v"(x1), "=&v"(x2)
        : "v"(p0), "v"(p1), "v"(p2)
        : "memory");
}

// lanes with e >= thr store (v0, v1, v2) to (p0, p1, p2); EXEC goes back to the loop-entry mask (see lds_store3_where)
__device__ __forceinline__ void lds_store3_a3_where(int e, int thr, unsigned p0, unsigned p1, unsigned p2, double v0,
                                                    double v1, double v2, unsigned long long full) {
    asm volatile(
        "v_cmpx_le_i32_e32 vcc, %[thr], %[e]\n\t"
        "ds_write_b64 %[p0], %[v0]\n\t"
        "ds_write_b64 %[p1], %[v1]\n\t"
        "ds_write_b64 %[p2], %[v2]\n\t"
        "s_mov_b64 exec, %[full]"
        :
        : [e] "v"(e), [thr] "s"(thr), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2),
          [full] "s"(full)
        : "vcc", "memory");
}

// an LDS byte address as the pointer the immediate-offset helpers take (they only use its low 32 bits)
__device__ __forceinline__ const double* as_lds(unsigned a) { return reinterpret_cast<const double*>((uintptr_t)a); }

__device__ __forceinline__ void lds_store1_a(unsigned p, double v) {
    asm volatile("ds_write_b64 %0, %1" : : "v"(p), "v"(v) : "memory");
}

// One merged iteration.  GEN as in qr2_iter: some group takes its first or its last step (`first`, `last` per lane,
// uniform per group; `top`: the sweep starts at the top of the window, column l-1 does not exist).
// SA / SB: some lane may be in its second role in pass A (fixed top row) / its first role in pass B (fixed right column)
// in this segment of the sweep; without them the pass has one role, one stride, and no selects.
template <bool GEN, bool SA, bool SB, class V>
__device__ __forceinline__ void qr2_iter_m(Qr2M& M, const int t, const int j, const bool first, const bool last,
                                           const bool top, const double v0, const double v1, const double v2,
                                           const unsigned long long full = ~0ULL) {
    constexpr int SR = V::kSR, SC = V::kSC;
    constexpr unsigned kSR8 = SR * 8, kSC8 = SC * 8, kDiag8 = (SR + SC) * 8;
    // ---- pass A: moving columns | fixed top rows ----
    const bool selA = SA && t >= M.loA2;
    const unsigned pa = selA ? M.pa2 : M.pa1;
    const unsigned sa = selA ? kSC8 : kSR8;
    const int thrA = selA ? kNever : M.hiA1;
    unsigned pa_1 = 0, pa_2 = 0;
    double a0, a1, a2, x0, x1, x2;
    if (SA) {
        pa_1 = pa + sa;
        pa_2 = pa_1 + sa;
        lds_issue_b3_a3<SR>(M.bp, pa, pa_1, pa_2, a0, a1, a2, x0, x1, x2);
    } else {                                                           // one role, one stride: immediate offsets
        lds_issue3x2<SR>(as_lds(M.bp), as_lds(pa), a0, a1, a2, x0, x1, x2);
    }
    if (GEN) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2));
        a0 = first ? v0 : a0;
        a1 = first ? v1 : a1;
        a2 = first ? v2 : (last ? 0.0 : a2);
        x2 = last ? 0.0 : x2;
    }
    double t1, t2, t3, u1, u2;
    larfg3_coeffs(a0, a1, a2, t1, t2, t3, u1, u2);
    if (!GEN) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(t1));
    {
        const double sum = fma(u2, x2, fma(u1, x1, x0));
        const double n0 = fma(-sum, t1, x0), n1 = fma(-sum, t2, x1), n2 = fma(-sum, t3, x2);
        if (!GEN) {
            if (SA) lds_store3_a3_where(thrA, t, pa, pa_1, pa_2, n0, n1, n2, full);
            else lds_store3_where<SR, true>(thrA, t, const_cast<double*>(as_lds(pa)), n0, n1, n2, full);
        } else {
            // a first step at the top of the window: column l-1 (lane 31 of the moving columns) is outside it
            const bool ok = thrA >= t && !(first && top && !selA && j == 31);
            if (ok) {
                lds_store1_a(pa, n0);
                lds_store1_a(pa_1, n1);
                if (!last) lds_store1_a(pa_2, n2);
            }
        }
    }
    wave_lds_sync();
    // ---- pass B: fixed right columns | moving bottom rows ----
    const bool inB2 = t >= M.loB2;                                     // the lane's bottom row has entered the window
    const bool selB = !SB || inB2;
    const unsigned pb = selB ? M.pb2 : M.pb1;
    const unsigned sb = selB ? kSC8 : kSR8;
    const int thrB = inB2 ? M.hiB2 : (SB ? M.hiB1 : -kNever);
    unsigned pb_1 = 0, pb_2 = 0;
    double c0, c1, c2;
    if (SB) {
        pb_1 = pb + sb;
        pb_2 = pb_1 + sb;
        lds_read3_a3(pb, pb_1, pb_2, c0, c1, c2);
    } else {
        lds_read3<SC>(as_lds(pb), c0, c1, c2);
    }
    if (GEN) c2 = last ? 0.0 : c2;
    {
        const double sum = fma(u2, c2, fma(u1, c1, c0));
        const double n0 = fma(-sum, t1, c0), n1 = fma(-sum, t2, c1), n2 = fma(-sum, t3, c2);
        if (!GEN) {
            if (SB) lds_store3_a3_where(thrB, t, pb, pb_1, pb_2, n0, n1, n2, full);
            else lds_store3_where<SC, true>(thrB, t, const_cast<double*>(as_lds(pb)), n0, n1, n2, full);
        } else if (thrB >= t) {
            lds_store1_a(pb, n0);
            lds_store1_a(pb_1, n1);
            if (!last) lds_store1_a(pb_2, n2);
        }
    }
    wave_lds_sync();
    // ---- the bulge moves one row down (a group parks where its last step left it) ----
    const unsigned adv = (GEN && last) ? 0u : M.live;
    M.pa1 += adv * kDiag8;
    M.pa2 += adv * kSC8;
    M.pb1 += adv * kSR8;
    M.pb2 += adv * kDiag8;
    M.bp += ((GEN && (first || last)) ? 0u : M.live) * kDiag8;   // bp starts at &H(m+1, m): the first step does not read it
}

template <bool SA, bool SB, class V>
__device__ __forceinline__ void qr2_segment_m(Qr2M& M, int t0, const int t1, const int j) {
    const unsigned long long full = __builtin_amdgcn_read_exec();
    for (int t = t0; t < t1; ++t) qr2_iter_m<false, SA, SB, V>(M, t, j, false, false, false, 0.0, 0.0, 0.0, full);
}

// One sweep with merged iterations (window of 33 .. 59 rows); same special iterations, start vectors and parking as
// qr2_sweep.
template <class V>
__device__ __forceinline__ void qr2_sweep_m(const V& H, const int l, const int i, const int m, const int lane, double v0,
                                            double v1, double v2, const double rt1r, const double rt1i,
                                            const double rt2r, const double rt2i) {
    const int g = lane >> 5, j = lane & 31;
    const bool two = (m == l);
    const int nA = i - m;                                   // steps of one bulge: k = m .. i-1
    const int T = nA + (two ? kBulgeDist : 0);
    Qr2M M;
    M.pa1 = lds_addr(&H.at(m, m + 30 - j));                 // (addresses of lanes that never become valid may lie outside
    M.pa2 = lds_addr(&H.at(l + j, m));                      //  the array: they are only ever loaded from, and LDS loads
    M.pb1 = lds_addr(&H.at(m, i - j));                      //  beyond the allocation return zero)
    M.pb2 = lds_addr(&H.at(m - 28 + j, m));
    M.bp = lds_addr(&H.at(m + 1, m));
    const int off = kBulgeDist * g;                         // the trailing group is `off` iterations behind
    const int hiA1 = i - 30 + j - m + off, loA2 = l + j + 29 - m + off;
    const int hiB1 = i - j - 31 - m + off, loB2 = l + 28 - j - m + off, hiB2 = i + 28 - j - m + off;
    const bool liveA = g == 0;
    M.hiA1 = liveA ? hiA1 : -kNever;
    M.loA2 = liveA ? loA2 : kNever;
    M.hiB1 = liveA ? hiB1 : -kNever;
    M.loB2 = liveA ? loB2 : kNever;
    M.hiB2 = liveA ? hiB2 : -kNever;
    M.live = liveA ? 1u : 0u;
    const int tLastA = nA - 1, tFirstB = two ? kBulgeDist : T, tLastB = two ? nA - 1 + kBulgeDist : T;
    const int tA2 = l + 29 - m;                              // first iteration in which some lane's top row is valid (g = 0, j = 0)
    const int tB1 = i - 31 - m + (two ? kBulgeDist : 0);     // last iteration in which some lane's right column is (j = 0)
    int t = 0;
    while (t < T) {
        int tg = tLastB;
        if (tLastA >= t) tg = min(tg, tLastA);
        if (tFirstB >= t) tg = min(tg, tFirstB);
        if (t == 0) tg = 0;
        // steady-state iterations t .. tg-1, cut into segments by which second roles can occur: fixed right columns
        // (pass B) while t <= tB1, fixed top rows (pass A) from t >= tA2 on
        while (t < tg) {
            const bool sb = t <= tB1, sa = t >= tA2;
            int e = tg;
            if (sb) e = min(e, tB1 + 1);
            if (!sa) e = min(e, tA2);
            if (sa) {
                if (sb) qr2_segment_m<true, true, V>(M, t, e, j);
                else qr2_segment_m<true, false, V>(M, t, e, j);
            } else {
                if (sb) qr2_segment_m<false, true, V>(M, t, e, j);
                else qr2_segment_m<false, false, V>(M, t, e, j);
            }
            t = e;
        }
        if (t >= T) break;
        const bool fA = t == 0, lA = t == tLastA, fB = t == tFirstB, lB = t == tLastB;
        if (fB) {
            // the trailing bulge starts at row m on the matrix as the leading bulge left it (see qr2_sweep)
            const double hmm = H.at(m, m), hm1m = H.at(m + 1, m), hm1m1 = H.at(m + 1, m + 1);
            double s = fabs(hmm - rt2r) + fabs(rt2i) + fabs(hm1m);
            double rs = fast_rcp(s);
            const double h21s = hm1m * rs;
            v0 = h21s * H.at(m, m + 1) + (hmm - rt1r) * ((hmm - rt2r) * rs) - rt1i * (rt2i * rs);
            v1 = h21s * (hmm + hm1m1 - rt1r - rt2r);
            v2 = h21s * H.at(m + 2, m + 1);
            s = fabs(v0) + fabs(v1) + fabs(v2);
            rs = fast_rcp(s);
            v0 *= rs;
            v1 *= rs;
            v2 *= rs;
            M.hiA1 = liveA ? M.hiA1 : hiA1;
            M.loA2 = liveA ? M.loA2 : loA2;
            M.hiB1 = liveA ? M.hiB1 : hiB1;
            M.loB2 = liveA ? M.loB2 : loB2;
            M.hiB2 = liveA ? M.hiB2 : hiB2;
            M.live = liveA ? M.live : 1u;
        }
        {
            const bool first = liveA ? fA : fB, last = liveA ? lA : lB;
            qr2_iter_m<true, true, true, V>(M, t, j, first, last, two, v0, v1, v2);
            const bool park = last;
            M.hiA1 = park ? -kNever : M.hiA1;
            M.loA2 = park ? kNever : M.loA2;
            M.hiB1 = park ? -kNever : M.hiB1;
            M.loB2 = park ? kNever : M.loB2;
            M.hiB2 = park ? -kNever : M.hiB2;
            M.live = park ? 0u : M.live;
        }
        ++t;
    }
}

// One sweep over the window [l, i] from start row m with start vector (v0, v1, v2); (rt1r, rt1i, rt2r, rt2i) are the
// shifts, needed again for the trailing bulge's start vector.
template <class V>
__device__ __forceinline__ void qr2_sweep(const V& H, const int l, const int i, const int m, const int lane, double v0,
                                          double v1, double v2, const double rt1r, const double rt1i, const double rt2r,
                                          const double rt2i, const int no_merge
#ifdef EGS_QR_PROF
                                          , QrProf& P
#endif
                                          ) {
    QRP(long long tp = qr_now();)
#ifndef EGS_QR2_MERGE_KN
#define EGS_QR2_MERGE_KN 0
#endif
    // window of 33 .. 59 rows: two sub-passes per iteration (EGS_QR2_NO_MERGE=1 at launch: the three-pass form, A/B and
    // the bit-identity test)
    if (V::kN >= EGS_QR2_MERGE_KN && !no_merge && i - l >= 32 && i - l <= 58) {
        qr2_sweep_m<V>(H, l, i, m, lane, v0, v1, v2, rt1r, rt1i, rt2r, rt2i);
        return;
    }
    constexpr int SR = V::kSR, SC = V::kSC;
    const int g = lane >> 5, j = lane & 31;
    const bool two = (m == l);
    const int nA = i - m;                                   // steps of one bulge: k = m .. i-1
    const int T = nA + (two ? kBulgeDist : 0);
    const int c1 = max(i - 31 + j, 0), c0 = max(i - 63 + j, 0);
    const int r0 = min(l + j, V::kN - 1), r1 = min(l + 32 + j, V::kN - 1);
    Qr2Lane S;
    S.rowp = &H.at(m, c1);
    S.colp = &H.at(r0, m);
    S.bp = &H.at(m + 1, m);
    S.d_r = (c0 - c1) * SC;
    S.d_c = (r1 - r0) * SR;
    // thresholds of iteration t:  L = i - (m + t) + 1 = Lbase - t,  M = m + t + 3 - l = Mbase + t
    const int Lbase = nA + 1, Mbase = m + 3 - l;
    // row pass:    31 - j - dist g      <= L  (primary; secondary: <= L - 32)  <=>  t <= er0 (er0 - 32)
    // column pass: j (+ 32) + dist g    <= M                                   <=>  t >= ec0 (ec1)
    const int er0_live = Lbase - (31 - j - kBulgeDist * g);
    const int ec0_live = (l + j <= i) ? j + kBulgeDist * g - Mbase : kNever;
    const int ec1_live = (l + 32 + j <= i) ? j + 32 + kBulgeDist * g - Mbase : kNever;
    const bool liveA = g == 0;
    S.er0 = liveA ? er0_live : -kNever;
    S.er1 = liveA ? er0_live - 32 : -kNever;
    S.ec0 = liveA ? ec0_live : kNever;
    S.ec1 = liveA ? ec1_live : kNever;
    S.inc_r = liveA ? SR : 0;
    S.inc_c = liveA ? SC : 0;
    S.inc_b = liveA ? SR + SC : 0;
    const int tR = Lbase - (32 - kBulgeDist) + 1;            // R2 holds for t < tR   (L >= 32 - kBulgeDist: group 1, lane 31)
    const int tC = 32 - Mbase;                              // C2 holds for t >= tC  (M >= 32)
    // the iterations in which a group starts or ends (at most four, in increasing order of t)
    const int tLastA = nA - 1, tFirstB = two ? kBulgeDist : T, tLastB = two ? nA - 1 + kBulgeDist : T;
    int t = 0;
    QRP({ const long long n_ = qr_now(); P.setup += n_ - tp; tp = n_; })
    while (t < T) {
        // next special iteration at or after t
        int tg = tLastB;
        if (tLastA >= t) tg = min(tg, tLastA);
        if (tFirstB >= t) tg = min(tg, tFirstB);
        if (t == 0) tg = 0;
        // steady-state iterations t .. tg-1, cut into segments of constant (R2, C2)
        while (t < tg) {
            const bool r2 = t < tR, c2 = t >= tC;
            int e = tg;
            if (r2) e = min(e, tR);
            if (!c2) e = min(e, tC);
            if (r2) {
                if (c2) qr2_segment<true, true, V>(S, t, e);
                else qr2_segment<true, false, V>(S, t, e);
            } else {
                if (c2) qr2_segment<false, true, V>(S, t, e);
                else qr2_segment<false, false, V>(S, t, e);
            }
            QRP(P.nfast += e - t;)
            t = e;
        }
        QRP({ const long long n_ = qr_now(); P.fast += n_ - tp; tp = n_; })
        if (t >= T) break;
        const bool fA = t == 0, lA = t == tLastA, fB = t == tFirstB, lB = t == tLastB;
        if (fB) {
            // the trailing bulge starts at row m on the matrix as the leading bulge left it (its rows m..m+2 and
            // columns m, m+1 are final after three steps); same formula as the start search of hqr_wave
            const double hmm = H.at(m, m), hm1m = H.at(m + 1, m), hm1m1 = H.at(m + 1, m + 1);
            double s = fabs(hmm - rt2r) + fabs(rt2i) + fabs(hm1m);
            double rs = fast_rcp(s);
            const double h21s = hm1m * rs;
            v0 = h21s * H.at(m, m + 1) + (hmm - rt1r) * ((hmm - rt2r) * rs) - rt1i * (rt2i * rs);
            v1 = h21s * (hmm + hm1m1 - rt1r - rt2r);
            v2 = h21s * H.at(m + 2, m + 1);
            s = fabs(v0) + fabs(v1) + fabs(v2);
            rs = fast_rcp(s);
            v0 *= rs;
            v1 *= rs;
            v2 *= rs;
            S.er0 = liveA ? S.er0 : er0_live;
            S.er1 = liveA ? S.er1 : er0_live - 32;
            S.ec0 = liveA ? S.ec0 : ec0_live;
            S.ec1 = liveA ? S.ec1 : ec1_live;
            S.inc_r = liveA ? S.inc_r : SR;
            S.inc_c = liveA ? S.inc_c : SC;
            S.inc_b = liveA ? S.inc_b : SR + SC;
        }
        {
            const bool first = liveA ? fA : fB, last = liveA ? lA : lB;
            const bool r2 = t < tR, c2 = t >= tC;
            if (r2) {
                if (c2) qr2_iter<true, true, true, V>(S, t, first, last, two, v0, v1, v2);
                else qr2_iter<true, true, false, V>(S, t, first, last, two, v0, v1, v2);
            } else {
                if (c2) qr2_iter<true, false, true, V>(S, t, first, last, two, v0, v1, v2);
                else qr2_iter<true, false, false, V>(S, t, first, last, two, v0, v1, v2);
            }
            const bool park = last;
            S.er0 = park ? -kNever : S.er0;
            S.er1 = park ? -kNever : S.er1;
            S.ec0 = park ? kNever : S.ec0;
            S.ec1 = park ? kNever : S.ec1;
            S.inc_r = park ? 0 : S.inc_r;
            S.inc_c = park ? 0 : S.inc_c;
            S.inc_b = park ? 0 : S.inc_b;
        }
        QRP({ const long long n_ = qr_now(); P.gen += n_ - tp; tp = n_; P.ngen += 1; })
        ++t;
    }
}

// Returns 0 when the whole spectrum has been written (or the iteration failed: info != 0), otherwise -- stop_below > 0 --
// the order r = i + 1 <= stop_below of the leading block that is still to be solved: eigenvalues r .. n-1 are written,
// rows / columns 0 .. r-1 of H hold the unreduced Hessenberg block (see hqr_tail_kernel).  n_it = the order the
// iteration limit is derived from (that of the original matrix).
template <class V>
__device__ __forceinline__ int hqr_wave(const V H, const int n, const int lane, double* __restrict__ w_out,
                                        int32_t* __restrict__ info_out, const double rescale, const int stop_below = 0,
                                        const int n_it = 0, const int no_merge = 0) {
    const double ulp = 2.220446049250313e-16;                      // dlamch('P')
    const double safmin = 2.2250738585072014e-308;
    // n = 64 layouts: the matrix was scaled to max |entry| in [1, 2), so a sub-diagonal entry below 1e-200 is negligible
    // ABSOLUTELY (xLAHQR's smlnum is 6e-291: with a zero diagonal its relative tests never fire and the iteration would
    // chase blocks all the way down into the underflow range)
    const double smlnum = V::kFast ? 1e-200 : safmin * ((double)n / ulp);
    const int nit = n_it > 0 ? n_it : n;
    const int itmax = 30 * (nit > 10 ? nit : 10);
    const int kexsh = 10;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);

    int info = 0;
    int i = n - 1;
    int kdefl = 0;
    QRP(QrProf P = {0, 0, 0, 0, 0, 0, 0}; long long tscan = qr_now(); const long long tstart = tscan;)
    while (i >= stop_below && i >= 0) {
        int l = 0;
        bool converged = false;
        for (int its = 0; its <= itmax; ++its) {
            double rt1r, rt1i, rt2r, rt2i;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            int m;
            bool done = false;
            if constexpr (V::kFast) {
                // n = 64: everything the deflation scan, the shifts and the start search read lies on three diagonals.
                // Lane k fetches its neighbourhood of them in ONE batch (clamped, always-valid addresses) together with
                // the trailing 2 x 2 block, so the preamble of a sweep is one LDS round trip instead of one per
                // section and per nested condition (it was 3.1 k cycles per sweep, 13 % of the QR stage).
                constexpr int kLast = V::kN - 1;
                const int k = (lane < kLast) ? lane : kLast, km = (k > 0) ? k - 1 : 0, kp = (k < kLast) ? k + 1 : kLast,
                          kp2 = (k < kLast - 1) ? k + 2 : kLast;
                const int i1 = (i > 0) ? i - 1 : 0;
                const double d0 = H.at(k, k), dm = H.at(km, km), dp = H.at(kp, kp);
                const double s0 = H.at(k, km), s1 = H.at(kp, k), s2 = H.at(kp2, kp);
                const double u0 = H.at(km, k), u1 = H.at(k, kp);
                double h11 = H.at(i1, i1), h21 = H.at(i, i1), h12 = H.at(i1, i), h22 = H.at(i, i);
                // ---- single small subdiagonal element: lane k tests H(k,k-1), l < k <= i (branch-free) ----
                const double hkk1 = fabs(s0);
                double tst = fabs(dm) + fabs(d0);
                if (__ballot(tst == 0.0 && lane > l && lane <= i) != 0ULL) {     // rare: both diagonal entries vanish
                    const double below = (k + 1 <= n - 1) ? fabs(s1) : 0.0;
                    const double left = (k - 2 >= 0) ? fabs(H.at(km, (k >= 2) ? k - 2 : 0)) : 0.0;
                    tst = (tst == 0.0) ? left + below : tst;
                }
                const double hk1k = fabs(u0);
                const double ab = fmax(hkk1, hk1k), ba = fmin(hkk1, hk1k);
                const double df = fabs(dm - d0);
                const double aa = fmax(fabs(d0), df), bb = fmin(fabs(d0), df);
                const double rs0 = fast_rcp(aa + ab);
                const bool crit = ba * (ab * rs0) <= fmax(smlnum, ulp * (bb * (aa * rs0)));
                const bool small = lane > l && lane <= i && (hkk1 <= smlnum || (hkk1 <= ulp * tst && crit));
                const unsigned long long smask = __ballot(small);
                l = smask ? (63 - __builtin_clzll(smask)) : l;
                l = uniform(l);
                if (l > 0 && lane == 0) H.at(l, l - 1) = 0.0;
                if (l >= i - 1) {
                    done = true;
                } else {
                    wave_lds_sync();
                    ++kdefl;
                    // ---- shifts (uniform) ----
                    if (kdefl % kexsh == 0) {
                        if (kdefl % (2 * kexsh) == 0) {
                            const double s = fabs(H.at(i, i - 1)) + fabs(H.at(i - 1, i - 2));
                            h11 = 0.75 * s + H.at(i, i);
                            h12 = -0.4375 * s;
                            h21 = s;
                            h22 = h11;
                        } else {
                            const double s = fabs(H.at(l + 1, l)) + fabs(H.at(l + 2, l + 1));
                            h11 = 0.75 * s + H.at(l, l);
                            h12 = -0.4375 * s;
                            h21 = s;
                            h22 = h11;
                        }
                    }
                    {
                        const double s = fabs(h11) + fabs(h12) + fabs(h21) + fabs(h22);
                        if (s == 0.0) {
                            rt1r = rt1i = rt2r = rt2i = 0.0;
                        } else {
                            const double rs = fast_rcp(s);
                            h11 *= rs;
                            h21 *= rs;
                            h12 *= rs;
                            h22 *= rs;
                            const double tr = (h11 + h22) * 0.5;
                            const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
                            const double ad = fmax(fabs(det), 1e-300);
                            const double rtdisc = ad * fast_rsqrt(ad);            // sqrt |det| (a shift: a few ulps are irrelevant)
                            if (det >= 0.0) {
                                rt1r = tr * s;
                                rt2r = rt1r;
                                rt1i = rtdisc * s;
                                rt2i = -rt1i;
                            } else {
                                rt1r = tr + rtdisc;
                                rt2r = tr - rtdisc;
                                if (fabs(rt1r - h22) <= fabs(rt2r - h22)) {
                                    rt1r = rt1r * s;
                                    rt2r = rt1r;
                                } else {
                                    rt2r = rt2r * s;
                                    rt1r = rt2r;
                                }
                                rt1i = 0.0;
                                rt2i = 0.0;
                            }
                        }
                    }
                    // ---- two consecutive small subdiagonals: lane mm evaluates the start vector at row mm ----
                    // (the scan may just have zeroed H(l, l-1): s0 of lane l is stale, and unused -- row l always starts)
                    bool start_ok = false;
                    {
                        const double hmm = d0, hm1m = s1, hm1m1 = dp;
                        double s = fabs(hmm - rt2r) + fabs(rt2i) + fabs(hm1m);
                        double rs = fast_rcp(s);
                        const double h21s = hm1m * rs;
                        v0 = h21s * u1 + (hmm - rt1r) * ((hmm - rt2r) * rs) - rt1i * (rt2i * rs);
                        v1 = h21s * (hmm + hm1m1 - rt1r - rt2r);
                        v2 = h21s * s2;
                        s = fabs(v0) + fabs(v1) + fabs(v2);
                        rs = fast_rcp(s);
                        v0 *= rs;
                        v1 *= rs;
                        v2 *= rs;
                        const bool crit2 = fabs(s0) * (fabs(v1) + fabs(v2)) <= ulp * fabs(v0) * (fabs(dm) + fabs(hmm) + fabs(hm1m1));
                        start_ok = lane >= l && lane <= i - 2 && (lane == l || crit2);
                    }
                    const unsigned long long mmask = __ballot(start_ok);  // bit l is always set
                    m = uniform(63 - __builtin_clzll(mmask));
                    v0 = bcast_lane(v0, m);
                    v1 = bcast_lane(v1, m);
                    v2 = bcast_lane(v2, m);
                }
            } else {
            // ---- look for a single small subdiagonal element: lane k tests H(k,k-1), l < k <= i ----
            bool small = false;
            if (lane > l && lane <= i) {
                const int k = lane;
                const double hkk1 = fabs(H.at(k, k - 1));
                if (hkk1 <= smlnum) {
                    small = true;
                } else {
                    const double hkk = H.at(k, k);
                    const double hk1k1 = H.at(k - 1, k - 1);
                    double tst = fabs(hk1k1) + fabs(hkk);
                    if (tst == 0.0) {
                        if (k - 2 >= 0) tst += fabs(H.at(k - 1, k - 2));
                        if (k + 1 <= n - 1) tst += fabs(H.at(k + 1, k));
                    }
                    if (hkk1 <= ulp * tst) {
                        const double hk1k = fabs(H.at(k - 1, k));
                        const double ab = fmax(hkk1, hk1k);
                        const double ba = fmin(hkk1, hk1k);
                        const double df = fabs(hk1k1 - hkk);
                        const double aa = fmax(fabs(hkk), df);
                        const double bb = fmin(fabs(hkk), df);
                        const double s = aa + ab;
                        if (ba * (ab / s) <= fmax(smlnum, ulp * (bb * (aa / s)))) small = true;
                    }
                }
            }
            const unsigned long long smask = __ballot(small);
            l = smask ? (63 - __builtin_clzll(smask)) : l;
            l = uniform(l);
            if (l > 0 && lane == 0) H.at(l, l - 1) = 0.0;
            if (l >= i - 1) done = true;
            if (!done) {
            wave_lds_sync();
            ++kdefl;

            // ---- shifts (uniform) ----
            double h11, h21, h12, h22;
            if (kdefl % (2 * kexsh) == 0) {
                const double s = fabs(H.at(i, i - 1)) + fabs(H.at(i - 1, i - 2));
                h11 = 0.75 * s + H.at(i, i);
                h12 = -0.4375 * s;
                h21 = s;
                h22 = h11;
            } else if (kdefl % kexsh == 0) {
                const double s = fabs(H.at(l + 1, l)) + fabs(H.at(l + 2, l + 1));
                h11 = 0.75 * s + H.at(l, l);
                h12 = -0.4375 * s;
                h21 = s;
                h22 = h11;
            } else {
                h11 = H.at(i - 1, i - 1);
                h21 = H.at(i, i - 1);
                h12 = H.at(i - 1, i);
                h22 = H.at(i, i);
            }
            {
                const double s = fabs(h11) + fabs(h12) + fabs(h21) + fabs(h22);
                if (s == 0.0) {
                    rt1r = rt1i = rt2r = rt2i = 0.0;
                } else {
                    const double rs = fast_rcp(s);                      // (shifts need no correctly rounded quotients)
                    h11 *= rs;
                    h21 *= rs;
                    h12 *= rs;
                    h22 *= rs;
                    const double tr = (h11 + h22) * 0.5;
                    const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
                    const double rtdisc = sqrt(fabs(det));
                    if (det >= 0.0) {
                        rt1r = tr * s;
                        rt2r = rt1r;
                        rt1i = rtdisc * s;
                        rt2i = -rt1i;
                    } else {
                        rt1r = tr + rtdisc;
                        rt2r = tr - rtdisc;
                        if (fabs(rt1r - h22) <= fabs(rt2r - h22)) {
                            rt1r = rt1r * s;
                            rt2r = rt1r;
                        } else {
                            rt2r = rt2r * s;
                            rt1r = rt2r;
                        }
                        rt1i = 0.0;
                        rt2i = 0.0;
                    }
                }
            }

            // ---- two consecutive small subdiagonals: lane m evaluates the start vector at row m ----
            bool start_ok = false;
            if (lane >= l && lane <= i - 2) {
                const int mm = lane;
                const double hmm = H.at(mm, mm);
                const double hm1m = H.at(mm + 1, mm);
                const double hm1m1 = H.at(mm + 1, mm + 1);
                double s = fabs(hmm - rt2r) + fabs(rt2i) + fabs(hm1m);
                double rs = fast_rcp(s);                                // the start vector is only a direction
                const double h21s = hm1m * rs;
                v0 = h21s * H.at(mm, mm + 1) + (hmm - rt1r) * ((hmm - rt2r) * rs) - rt1i * (rt2i * rs);
                v1 = h21s * (hmm + hm1m1 - rt1r - rt2r);
                v2 = h21s * H.at(mm + 2, mm + 1);
                s = fabs(v0) + fabs(v1) + fabs(v2);
                rs = fast_rcp(s);
                v0 *= rs;
                v1 *= rs;
                v2 *= rs;
                if (mm == l) {
                    start_ok = true;
                } else {
                    const double h00 = fabs(H.at(mm - 1, mm - 1));
                    start_ok = fabs(H.at(mm, mm - 1)) * (fabs(v1) + fabs(v2)) <=
                               ulp * fabs(v0) * (h00 + fabs(hmm) + fabs(hm1m1));
                }
            }
            const unsigned long long mmask = __ballot(start_ok);      // bit l is always set
            m = uniform(63 - __builtin_clzll(mmask));
            v0 = bcast_lane(v0, m);
            v1 = bcast_lane(v1, m);
            v2 = bcast_lane(v2, m);
            }   // !done
            }   // generic layouts
            if (done) {
                converged = true;
                break;
            }

            // ---- double-shift QR sweep: chase the bulge(s) from row m to row i (m <= i-2) ----
#ifndef EGS_QR_SINGLE_BULGE
            if constexpr (V::kFast) {
#ifdef EGS_QR_PROF
                { const long long n_ = qr_now(); P.scan += n_ - tscan; P.nsweep += 1; }
                qr2_sweep(H, l, i, m, lane, v0, v1, v2, rt1r, rt1i, rt2r, rt2i, no_merge, P);
                tscan = qr_now();
#else
                qr2_sweep(H, l, i, m, lane, v0, v1, v2, rt1r, rt1i, rt2r, rt2i, no_merge);
#endif
            } else
#endif
            {
                double a0 = v0, a1 = v1, a2 = v2;
                double* rowp = &H.at(m, lane);
                double* colp = &H.at(lane, m);
                qr_step3<true>(H, rowp, colp, m, l, i, lane, m > l, a0, a1, a2);
                for (int k = m + 1; k <= i - 2; ++k) qr_step3<false>(H, rowp, colp, k, l, i, lane, false, a0, a1, a2);
                qr_step2(H, i - 1, l, i, lane, a0, a1);
            }
        }
        if (!converged) {
            info = i + 1;
            break;
        }
        if (l == i) {
            const double wr = H.at(i, i) * rescale;
            if (lane == 0) {
                w_out[2 * i] = wr;
                w_out[2 * i + 1] = 0.0;
            }
        } else {   // l == i-1: a 2x2 block
            double rt1r, rt1i, rt2r, rt2i;
            lanv2_eig(H.at(i - 1, i - 1), H.at(i - 1, i), H.at(i, i - 1), H.at(i, i), rt1r, rt1i, rt2r, rt2i);
            if (lane == 0) {
                w_out[2 * (i - 1)] = rt1r * rescale;
                w_out[2 * (i - 1) + 1] = rt1i * rescale;
                w_out[2 * i] = rt2r * rescale;
                w_out[2 * i + 1] = rt2i * rescale;
            }
        }
        kdefl = 0;
        i = l - 1;
        wave_lds_sync();
    }
    if (info != 0) {   // eigenvalues 0..info-1 did not converge
        for (int j = lane; j < info; j += kWave) {
            w_out[2 * j] = qnan;
            w_out[2 * j + 1] = qnan;
        }
    }
    if (lane == 0) *info_out = info;
    const int rem = (info == 0 && i >= 0) ? i + 1 : 0;
    QRP(if (blockIdx.x == 0 && threadIdx.x == 0) printf("QRPROF total %lld scan %lld setup %lld gen %lld (%lld) fast %lld (%lld) sweeps %lld\n",
                                                          qr_now() - tstart, P.scan, P.setup, P.gen, P.ngen, P.fast, P.nfast, P.nsweep);)
    return rem;
}

// Packed-pair LDS layout of the QR kernel: array of n rows x ldp columns per pair.
//   role 0: H(i,j) at row i, column j+7   (occupied: column >= row+4, since i <= j+3 during a chase)
//   role 1: H(i,j) at row j, column i     (occupied: column <= row+3)
__host__ __device__ inline int hqr_pair_ld(int n) {
    if (n == 64) return kLd64;
    int ld = n + 7;
    return (ld & 1) ? ld : ld + 1;
}

template <class V>
__device__ __forceinline__ void hqr_run(const V H, const double* __restrict__ src, int n, int lane,
                                        double* __restrict__ wo, int32_t* __restrict__ info, double rescale,
                                        const bool normalise, const int no_merge) {
    // load the Hessenberg part plus the two extra sub-diagonals a bulge may touch (zeros); 8 rows of global loads in
    // flight at a time, always from valid addresses (clamped lane), zeroed / dropped by value afterwards
    const int lcl = (lane < n) ? lane : n - 1;
    double amax = 0.0;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(i + u) * n + lcl];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (lane < n && i + u <= lane + 3) {
                const double v = (i + u <= lane + 1) ? t[u] : 0.0;
                H.at(i + u, lane) = v;
                amax = fmax(amax, fabs(v));
            }
    }
    for (; i < n; ++i) {
        const double t = src[(int64_t)i * n + lcl];
        if (lane < n && i <= lane + 3) {
            const double v = (i <= lane + 1) ? t : 0.0;
            H.at(i, lane) = v;
            amax = fmax(amax, fabs(v));
        }
    }
    wave_lds_sync();
    if (normalise) {
        // The n = 64 layouts deflate and guard with ABSOLUTE thresholds that assume max |entry| in [1, 2) (the eigenvalue
        // driver's Hessenberg stage scales so).  A Hessenberg matrix handed in through egs_hseqr_batched comes as the
        // caller made it: scale it here by a power of two (exact) and multiply the eigenvalues back.
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
        if (amax > 0.0 && amax <= 1.7976931348623157e308) {
            const int e = ilogb(amax);
            if (e != 0) {
                for (int r = 0; r < n; ++r)
                    if (lane < n && r <= lane + 1) H.at(r, lane) = scalbn(H.at(r, lane), -e);
                wave_lds_sync();
                rescale = scalbn(rescale, e);
            }
        }
    }
    hqr_wave(H, n, lane, wo, info, rescale, 0, 0, no_merge);
}

// N64 = true: n == 64, only the two compile-time layouts (two-bulge sweep); false: any n < 64 (run-time leading dimension,
// single-bulge sweep).  Two kernels instead of four inlined copies of hqr_wave in one: the scalar state of each fits
// the SGPR file (the single kernel of rounds 1-2 spilled 257 SGPRs).
#ifndef EGS_HQR_MIN_WAVES_PER_EU
#define EGS_HQR_MIN_WAVES_PER_EU 1
#endif
template <bool N64>
__global__ __launch_bounds__(512, EGS_HQR_MIN_WAVES_PER_EU) void hqr_real_kernel(const double* __restrict__ hin, int64_t nmat, int n,
                                                       double* __restrict__ w, int32_t* __restrict__ info,
                                                       int waves_per_block, int scaled,
                                                       unsigned long long* __restrict__ next_mat, int no_merge) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int ldp = hqr_pair_ld(n);
    double* base = smem + (size_t)(wave >> 1) * ((size_t)n * ldp);
    // persistent waves with a work queue: the launch fills the chip exactly once (2 workgroups per CU); a wave that has
    // finished its matrix takes the next unclaimed one from a global counter, so nobody waits for the slowest matrix of
    // a workgroup and the kernel's tail is one matrix long.  (A static grid-stride assignment measured 4 % SLOWER than
    // plain one-matrix-per-wave workgroups: the unluckiest of 2048 static partitions sets the kernel time.)  No
    // workgroup barriers are used and a wave reloads only its own half of the pair array.  Every wave leaves the loop
    // when the counter passes nmat.  next_mat == null: one matrix per wave (the grid covers all matrices).
    const int64_t first = (int64_t)blockIdx.x * waves_per_block + wave;
    const int64_t stride0 = (int64_t)gridDim.x * waves_per_block;
    for (int64_t mat = first; mat < nmat;) {
        const double* src = hin + mat * (int64_t)n * n;
        double* wo = w + mat * 2 * (int64_t)n;
        // when the Hessenberg stage scaled the matrix, info[mat] carries the power-of-two exponent to undo
        if (scaled && info[mat] == kNonFinite) {     // like numpy.linalg.eigvals, non-finite input is an error, not a spectrum
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            for (int j = lane; j < 2 * n; j += kWave) wo[j] = qnan;
            if (lane == 0) info[mat] = n;
        } else {
        const double rescale = scaled ? scalbn(1.0, info[mat]) : 1.0;
        if constexpr (N64) {
            if ((wave & 1) == 0) hqr_run(MatViewT<kLd64, false>{base + 7, ldp}, src, n, lane, wo, info + mat, rescale, !scaled, no_merge);
            else hqr_run(MatViewT<kLd64, true>{base, ldp}, src, n, lane, wo, info + mat, rescale, !scaled, no_merge);
        } else {
            if ((wave & 1) == 0) hqr_run(MatViewT<0, false>{base + 7, ldp}, src, n, lane, wo, info + mat, rescale, false, no_merge);
            else hqr_run(MatViewT<0, true>{base, ldp}, src, n, lane, wo, info + mat, rescale, false, no_merge);
        }
        }
        wave_lds_sync();                                          // this matrix's LDS traffic is complete before the next load
        if (next_mat == nullptr) break;
        unsigned long long nx = 0ULL;
        if (lane == 0) nx = atomicAdd(next_mat, 1ULL);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(nx & 0xffffffffULL));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(nx >> 32));
        mat = stride0 + (int64_t)(((unsigned long long)hi << 32) | lo);
    }
}


// ---------------------------------------------------------------------------------------------------
// n = 64, both stages in ONE kernel: each resident wave takes a matrix from the queue, reduces it to Hessenberg form
// in registers (hess_reg.h: no LDS traffic for the matrix, VALU / MFMA work), drops the result into its half of the
// pair array and runs the QR iteration there.  The Hessenberg matrix never touches HBM, the whole batch is one launch
// (one tail instead of one per 64 Ki-matrix chunk), and because the waves of a SIMD are at different points of their
// matrices the VALU-dense reduction of one wave fills issue slots that the latency-bound QR sweep of its neighbour
// leaves empty.  Only the two n = 64 layouts are instantiated here.
// ---------------------------------------------------------------------------------------------------
constexpr int kFusedWaves = 4;                                   // 2 pairs per workgroup, 2 workgroups per CU
constexpr int kFusedPairDoubles = 64 * kLd64;                    // hqr_pair_ld(64)
constexpr size_t kFusedLds = (size_t)(2 * kFusedPairDoubles + kFusedWaves * kRegVec) * sizeof(double);

// ---------------------------------------------------------------------------------------------------
// Two phases (n = 64).  A 64 x 64 Hessenberg matrix takes 18.7 KB of LDS, so only two waves fit a SIMD, and at two per
// SIMD a wave spends half its life waiting on its own dependent chain (profiles/r03_rejected_sweep_variants.md: the
// occupancy experiment).  But the iteration deflates from the bottom: once the eigenvalues at positions >= kTailN
// are out, what is left is a kTailN x kTailN problem that needs (kTailN / 64)^2 of the LDS -- and it is about half of the
// work (an iteration costs the same whatever the window's length; their number goes with the window's end).  So the
// fused kernel stops a matrix when its active block ends below row kTailN, writes that block (with the rounding-level
// entries a chase leaves on the two diagonals below the sub-diagonal: the continuation is bit-identical to the
// one-kernel iteration) plus its order and scaling exponent to a hand-off record in the workspace, and
// hqr_tail_kernel finishes all of them at twice the occupancy (kTailN = 44: 16 waves per CU).
// Hand-off record: {order of the remaining block (0 = nothing left), exponent}, then the packed rows (see kTailRec).
// ---------------------------------------------------------------------------------------------------
#ifndef EGS_TAIL_N
#define EGS_TAIL_N 44
#endif
constexpr int kTailN = EGS_TAIL_N;
static_assert(kTailN % 2 == 0 && kTailN >= 34 && kTailN <= 56, "tail order: even (odd leading dimension), > 32 + bulge distance");
constexpr int kTailLd = kTailN + 9;                              // odd; diagonal stride kTailLd + 1 spreads over >= 16 banks
// Hand-off record, PACKED (round 5): {order of the remaining block, scaling exponent} and then, row by row, only what a
// continuation reads -- row r holds columns max(0, r - 3) .. kTailN - 1 (the Hessenberg part and the two diagonals below
// the sub-diagonal).  1118 doubles (8.9 KB) at kTailN = 44 instead of the 44 x 44 + 2 = 1938 (15.5 KB) of rounds 3-4:
// the C2 workspace drops from 8.2 to 4.8 GB and the records' HBM traffic with it.
__host__ __device__ constexpr int tail_row_off(int r) {
    // doubles in front of row r: rows 0..3 are full, row r' >= 4 starts at column r' - 3
    return r <= 4 ? r * kTailN : 4 * kTailN + (r - 4) * (kTailN + 3) - (r * (r - 1) / 2 - 6);
}
constexpr int kTailHdr = 2;
constexpr int kTailRec = kTailHdr + tail_row_off(kTailN);        // doubles per hand-off record
static_assert(tail_row_off(5) - tail_row_off(4) == kTailN - 1 && tail_row_off(4) == 4 * kTailN, "packed record offsets");
#ifndef EGS_TAIL_WAVES
#define EGS_TAIL_WAVES 4
#endif
constexpr int kTailWaves = EGS_TAIL_WAVES;                       // 2 pairs per workgroup
constexpr int kTailPairDoubles = kTailN * kTailLd;
constexpr size_t kTailLds = (size_t)(kTailWaves / 2 * kTailPairDoubles) * sizeof(double) + 1024;   // + the order-2 step's over-read

// Only the part of the record a continuation reads is moved: rows r <= c + 3 of columns c < order (the Hessenberg part
// and the two diagonals below it) -- about half of the order^2 block, ~9 KB per matrix each way.
template <class V>
__device__ __forceinline__ void tail_handoff_write(const V H, const int rem, const int e, double* __restrict__ rec,
                                                   const int lane) {
    const int c = (lane < kTailN) ? lane : kTailN - 1;
    for (int r = 0; r < rem; ++r) {
        const double v = H.at(r, c);
        const int c0 = (r > 3) ? r - 3 : 0;
        if (lane < rem && r <= lane + 3) rec[kTailHdr + tail_row_off(r) + lane - c0] = v;
    }
    if (lane == 0) {
        rec[0] = (double)rem;
        rec[1] = (double)e;
    }
}

template <class V>
__device__ __forceinline__ void tail_handoff_read(const V H, const int rem, const double* __restrict__ rec, const int lane) {
    int r = 0;
    for (; r + 4 <= rem; r += 4) {
        double t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c0 = (r + u > 3) ? r + u - 3 : 0;
            t[u] = (lane < rem && r + u <= lane + 3) ? rec[kTailHdr + tail_row_off(r + u) + lane - c0] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane < rem && r + u <= lane + 3) H.at(r + u, lane) = t[u];
    }
    for (; r < rem; ++r) {
        const int c0 = (r > 3) ? r - 3 : 0;
        if (lane < rem && r <= lane + 3) H.at(r, lane) = rec[kTailHdr + tail_row_off(r) + lane - c0];
    }
}

__global__ __launch_bounds__(kTailWaves* kWave, 4) void hqr_tail_kernel(const double* __restrict__ handoff, int64_t nmat,
                                                                         double* __restrict__ w, int32_t* __restrict__ info,
                                                                         unsigned long long* __restrict__ next_mat,
                                                                         int no_merge) {
    extern __shared__ __attribute__((aligned(16))) double smem_tail[];
    const int lane0 = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    double* base = smem_tail + (size_t)(wave >> 1) * kTailPairDoubles;
    const int64_t first = (int64_t)blockIdx.x * kTailWaves + wave;
    const int64_t stride0 = (int64_t)gridDim.x * kTailWaves;
    for (int64_t mat = first; mat < nmat;) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const double* rec = handoff + mat * (int64_t)kTailRec;
        const int rem = __builtin_amdgcn_readfirstlane((int)rec[0]);
        if (rem > 0 && rem <= kTailN) {
            const int e = __builtin_amdgcn_readfirstlane((int)rec[1]);
            const double rescale = (e != 0) ? scalbn(1.0, e) : 1.0;
            double* wo = w + mat * 128;
            if ((wave & 1) == 0) {
                const MatViewT<kTailLd, false, kTailN> H{base + 7, kTailLd};
                tail_handoff_read(H, rem, rec, lane);
                wave_lds_sync();
                hqr_wave(H, rem, lane, wo, info + mat, rescale, 0, 64, no_merge);
            } else {
                const MatViewT<kTailLd, true, kTailN> H{base, kTailLd};
                tail_handoff_read(H, rem, rec, lane);
                wave_lds_sync();
                hqr_wave(H, rem, lane, wo, info + mat, rescale, 0, 64, no_merge);
            }
        }
        wave_lds_sync();
        if (next_mat == nullptr) break;
        unsigned long long nx = 0ULL;
        if (lane == 0) nx = atomicAdd(next_mat, 1ULL);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(nx & 0xffffffffULL));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(nx >> 32));
        mat = stride0 + (int64_t)(((unsigned long long)hi << 32) | lo);
    }
}

template <class V>
__device__ __forceinline__ void store_hessenberg(const V H, const double (&A)[16][4], const RegLane& L) {
    // the Hessenberg part plus the two extra sub-diagonals a bulge may touch (zeros)
#pragma unroll
    for (int R = 0; R < 16; ++R)
#pragma unroll
        for (int C = 0; C < 4; ++C) {
            const int row = 4 * R + L.q, col = 16 * C + L.c;
            if (row <= col + 3) H.at(row, col) = (row <= col + 1) ? A[R][C] : 0.0;
        }
}

__global__ __launch_bounds__(kFusedWaves* kWave, 2) void eig_real64_kernel(
    const double* __restrict__ a, int64_t nmat, int lda, int64_t stride, double* __restrict__ w,
    int32_t* __restrict__ info, double* __restrict__ scratch, int balance, unsigned long long* __restrict__ next_mat,
    double* __restrict__ handoff, int no_merge) {
    extern __shared__ __attribute__((aligned(16))) double smem_fused[];
    constexpr int n = 64, ldp = kLd64;
    const int lane0 = threadIdx.x & (kWave - 1);
    // wave-uniform BY CONSTRUCTION, which the compiler cannot see through threadIdx: as a scalar, everything derived from
    // it (this wave's LDS halves, its scratch slot, its first matrix) lives in SGPRs instead of being spilled as
    // per-lane values across the register-hungry reduction
#ifndef EGS_WAVE_VGPR
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
#else
    const int wave = threadIdx.x / kWave;                           // rounds 2-4 (A/B: profiles/r05_variants.md)
#endif
    double* base = smem_fused + (size_t)(wave >> 1) * kFusedPairDoubles;
    double* vecs = smem_fused + 2 * kFusedPairDoubles + (size_t)wave * kRegVec;
    const int64_t first = (int64_t)blockIdx.x * kFusedWaves + wave;
    const int64_t stride0 = (int64_t)gridDim.x * kFusedWaves;
    double* my_scratch = scratch + first * (int64_t)(n * n);     // permutation path of the balancing only
    for (int64_t mat = first; mat < nmat;) {
        // Everything both stages derive from the lane index (row / column indices, masks, LDS addresses) is invariant
        // across matrices; hoisted out of this loop it would have to stay alive -- spilled -- across the register-hungry
        // reduction.  An opaque per-matrix copy keeps those derivations (a few dozen integer instructions) inside the
        // loop body.  (Only the integer: an opaque POINTER loses its LDS address space and turns every ds_ access into
        // a flat one -- measured 1.7x slower.)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        RegLane L;
        reg_lane_init(L, lane);
        double* wo = w + mat * 2 * (int64_t)n;
        double A[16][4];
        // (wave priorities were tried both ways round -- reduction below / above the QR sweep: 2 - 3 % slower than none)
        const int e = hess_reg_reduce(A, L, a + mat * stride, lda, vecs, my_scratch, true, balance);
        if (e == kNonFinite) {        // like numpy.linalg.eigvals, non-finite input is an error, not a spectrum
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            for (int j = lane; j < 2 * n; j += kWave) wo[j] = qnan;
            if (lane == 0) {
                info[mat] = n;
                if (handoff != nullptr) handoff[mat * (int64_t)kTailRec] = 0.0;
            }
        } else {
            const double rescale = (e != 0) ? scalbn(1.0, e) : 1.0;
            const int stop_below = (handoff != nullptr) ? kTailN : 0;
            if ((wave & 1) == 0) {
                const MatViewT<kLd64, false> H{base + 7, ldp};
                store_hessenberg(H, A, L);
                wave_lds_sync();
                const int rem = hqr_wave(H, n, lane, wo, info + mat, rescale, stop_below, 0, no_merge);
                if (handoff != nullptr) tail_handoff_write(H, rem, e, handoff + mat * (int64_t)kTailRec, lane);
            } else {
                const MatViewT<kLd64, true> H{base, ldp};
                store_hessenberg(H, A, L);
                wave_lds_sync();
                const int rem = hqr_wave(H, n, lane, wo, info + mat, rescale, stop_below, 0, no_merge);
                if (handoff != nullptr) tail_handoff_write(H, rem, e, handoff + mat * (int64_t)kTailRec, lane);
            }
        }
        wave_lds_sync();                                          // this matrix's LDS traffic is complete before the next one
        if (next_mat == nullptr) break;
        unsigned long long nx = 0ULL;
        if (lane == 0) nx = atomicAdd(next_mat, 1ULL);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(nx & 0xffffffffULL));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(nx >> 32));
        mat = stride0 + (int64_t)(((unsigned long long)hi << 32) | lo);
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------------------------
// waves of the fused kernel's grid: one resident set (two workgroups per CU), fewer for small batches
static int64_t fused_resident_waves(int64_t nmat) {
    int64_t blocks = (nmat + kFusedWaves - 1) / kFusedWaves;
    const int64_t resident = (int64_t)current_device_cus() * (int64_t)((160 * 1024) / kFusedLds);
    if (blocks > resident) blocks = resident;
    return blocks * kFusedWaves;
}

size_t eig_real_lds_workspace_bytes(int64_t nmat, int n) {
    // Hessenberg matrices of one chunk.  64 Ki matrices per launch (2 GiB at n = 64): every launch ends with a tail in
    // which the slowest matrices keep a few CUs busy; 4x fewer launches than with 16 Ki chunks measured 2.6 % faster
    const int64_t chunk = nmat < 65536 ? nmat : 65536;
    size_t b = (size_t)chunk * n * n * sizeof(double) + 256;      // + the work-queue counter of the QR kernel
    if (n == 64 && nmat >= kFusedWaves) {
        // two-phase solver: one scratch slot per resident wave, one hand-off record per matrix (up to 2^20 of them per
        // pass: 16 GB; larger batches run in several passes), two work-queue counters
        const int64_t waves = fused_resident_waves(nmat);
        const int64_t recs = nmat < (1 << 20) ? nmat : (1 << 20);
        const size_t two = (size_t)waves * n * n * sizeof(double) + (size_t)recs * kTailRec * sizeof(double) + 256;
        if (two > b) b = two;
    }
    return b;
}

static int set_lds_attrs() {
    EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hess_real_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hqr_real_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hqr_real_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

int hess_reg_launch(const double* a, int64_t nmat, int lda, int64_t stride, double* h, int32_t* scale_exp, int balance,
                    hipStream_t stream);          // eig_real_reg.hip: n = 64 with the matrix in registers

int hess_real_launch(const double* a, int64_t nmat, int n, int lda, int64_t stride, double* h, int32_t* scale_exp,
                     hipStream_t stream) {
    if (nmat == 0 || n == 0) return 0;
    int st = set_lds_attrs();
    if (st) return st;
    const size_t hess_lds = (size_t)kHessWaves * kHessStride * sizeof(double);
    const unsigned gh = (unsigned)((nmat + kHessWaves - 1) / kHessWaves);
    // balancing (xGEBAL) belongs to the eigenvalue driver: the stand-alone Hessenberg entry point (scale_exp == null)
    // returns a matrix orthogonally similar to its input, without it
    const int balance = (scale_exp != nullptr && !getenv("EGS_NO_BALANCE")) ? 1 : 0;
    if (n == 64 && !getenv("EGS_HESS_LDS"))       // A/B switch: the LDS-resident kernel below
        return hess_reg_launch(a, nmat, lda, stride, h, scale_exp, balance, stream);
    hipLaunchKernelGGL(hess_real_kernel, dim3(gh), dim3(kHessWaves * kWave), hess_lds, stream, a, nmat, n, lda, stride,
                       h, scale_exp, balance);
    EGS_LAUNCH_CHECK();
    return 0;
}

// A/B switch, read at launch: the three-pass n = 64 sweep instead of the merged two-pass one (bit-identical results)
static int qr2_no_merge() { return getenv("EGS_QR2_NO_MERGE") ? 1 : 0; }

int hqr_real_launch(const double* h, int64_t nmat, int n, double* w, int32_t* info, int scaled, hipStream_t stream,
                    unsigned long long* queue) {
    if (nmat == 0 || n == 0) return 0;
    int st = set_lds_attrs();
    if (st) return st;
    const int ldp = hqr_pair_ld(n);
    const size_t pair_bytes = (size_t)n * ldp * sizeof(double);
    int pairs = (int)((160 * 1024) / pair_bytes);
    // 2 pairs (4 waves, 73 KB) per workgroup: two workgroups share a CU, so 8 matrices stay resident per CU while a
    // workgroup only waits for the slowest of 4 (measured 6 % faster than one 8-wave workgroup per CU)
    if (pairs > 2) pairs = 2;
    if (pairs < 1) pairs = 1;
    if (const char* e = getenv("EGS_HQR_PAIRS")) {          // tuning knob (occupancy experiments)
        const int v = atoi(e);
        if (v >= 1 && v <= 4 && (size_t)v * pair_bytes <= 160 * 1024) pairs = v;
    }
    const int hqr_waves = 2 * pairs;
    // + 1 KB: the order-2 step at the bottom of a 64-row window addresses row / column 64 of its (unused) third operand,
    // up to 70 doubles past the last pair array; reads only, the values are discarded
    size_t hqr_lds = pairs * pair_bytes + 1024;
    if (const char* e = getenv("EGS_HQR_LDS_PAD")) {        // occupancy experiments: inflate the request -> fewer workgroups per CU
        const long v = atol(e);
        if (v > 0 && hqr_lds + (size_t)v <= 160 * 1024) hqr_lds += (size_t)v;
    }
    unsigned gq = (unsigned)((nmat + hqr_waves - 1) / hqr_waves);
    if (queue != nullptr && !getenv("EGS_HQR_NONPERSISTENT")) {   // one resident wave set + work queue
        const int cus = current_device_cus();
        const unsigned resident = (unsigned)cus * (unsigned)((160 * 1024) / hqr_lds);
        if (gq > resident) {
            gq = resident;
            EGS_HIP(hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream));
        } else {
            queue = nullptr;
        }
    } else {
        queue = nullptr;
    }
    if (n == 64)
        hipLaunchKernelGGL(hqr_real_kernel<true>, dim3(gq), dim3(hqr_waves * kWave), hqr_lds, stream, h, nmat, n, w, info,
                           hqr_waves, scaled, queue, qr2_no_merge());
    else
        hipLaunchKernelGGL(hqr_real_kernel<false>, dim3(gq), dim3(hqr_waves * kWave), hqr_lds, stream, h, nmat, n, w, info,
                           hqr_waves, scaled, queue, 0);
    EGS_LAUNCH_CHECK();
    return 0;
}

int eig_real_lds_launch(const double* a, int64_t nmat, int n, int lda, int64_t stride, double* w, int32_t* info,
                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (nmat == 0 || n == 0) return 0;
    const size_t per_mat = (size_t)n * n * sizeof(double);
    int64_t chunk = workspace_bytes > 256 ? (int64_t)((workspace_bytes - 256) / per_mat) : 0;
    if (chunk > nmat) chunk = nmat;
    if (chunk < 1) {
        set_error("workspace too small: %zu bytes, need at least %zu", workspace_bytes, per_mat + 256);
        return -9;
    }
    double* hws = static_cast<double*>(workspace);
    if (n == 64 && chunk >= kFusedWaves && !getenv("EGS_EIG_UNFUSED")) {     // A/B switch: the two-kernel path below
        EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(eig_real64_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hqr_tail_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const int cus = current_device_cus();
        const int balance = getenv("EGS_NO_BALANCE") ? 0 : 1;
        // Two phases when the workspace holds a scratch slot per wave of the grid plus hand-off records for a useful
        // number of matrices (egs_eig_workspace_bytes asks for all of them); otherwise the one-kernel form.
        const int64_t waves2 = fused_resident_waves(nmat);
        const size_t slots2 = (size_t)waves2 * per_mat, rec_bytes = (size_t)kTailRec * sizeof(double);
        int64_t pass = workspace_bytes > slots2 + 256 ? (int64_t)((workspace_bytes - slots2 - 256) / rec_bytes) : 0;
        if (pass > nmat) pass = nmat;
        const int64_t pass_min = nmat < 32768 ? nmat : 32768;
        if (pass >= pass_min && !getenv("EGS_EIG_ONE_PHASE")) {
            char* wsb = reinterpret_cast<char*>(workspace);
            double* hand = reinterpret_cast<double*>(wsb + slots2);
            unsigned long long* queues = reinterpret_cast<unsigned long long*>(wsb + slots2 + (size_t)pass * rec_bytes);
            const int64_t tail_resident = (int64_t)cus * (int64_t)((160 * 1024) / kTailLds);
            for (int64_t m0 = 0; m0 < nmat; m0 += pass) {
                const int64_t cnt = (nmat - m0 < pass) ? (nmat - m0) : pass;
                int64_t blocks = (cnt + kFusedWaves - 1) / kFusedWaves;
                if (blocks * kFusedWaves > waves2) blocks = waves2 / kFusedWaves;
                int64_t tblocks = (cnt + kTailWaves - 1) / kTailWaves;
                if (tblocks > tail_resident) tblocks = tail_resident;
                EGS_HIP(hipMemsetAsync(queues, 0, 2 * sizeof(unsigned long long), stream));
                hipLaunchKernelGGL(eig_real64_kernel, dim3((unsigned)blocks), dim3(kFusedWaves * kWave), kFusedLds, stream,
                                   a + m0 * stride, cnt, lda, stride, w + m0 * 2 * (int64_t)n, info + m0, hws, balance,
                                   queues, hand, qr2_no_merge());
                EGS_LAUNCH_CHECK();
                hipLaunchKernelGGL(hqr_tail_kernel, dim3((unsigned)tblocks), dim3(kTailWaves * kWave), kTailLds, stream,
                                   hand, cnt, w + m0 * 2 * (int64_t)n, info + m0, queues + 1, qr2_no_merge());
                EGS_LAUNCH_CHECK();
            }
            return 0;
        }
        // one resident set of waves (two workgroups per CU) fed by a work queue; every wave owns one scratch slot of the
        // workspace, so the grid never holds more waves than the workspace has slots
        int64_t blocks = (nmat + kFusedWaves - 1) / kFusedWaves;
        const int64_t resident = (int64_t)cus * (int64_t)((160 * 1024) / kFusedLds);
        unsigned long long* queue = nullptr;
        if (blocks > resident || blocks * kFusedWaves > chunk) {
            if (blocks > resident) blocks = resident;
            if (blocks * kFusedWaves > chunk) blocks = chunk / kFusedWaves;
            queue = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + (size_t)chunk * per_mat);
            EGS_HIP(hipMemsetAsync(queue, 0, sizeof(unsigned long long), stream));
        }
        hipLaunchKernelGGL(eig_real64_kernel, dim3((unsigned)blocks), dim3(kFusedWaves * kWave), kFusedLds, stream, a, nmat,
                           lda, stride, w, info, hws, balance, queue, static_cast<double*>(nullptr), qr2_no_merge());
        EGS_LAUNCH_CHECK();
        return 0;
    }
    for (int64_t m0 = 0; m0 < nmat; m0 += chunk) {
        const int64_t cnt = (nmat - m0 < chunk) ? (nmat - m0) : chunk;
        int st = hess_real_launch(a + m0 * stride, cnt, n, lda, stride, hws, info + m0, stream);
        if (st) return st;
        st = hqr_real_launch(hws, cnt, n, w + m0 * 2 * (int64_t)n, info + m0, 1, stream,
                             reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + (size_t)chunk * per_mat));
        if (st) return st;
    }
    return 0;
}

}  // namespace egs
