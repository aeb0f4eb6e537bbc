// S4 fast path that uses the structure (optional analysis key `s4_solver: dplr`; review r4 #5).
//
// discrete_DPLR's Ab (analysis/eval_eig.py:254-273) is the Cayley transform of A = diag(Lambda) - P Q^H:
//     Ab = (g I - A)^-1 (g I + A),  g = 2 / step        (A1 of :270 is the Woodbury form of (g I - A)^-1)
// so eig(Ab) = (g + a) / (g - a) for a in eig(A) (the Cayley known answer), and `a` does not depend
// on the step: feature columns that share (Lambda, P, Q) -- every column at initialisation -- need ONE solve.
// A is diagonal plus rank one, so its eigenvalues are the roots of the secular equation
//     f(a) = 1 - sum_k W_k / (Lambda_k - a) = 0,     W_k = conj(Q_k) P_k,
// (det(A - a I) = f(a) prod_k (Lambda_k - a)), found here by the Ehrlich-Aberth iteration on p(a) = f(a) prod (Lambda_k - a):
//     p'/p = f'/f - sum_k 1/(Lambda_k - a),   a_i <- a_i - 1 / (p'/p - sum_{j != i} 1/(a_i - a_j)),
// all roots at once (Jacobi style), O(n^2) per iteration instead of the dense solver's O(n^3) per matrix.
// Trivial eigenvalues are split off first: a pole with weight zero, and every pole that repeats an earlier one exactly
// (its weight moves to the first occurrence), is an eigenvalue Lambda_k and leaves both sums.
// Accuracy: f is a sum of n terms each computed with a few roundings, so a root accepted with |f| <= tol * (1 + sum |W_k /
// (Lambda_k - a)|) is an EXACT eigenvalue of diag(Lambda) - P~ Q^H with |P~_k - P_k| <= O(n ulp) |P_k| elementwise --
// the backward-error sense in which the dense path (and LAPACK) are accurate, root by root.  For ill-conditioned spectra
// (raw HiPPO, N >= 64) the two paths therefore agree by backward error, not eigenvalue by eigenvalue -- neither is
// "the" spectrum there (HiPPO caveat).  The dense solver stays the default and the checker.
//
// One workgroup per (Lambda, P, Q) set, thread t owns roots t, t + blockDim, ...; poles, weights and the current
// roots live in LDS and are read as broadcasts.  fp64 VALU bound: 2 n complex reciprocals per root and iteration.
#include "egs_common.h"

namespace egs {

namespace {

struct zc {
    double re, im;
};
__device__ __forceinline__ zc zsub(zc a, zc b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ zc zadd(zc a, zc b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ zc zmul(zc a, zc b) { return {fma(a.re, b.re, -a.im * b.im), fma(a.re, b.im, a.im * b.re)}; }

constexpr double kOnPole = 1e-290;      // |Lambda_k - a|^2 below this: a IS the pole to working precision
constexpr int kMaxIter = 400;

// a / b, moderate magnitudes (the inputs are S4 parameters: O(1) .. O(n))
__device__ __forceinline__ zc zdiv(zc a, zc b) {
    const double den = fma(b.re, b.re, b.im * b.im);
    const double inv = 1.0 / den;
    return {fma(a.re, b.re, a.im * b.im) * inv, fma(a.im, b.re, -a.re * b.im) * inv};
}

}  // namespace

// lambda, p, q: [nmat, n] as separate real / imaginary planes; a_out: complex128 [nmat, n]; info[m] = number of roots
// that did not converge within kMaxIter iterations (0 = ok).  Dynamic LDS: (8 n) doubles + n ints.
__global__ __launch_bounds__(256) void dplr_secular_kernel(const double* __restrict__ lam_re,
                                                          const double* __restrict__ lam_im,
                                                          const double* __restrict__ p_re,
                                                          const double* __restrict__ p_im,
                                                          const double* __restrict__ q_re,
                                                          const double* __restrict__ q_im, int n, int clip,
                                                          double* __restrict__ a_out, int32_t* __restrict__ info) {
    extern __shared__ double lds_s[];
    zc* pole = reinterpret_cast<zc*>(lds_s);          // Lambda_k
    zc* wgt = pole + n;                               // W_k (combined over exact repeats; 0 for removed poles)
    zc* cur = wgt + n;                                // roots of this iteration
    zc* nxt = cur + n;                                // roots of the next one
    int* active = reinterpret_cast<int*>(nxt + n);    // 1: pole k / root k take part in the iteration
    const int64_t m = blockIdx.x;
    const int tid = threadIdx.x, nth = blockDim.x;
    const double tol = 8.0 * 2.220446049250313e-16;

    for (int k = tid; k < n; k += nth) {
        const double lre = lam_re[m * n + k];
        pole[k] = {clip ? fmin(lre, -1e-4) : lre, lam_im[m * n + k]};          // the caller's clip, eval_eig.py:288
        const zc P = {p_re[m * n + k], p_im[m * n + k]};
        const zc Qc = {q_re[m * n + k], -q_im[m * n + k]};
        nxt[k] = zmul(Qc, P);                                                   // raw weight, combined below
    }
    __syncthreads();
    // exact repeats of a pole: the first occurrence carries the combined weight, the others are eigenvalues as they stand
    for (int k = tid; k < n; k += nth) {
        const zc L = pole[k];
        int leader = k;
        zc W = {0.0, 0.0};
        for (int j = 0; j < n; ++j) {
            const zc Lj = pole[j];
            const bool same = (Lj.re == L.re) && (Lj.im == L.im);
            if (same && j < leader) leader = j;
            if (same) W = zadd(W, nxt[j]);
        }
        const bool act = (leader == k) && (W.re != 0.0 || W.im != 0.0);
        wgt[k] = act ? W : zc{0.0, 0.0};
        active[k] = act ? 1 : 0;
        // start: first-order eigenvalue of the rank-one update (sum of the starts = trace of A)
        cur[k] = act ? zsub(L, W) : L;
    }
    __syncthreads();
    // two starts that coincide exactly would move together for ever: nudge the later one (never seen on S4 parameters)
    for (int k = tid; k < n; k += nth) {
        zc a = cur[k];
        if (active[k]) {
            int clash = 0;
            for (int j = 0; j < k; ++j) clash += (active[j] && cur[j].re == a.re && cur[j].im == a.im) ? 1 : 0;
            if (clash) {
                const double s = 1.0 + 1.4901161193847656e-08 * (double)(k + 1);
                a = {a.re * s + 1e-300 * (k + 1), a.im * s};
            }
        }
        nxt[k] = a;
    }
    __syncthreads();
    for (int k = tid; k < n; k += nth) cur[k] = nxt[k];
    __syncthreads();

    // per-thread convergence flags of the roots this thread owns (n <= 2048, 256 threads: at most 8 roots per thread)
    unsigned done_mask = 0u;
    {
        int b = 0;
        for (int i = tid; i < n; i += nth, ++b)
            if (!active[i]) done_mask |= 1u << b;
    }
    int it = 0;
    for (; it < kMaxIter; ++it) {
        int open = 0, b = 0;
        for (int i = tid; i < n; i += nth, ++b) {
            const zc a = cur[i];
            if (done_mask & (1u << b)) {
                nxt[i] = a;
                continue;
            }
            zc s0 = {0.0, 0.0}, s1 = {0.0, 0.0}, s2 = {0.0, 0.0}, rr = {0.0, 0.0};
            double mag = 1.0;
            bool on_pole = false;
#pragma unroll 2
            for (int k = 0; k < n; ++k) {
                const zc d = zsub(pole[k], a);
                const zc w = wgt[k];
                const int act = active[k];
                const double den = fma(d.re, d.re, d.im * d.im);
                on_pole = on_pole || (act && den < kOnPole);
                const double inv = act ? fast_rcp(fmax(den, kOnPole)) : 0.0;
                const zc r = {d.re * inv, -d.im * inv};
                const zc t = zmul(w, r);
                s0 = zadd(s0, r);
                s1 = zadd(s1, t);
                s2 = zadd(s2, zmul(t, r));
                mag += fabs(t.re) + fabs(t.im);
                // the other roots (same index space: root j exists iff pole j is active)
                const zc e = zsub(a, cur[k]);
                const double den2 = fma(e.re, e.re, e.im * e.im);
                const bool other = act && (k != i) && den2 > 0.0;
                const double inv2 = other ? fast_rcp(den2) : 0.0;
                rr = zadd(rr, zc{e.re * inv2, -e.im * inv2});
            }
            const zc f = {1.0 - s1.re, -s1.im};
            // correction = f / (f' - f (S0 + R)),  f' = -S2
            const zc g = zmul(f, zadd(s0, rr));
            const zc den = {-s2.re - g.re, -s2.im - g.im};
            const zc corr = zdiv(f, den);
            const bool fin = isfinite(corr.re) && isfinite(corr.im);
            const bool conv = on_pole || (fabs(f.re) + fabs(f.im) <= tol * mag) ||
                              (fin && fabs(corr.re) + fabs(corr.im) <= tol * (fabs(a.re) + fabs(a.im)));
            if (conv) {
                done_mask |= 1u << b;
                nxt[i] = a;
            } else {
                // a non-finite correction (f' - f (S0 + R) = 0 exactly) leaves the root where it is for this round; the
                // other roots move and the configuration changes
                nxt[i] = fin ? zsub(a, corr) : zc{a.re * (1.0 + 1e-8), a.im};
                ++open;
            }
        }
        const int still = __syncthreads_count(open);
        for (int i = tid; i < n; i += nth) cur[i] = nxt[i];
        __syncthreads();
        if (still == 0) break;
    }
    __shared__ int bad_total;
    if (tid == 0) bad_total = 0;
    __syncthreads();
    int bad = 0, b = 0;
    for (int i = tid; i < n; i += nth, ++b) {
        a_out[2 * (m * n + i)] = cur[i].re;
        a_out[2 * (m * n + i) + 1] = cur[i].im;
        bad += (done_mask & (1u << b)) ? 0 : 1;
    }
    if (bad) atomicAdd(&bad_total, bad);
    __syncthreads();
    if (tid == 0) info[m] = bad_total;
}

// w[c, i] = (g_c + a[set[c], i]) / (g_c - a[set[c], i]),  g_c = 2 / exp(log_step[c])
__global__ __launch_bounds__(256) void s4_cayley_map_kernel(const double* __restrict__ a, const int32_t* __restrict__ set,
                                                           const double* __restrict__ log_step, int64_t ncol, int n,
                                                           double* __restrict__ w) {
    const int64_t total = ncol * n;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = idx / n;
        const int i = (int)(idx - c * n);
        const int64_t s = set ? set[c] : c;
        const double g = 2.0 / exp(log_step[c]);
        const zc av = {a[2 * (s * n + i)], a[2 * (s * n + i) + 1]};
        const zc num = {g + av.re, av.im}, den = {g - av.re, -av.im};
        const zc v = zdiv(num, den);
        w[2 * idx] = v.re;
        w[2 * idx + 1] = v.im;
    }
}

int dplr_secular_launch(const double* lam_re, const double* lam_im, const double* p_re, const double* p_im,
                        const double* q_re, const double* q_im, int64_t nmat, int n, int clip, double* a_out,
                        int32_t* info, hipStream_t stream) {
    if (nmat == 0 || n == 0) return 0;
    const size_t lds = (size_t)8 * n * sizeof(double) + (size_t)n * sizeof(int);
    if (lds > 64 * 1024)
        EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dplr_secular_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int threads = ((n + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    hipLaunchKernelGGL(dplr_secular_kernel, dim3((unsigned)nmat), dim3(threads), lds, stream, lam_re, lam_im, p_re, p_im,
                       q_re, q_im, n, clip, a_out, info);
    EGS_LAUNCH_CHECK();
    return 0;
}

int s4_cayley_map_launch(const double* a, const int32_t* set, const double* log_step, int64_t ncol, int n, double* w,
                         hipStream_t stream) {
    if (ncol == 0 || n == 0) return 0;
    const int64_t total = ncol * n;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(s4_cayley_map_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, set, log_step, ncol, n, w);
    EGS_LAUNCH_CHECK();
    return 0;
}

}  // namespace egs
