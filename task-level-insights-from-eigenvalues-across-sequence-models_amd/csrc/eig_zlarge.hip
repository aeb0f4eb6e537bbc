// K5 (complex, n > 64): batched eigenvalues of dense complex128 matrices by LEVEL-3 algorithms, one workgroup of
// 16 wavefronts per matrix.
//
// Replaces np.linalg.eigvals(Ab) on the bilinear-discretised S4 DPLR matrix (analysis/eval_eig.py:296) at the orders
// the reference's configs use above the LDS-resident kernels: state_dim 128 (configs/iclr2026/mqar/mqar-s4.yaml:33),
// 512 (configs/iclr2026/wikitext/wikitext-103-s4.yaml:34), and BASELINE config C5 (n = 256).
//
//   zhess_blocked_kernel : blocked Householder reduction to Hessenberg form (the published xGEHRD / xLAHR2 scheme).
//       Per panel of NB columns: one matrix-vector product with the trailing matrix per column (the only
//       level-2 part: coalesced 16-byte loads, partial sums over column groups), V / Y / T kept in LDS; then the
//       level-3 updates on v_mfma_f64_16x16x4_f64: Y_top = A V T, A <- A - Y V^H, A <- (I - V T^H V^H) A with the
//       trailing tiles held in accumulator registers between the right and the left update (one read + one write of
//       the trailing matrix per panel).  A complex product is four real MFMAs.
//   zmsqr_kernel         : small-bulge multishift QR (Braman / Byers / Mathias, no aggressive early deflation).
//       Chains of 16 single-shift bulges two rows apart are chased through 64 x 64 diagonal windows held in LDS
//       (wave = bulge; lane = column in the row phase, lane = row in the column phase), the reflectors accumulated
//       into U; rows above / columns right of the window are then updated by U as 64 x 64 x 64 complex products on
//       the fp64 matrix cores.  Shifts: eigenvalues of the trailing 8 x 8 block (each used twice), computed by the
//       wave-per-matrix single-shift iteration of zq_common.h, which also finishes deflated blocks of order <= 16.
//
// The working copy is the TRANSPOSE of the caller's row-major matrix (same bytes read as column-major; the spectrum
// of A^T is that of A), which makes every streamed access of the reduction contiguous.
// oracle/zms_proto.py restates both kernels step for step in NumPy (test infrastructure).
#include <stdlib.h>

#include "zq_common.h"

namespace egs {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma64(double a, double b, v4d c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

constexpr int kZL = 1024;      // threads per workgroup (16 wavefronts)
constexpr int kZLW = kZL / 64;

__host__ __device__ inline int zl_np(int n) { return (n + 15) & ~15; }

// LDS-safe accumulation of a double (ds_add_f64)
__device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// ---------------------------------------------------------------------------------------------------
// blocked Hessenberg reduction
// ---------------------------------------------------------------------------------------------------
template <int NB, int TPW>
__global__ __launch_bounds__(kZL) void zhess_blocked_kernel(const double* __restrict__ a, int in_real, int n, int lda,
                                                            int64_t stride, double* __restrict__ w,
                                                            int32_t* __restrict__ info_out, double* __restrict__ work,
                                                            int balance) {
    extern __shared__ __attribute__((aligned(16))) double dynl[];
    __shared__ double red[kZLW];
    __shared__ zq s_alpha;
    __shared__ int s_idx;
    // wave index as a scalar; thread index re-derived per panel from an opaque copy (see zmsqr_kernel): the per-lane
    // addresses of the trailing update are then computed where they are used instead of being held -- spilled -- across
    // the whole reduction
    const int tid0 = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    int tid = tid0, lane = tid0 & 63;
    const int64_t mat = blockIdx.x;
    const int np = zl_np(n);
    const int ldv = np + 1;
    zq* Vs = reinterpret_cast<zq*>(dynl);        // [NB][ldv]   V(r, jj) at Vs[jj * ldv + r]
    zq* Ys = Vs + NB * ldv;                      // [NB][ldv]
    zq* Tsh = Ys + NB * ldv;                     // [NB][NB]    upper triangular
    zq* scr = Tsh + NB * NB;                     // [kZL]       b vector / matvec partials / W and X of the left update
    zq* wsh = scr + kZL;                         // [NB]
    zq* w2sh = wsh + NB;                         // [NB]
    zq* zsh = w2sh + NB;                         // [NB]
    zq* M = reinterpret_cast<zq*>(work) + mat * (int64_t)n * n;
#define MM(i, j) M[(int64_t)(j) * n + (i)]

    // ---- load (transposed view: M(i,j) = A(j,i)) ----
    if (tid == 0) s_idx = 0;
    __syncthreads();
    {
        const double* src = a + mat * stride * (in_real ? 1 : 2);
        bool bad = false;
        double amax = 0.0;
        for (int idx = tid; idx < n * n; idx += kZL) {
            const int j = idx / n, i = idx - j * n;          // M(i,j) = A(j,i) = src[j*lda + i]
            zq v;
            if (in_real) v = {src[(int64_t)j * lda + i], 0.0};
            else v = {src[2 * ((int64_t)j * lda + i)], src[2 * ((int64_t)j * lda + i) + 1]};
            M[idx] = v;
            if (!(fabs(v.re) <= 1.7976931348623157e308) || !(fabs(v.im) <= 1.7976931348623157e308)) bad = true;
            amax = fmax(amax, fmax(fabs(v.re), fabs(v.im)));
        }
        if (bad) s_idx = 1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
        if (lane == 0) red[wv] = amax;
    }
    __syncthreads();
    if (s_idx != 0) {                    // NaN / Inf input: an error like numpy.linalg.eigvals, not a spectrum
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        double* wbad = w + mat * 2 * (int64_t)n;
        for (int j = tid; j < 2 * n; j += kZL) wbad[j] = qnan;
        if (tid == 0) info_out[mat] = n;
        return;
    }
    if (tid == 0) info_out[mat] = 0;
    // xGEEV-style scaling by a power of two (exact); the QR kernel
    // multiplies the eigenvalues back (exponent in the first eigenvalue slot, read before anything is written there).
    // Without it squared moduli underflow / overflow: entries of 1e-160 made every matrix fail to converge.
    {
        double amax = 0.0;
        for (int x = 0; x < kZLW; ++x) amax = fmax(amax, red[x]);
        int e = 0;
        // always normalised (max |entry| in [1, 2), exact): the QR stage then calls a sub-diagonal entry below 1e-200
        // negligible absolutely instead of chasing blocks down to the underflow threshold
        if (amax > 0.0) e = ilogb(amax);
        if (e != 0) {
            for (int idx = tid; idx < n * n; idx += kZL) {
                const zq v = M[idx];
                M[idx] = {scalbn(v.re, -e), scalbn(v.im, -e)};   // (2^-e itself overflows for denormal input)
            }
        }
        if (tid == 0) w[mat * 2 * (int64_t)n] = (double)e;
        __syncthreads();
    }

    // ---- xGEBAL: permutation to isolate eigenvalues + power-of-two equilibration (what xGEEV does first) ----
    if (balance) {
        double* cn = reinterpret_cast<double*>(Ys);
        double* rn = cn + n;
        double* fs = rn + n;
        int* cnt = reinterpret_cast<int*>(fs + n);
#define ZNZ(i, j) (MM(i, j).re != 0.0 || MM(i, j).im != 0.0)
#define ZSWAP(x, y)        \
    {                      \
        const zq t_ = (x); \
        (x) = (y);         \
        (y) = t_;          \
    }
        int k0 = 0, l0 = n - 1;
        for (int i = tid; i < n; i += kZL) {                          // off-diagonal nonzeros of row i
            int c = 0;
            for (int j = 0; j < n; ++j) c += (j != i && ZNZ(i, j)) ? 1 : 0;
            cnt[i] = c;
        }
        __syncthreads();
        while (l0 > 0) {
            if (tid == 0) s_idx = -1;
            __syncthreads();
            for (int i = tid; i <= l0; i += kZL)
                if (cnt[i] == 0) atomicMax(&s_idx, i);
            __syncthreads();
            const int r = s_idx;
            __syncthreads();
            if (r < 0) break;
            if (r != l0) {
                for (int t = tid; t <= l0; t += kZL) ZSWAP(MM(t, r), MM(t, l0));
                __syncthreads();
                for (int t = tid; t < n; t += kZL) ZSWAP(MM(r, t), MM(l0, t));
                if (tid == 0) {
                    const int c = cnt[r];
                    cnt[r] = cnt[l0];
                    cnt[l0] = c;
                }
                __syncthreads();
            }
            for (int t = tid; t < l0; t += kZL)                      // column l0 leaves the window
                if (ZNZ(t, l0)) cnt[t] -= 1;
            --l0;
            __syncthreads();
        }
        for (int j = wv; j <= l0; j += kZLW) {                        // off-diagonal nonzeros of column j, rows 0..l0
            double c = 0.0;
            for (int i = lane; i <= l0; i += 64) c += (i != j && ZNZ(i, j)) ? 1.0 : 0.0;
            c = wave_sum(c);
            if (lane == 0) cnt[j] = (int)c;
        }
        __syncthreads();
        while (k0 < l0) {
            if (tid == 0) s_idx = n;
            __syncthreads();
            for (int j = k0 + tid; j <= l0; j += kZL)
                if (cnt[j] == 0) atomicMin(&s_idx, j);
            __syncthreads();
            const int c = s_idx;
            __syncthreads();
            if (c >= n) break;
            if (c != k0) {
                for (int t = tid; t <= l0; t += kZL) ZSWAP(MM(t, c), MM(t, k0));
                __syncthreads();
                for (int t = k0 + tid; t < n; t += kZL) ZSWAP(MM(c, t), MM(k0, t));
                if (tid == 0) {
                    const int q = cnt[c];
                    cnt[c] = cnt[k0];
                    cnt[k0] = q;
                }
                __syncthreads();
            }
            for (int t = k0 + 1 + tid; t <= l0; t += kZL)            // row k0 leaves the window
                if (ZNZ(k0, t)) cnt[t] -= 1;
            ++k0;
            __syncthreads();
        }
        for (int pass = 0; pass < 16; ++pass) {
            for (int j = k0 + wv; j <= l0; j += kZLW) {               // column norms: wave per column
                double c = 0.0;
                for (int i = k0 + lane; i <= l0; i += 64)
                    if (i != j) c += qabs1(MM(i, j));
                c = wave_sum(c);
                if (lane == 0) cn[j] = c;
            }
            for (int i = k0 + tid; i <= l0; i += kZL) {               // row norms: thread per row
                double r = 0.0;
                for (int j = k0; j <= l0; ++j)
                    if (j != i) r += qabs1(MM(i, j));
                rn[i] = r;
            }
            if (tid == 0) s_idx = 0;
            __syncthreads();
            for (int i = tid; i < n; i += kZL) {
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
            for (int idx = tid; idx < n * n; idx += kZL) {
                const int j = idx / n, i = idx - j * n;
                M[idx] = qscale(M[idx], fs[j] * cn[i]);
            }
            __syncthreads();
        }
        __syncthreads();
#undef ZNZ
#undef ZSWAP
    }

    // ---- panels ----
    const int rowtiles = np / 16;
    for (int k = 0; k < n - 2; k += NB) {
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        const int jb = (NB < n - 2 - k) ? NB : (n - 2 - k);
        for (int idx = tid; idx < 2 * NB * ldv + NB * NB; idx += kZL) Vs[idx] = {0.0, 0.0};   // V, Y, T
        __syncthreads();
        const int r = tid;                                            // this thread's row in the vector steps
        const bool rin = r >= k + 1 && r < n;
        for (int j = 0; j < jb; ++j) {
            const int c = k + j;
            // (a) column c with the right update of the earlier reflectors of this panel
            if (rin) {
                zq b = MM(r, c);
                for (int jj = 0; jj < j; ++jj) b = qsub(b, qmulc(Ys[jj * ldv + r], Vs[jj * ldv + c]));
                scr[r] = b;
            }
            __syncthreads();
            if (j > 0) {
                // (b1) w = V^H b, one wave per earlier reflector
                if (wv < j) {
                    zq acc = {0.0, 0.0};
                    for (int rr = k + 1 + lane; rr < n; rr += 64) acc = qadd(acc, qcmul(Vs[wv * ldv + rr], scr[rr]));
                    acc.re = wave_sum(acc.re);
                    acc.im = wave_sum(acc.im);
                    if (lane == 0) wsh[wv] = acc;
                }
                __syncthreads();
                // (b2) w2 = T^H w
                if (tid < j) {
                    zq acc = {0.0, 0.0};
                    for (int bb = 0; bb <= tid; ++bb) acc = qadd(acc, qcmul(Tsh[bb * NB + tid], wsh[bb]));
                    w2sh[tid] = acc;
                }
                __syncthreads();
            }
            // (b3) b -= V w2, norm of the part below the subdiagonal
            zq bv = {0.0, 0.0};
            double part = 0.0;
            if (rin) {
                bv = scr[r];
                for (int jj = 0; jj < j; ++jj) bv = qsub(bv, qmul(Vs[jj * ldv + r], w2sh[jj]));
                if (r >= c + 2) part = fma(bv.re, bv.re, bv.im * bv.im);
                if (r == c + 1) s_alpha = bv;
            }
            part = wave_sum(part);
            if (lane == 0) red[wv] = part;
            __syncthreads();
            double xnorm2 = 0.0;
#pragma unroll
            for (int x = 0; x < kZLW; ++x) xnorm2 += red[x];
            const zq alpha = s_alpha;
            // (c) reflector (block-uniform scalars)
            // A column whose squared norm is below 2^-800 is left alone (tau = 0; its sub-sub-diagonal part, < 4e-121 of
            // a matrix scaled to max |entry| in [1, 2), is dropped): further down the squares underflow, the norm comes
            // out as zero and tau = (beta - alpha) / beta as NaN -- the trailing columns of c * ones(128, 128) did.
            const double tot = fma(alpha.re, alpha.re, alpha.im * alpha.im) + xnorm2;
            const bool act = !(xnorm2 == 0.0 && alpha.im == 0.0) && tot >= 1.4996968138956309e-241;
            const double nrm = sqrt(tot);
            const double beta = act ? -copysign(nrm, alpha.re) : alpha.re;
            const zq tau = qsel(act, zq{(beta - alpha.re) / beta, -alpha.im / beta}, zq{0.0, 0.0});
            const zq sc = qsel(act, qdiv({1.0, 0.0}, {alpha.re - beta, alpha.im}), zq{0.0, 0.0});
            if (r < np) {
                zq v = {0.0, 0.0};
                if (r == c + 1) v = {1.0, 0.0};
                else if (r >= c + 2 && r < n) v = qmul(bv, sc);
                Vs[j * ldv + r] = v;
                if (rin) {
                    if (r <= c) MM(r, c) = bv;
                    else if (r == c + 1) MM(r, c) = {beta, 0.0};
                    else MM(r, c) = {0.0, 0.0};
                }
            }
            __syncthreads();
            // (d) z = V^H v for the earlier reflectors; y = A(k+1:n, c+1:n) v
            if (wv < j) {
                zq acc = {0.0, 0.0};
                for (int rr = c + 1 + lane; rr < n; rr += 64) acc = qadd(acc, qcmul(Vs[wv * ldv + rr], Vs[j * ldv + rr]));
                acc.re = wave_sum(acc.re);
                acc.im = wave_sum(acc.im);
                if (lane == 0) zsh[wv] = acc;
            }
            {
                const int nr = n - k - 1;                             // rows k+1 .. n-1
                const int rp = (nr + 63) & ~63;
                const int ng = kZL / rp;                              // column groups (>= 1)
                const int g = tid / rp, rr = tid - g * rp;
                zq acc = {0.0, 0.0};
                if (g < ng) {
                    const int row = (rr < nr) ? k + 1 + rr : n - 1;   // clamped: always a valid address
                    const zq* vj = Vs + j * ldv;
                    int cc = c + 1 + g;
                    for (; cc + 3 * ng < n; cc += 4 * ng) {
                        const zq m0 = MM(row, cc), m1 = MM(row, cc + ng), m2 = MM(row, cc + 2 * ng), m3 = MM(row, cc + 3 * ng);
                        acc = qadd(acc, qmul(m0, vj[cc]));
                        acc = qadd(acc, qmul(m1, vj[cc + ng]));
                        acc = qadd(acc, qmul(m2, vj[cc + 2 * ng]));
                        acc = qadd(acc, qmul(m3, vj[cc + 3 * ng]));
                    }
                    for (; cc < n; cc += ng) acc = qadd(acc, qmul(MM(row, cc), vj[cc]));
                }
                scr[tid] = acc;
                __syncthreads();
                // (e) Y(:, j) = tau (y - Y z);  T(:, j) = -tau T z, T(j,j) = tau
                if (rin) {
                    const int r0 = r - k - 1;
                    zq y = scr[r0];
                    for (int x = 1; x < ng; ++x) y = qadd(y, scr[x * rp + r0]);
                    for (int jj = 0; jj < j; ++jj) y = qsub(y, qmul(Ys[jj * ldv + r], zsh[jj]));
                    Ys[j * ldv + r] = qmul(tau, y);
                }
                if (tid < j) {
                    zq acc2 = {0.0, 0.0};
                    for (int bb = tid; bb < j; ++bb) acc2 = qadd(acc2, qmul(Tsh[tid * NB + bb], zsh[bb]));
                    Tsh[tid * NB + j] = qmul({-tau.re, -tau.im}, acc2);
                }
                if (tid == j) Tsh[j * NB + j] = tau;
            }
            __syncthreads();
        }

        // ---- (f) top rows of Y:  Y(0:k+1, :) = A(0:k+1, k+1:n) V T   (matrix cores) ----
        {
            const int ntile = (k + 1 + 15) / 16;
            const int jjl = lane & 15, kk = lane >> 4;
            for (int rt = wv; rt < ntile; rt += kZLW) {
                v4d yr = {0.0, 0.0, 0.0, 0.0}, yi = {0.0, 0.0, 0.0, 0.0};
                const int row = rt * 16 + (lane & 15);
                const int rowc = (row < n) ? row : n - 1;
                for (int cc0 = k + 1; cc0 < n; cc0 += 4) {
                    const int cc = cc0 + kk;
                    const int ccc = (cc < n) ? cc : n - 1;
                    zq av = MM(rowc, ccc);
                    zq bvv = qsel(jjl < NB, Vs[(jjl < NB ? jjl : 0) * ldv + ccc], zq{0.0, 0.0});
                    if (cc >= n) bvv = {0.0, 0.0};
                    yr = mfma64(av.re, bvv.re, yr);
                    yr = mfma64(-av.im, bvv.im, yr);
                    yi = mfma64(av.re, bvv.im, yi);
                    yi = mfma64(av.im, bvv.re, yi);
                }
                if (jjl < NB) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ro = rt * 16 + kk + 4 * q;
                        if (ro <= k) Ys[jjl * ldv + ro] = {yr[q], yi[q]};
                    }
                }
            }
        }
        __syncthreads();
        if (r <= k) {
            // Y_top <- Y_top T (in place, last column first), then the panel columns' rows 0..k
            for (int aa = jb - 1; aa >= 0; --aa) {
                zq acc = {0.0, 0.0};
                for (int bb = 0; bb <= aa; ++bb) acc = qadd(acc, qmul(Ys[bb * ldv + r], Tsh[bb * NB + aa]));
                Ys[aa * ldv + r] = acc;
            }
            for (int j = 1; j < jb; ++j) {
                const int c = k + j;
                zq b = MM(r, c);
                for (int jj = 0; jj < j; ++jj) b = qsub(b, qmulc(Ys[jj * ldv + r], Vs[jj * ldv + c]));
                MM(r, c) = b;
            }
        }
        __syncthreads();

        // ---- (g, h) trailing columns: A <- A - Y V^H, then A <- A - V T^H (V^H A), tiles in registers ----
        {
            const int c0 = k + jb;
            const int ncb = (n - c0 + 15) / 16;
            const int gw = (rowtiles + TPW - 1) / TPW;                // waves per column block
            int cbn = kZLW / gw;                                      // concurrent column blocks
            if (cbn > 4) cbn = 4;
            const int grp = wv / gw, wi = wv - grp * gw;
            for (int cb0 = 0; cb0 < ncb; cb0 += cbn) {
                int lane_c = lane;                                   // per column block, see the top of the kernel
                asm volatile("" : "+v"(lane_c));
                const int cl = lane_c & 15, kk = lane_c >> 4;
                const bool wact = grp < cbn && cb0 + grp < ncb;
                const int cbase = c0 + (cb0 + grp) * 16;
                const int col = cbase + cl;
                const bool colok = wact && col < n;
                const int colc = colok ? col : n - 1;
                const int colv = (wact && col < np) ? col : np;       // Vs pad entry (zero) for columns beyond n
                for (int idx = tid; idx < kZL; idx += kZL) scr[idx] = {0.0, 0.0};
                __syncthreads();
                v4d tr[TPW], ti[TPW];
                if (wact) {
                    v4d wr = {0.0, 0.0, 0.0, 0.0}, wim = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < TPW; ++q) {
                        const int rt = wi * TPW + q;
                        tr[q] = {0.0, 0.0, 0.0, 0.0};
                        ti[q] = {0.0, 0.0, 0.0, 0.0};
                        if (rt < rowtiles) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int row = rt * 16 + kk + 4 * g;
                                const int rowc = (row < n) ? row : n - 1;
                                const zq t = MM(rowc, colc);
                                const bool ok = colok && row < n;
                                tr[q][g] = ok ? t.re : 0.0;
                                ti[q][g] = ok ? t.im : 0.0;
                            }
                            // right update: tile -= Y(rows, :) conj(V(cols, :))^T
#pragma unroll
                            for (int t = 0; t < NB / 4; ++t) {
                                const int jj = 4 * t + kk;
                                const zq y = Ys[jj * ldv + rt * 16 + cl];
                                const zq v = Vs[jj * ldv + colv];
                                tr[q] = mfma64(-y.re, v.re, tr[q]);
                                tr[q] = mfma64(-y.im, v.im, tr[q]);
                                ti[q] = mfma64(-y.im, v.re, ti[q]);
                                ti[q] = mfma64(y.re, v.im, ti[q]);
                            }
                            // left update, first half: W += V(rows, :)^H tile   (the tile registers are the B operand)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const zq v = qsel(cl < NB, Vs[(cl < NB ? cl : 0) * ldv + rt * 16 + 4 * t + kk], zq{0.0, 0.0});
                                wr = mfma64(v.re, tr[q][t], wr);
                                wr = mfma64(v.im, ti[q][t], wr);
                                wim = mfma64(v.re, ti[q][t], wim);
                                wim = mfma64(-v.im, tr[q][t], wim);
                            }
                        }
                    }
                    // W(jj, col) of this column block: sum over the gw waves of the group
                    double* wsum = reinterpret_cast<double*>(scr + grp * 256);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int jj = kk + 4 * g;
                        lds_add(wsum + 2 * (jj * 16 + cl), wr[g]);
                        lds_add(wsum + 2 * (jj * 16 + cl) + 1, wim[g]);
                    }
                }
                __syncthreads();
                // X = T^H W   (thread per element, then in place)
                zq xv = {0.0, 0.0};
                {
                    const int g2 = tid >> 8, aa = (tid >> 4) & 15, cc = tid & 15;
                    if (g2 < cbn && aa < NB) {
                        for (int bb = 0; bb <= aa; ++bb) xv = qadd(xv, qcmul(Tsh[bb * NB + aa], scr[g2 * 256 + bb * 16 + cc]));
                    }
                }
                __syncthreads();
                scr[tid] = xv;
                __syncthreads();
                if (wact) {
#pragma unroll
                    for (int q = 0; q < TPW; ++q) {
                        const int rt = wi * TPW + q;
                        if (rt < rowtiles) {
#pragma unroll
                            for (int t = 0; t < NB / 4; ++t) {
                                const int aa = 4 * t + kk;
                                const zq v = Vs[aa * ldv + rt * 16 + cl];
                                const zq x = scr[grp * 256 + aa * 16 + cl];
                                tr[q] = mfma64(-v.re, x.re, tr[q]);
                                tr[q] = mfma64(v.im, x.im, tr[q]);
                                ti[q] = mfma64(-v.re, x.im, ti[q]);
                                ti[q] = mfma64(-v.im, x.re, ti[q]);
                            }
                            if (colok) {
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const int row = rt * 16 + kk + 4 * g;
                                    if (row < n) MM(row, col) = {tr[q][g], ti[q][g]};
                                }
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
#undef MM
}

// ---------------------------------------------------------------------------------------------------
// multishift QR
// ---------------------------------------------------------------------------------------------------
constexpr int QW = 64;                  // window order
constexpr int QLD = 65;                 // leading dimension of the LDS windows (odd: rows and columns conflict-free)
constexpr int QM = 16;                  // bulges per chain (= wavefronts)
#ifndef EGS_ZQNS
#define EGS_ZQNS 8
#endif
constexpr int QNS = EGS_ZQNS;           // distinct shifts per sweep (QM / QNS bulges share one)
constexpr int QNBS = QW - 2 * QM - 1;   // time steps per window
constexpr int QSMALL = 16;              // deflated blocks up to this order are finished by one wavefront
constexpr int QSLD = 17;

// All bulges advance through time steps t0..t1 inside the window (Wn = H[s : s+wd, s : s+wd]).  Wave = bulge.
template <bool WITHU>
__device__ __forceinline__ void zchase(zq* __restrict__ Wn, zq* __restrict__ Un, const zq* __restrict__ shifts, int s,
                                       int wd, int l, int i, int t0, int t1, int lane, int wv) {
    for (int t = t0; t <= t1; ++t) {
        const int tb = t - 2 * wv;
        const int p = l + tb;
        const bool act = tb >= 0 && p <= i - 1;                        // wave-uniform
        const int pw = p - s;
        zq v2 = {0.0, 0.0}, tau = {0.0, 0.0};
        if (act) {
            zq a0, a1;
            if (p == l) {
                a0 = qsub(Wn[pw * QLD + pw], shifts[wv]);
                a1 = Wn[(pw + 1) * QLD + pw];
            } else {
                a0 = Wn[pw * QLD + pw - 1];
                a1 = Wn[(pw + 1) * QLD + pw - 1];
            }
            double beta;
            zlarfg2_fast(a0, a1, beta, v2, tau);
            // rows p, p+1 (lane = column): c <- H^H c
            const zq r0 = Wn[pw * QLD + lane], r1 = Wn[(pw + 1) * QLD + lane];
            const zq rs = qcmul(tau, qadd(r0, qcmul(v2, r1)));
            if (lane >= pw && lane < wd) {
                Wn[pw * QLD + lane] = qsub(r0, rs);
                Wn[(pw + 1) * QLD + lane] = qsub(r1, qmul(rs, v2));
            }
            if (p > l && lane == pw - 1) {
                Wn[pw * QLD + lane] = {beta, 0.0};
                Wn[(pw + 1) * QLD + lane] = {0.0, 0.0};
            }
        }
        __syncthreads();
        if (act) {
            // columns p, p+1 (lane = row): r <- r H
            const int rmaxw = ((p + 2 < i) ? p + 2 : i) - s;
            const zq c0 = Wn[lane * QLD + pw], c1 = Wn[lane * QLD + pw + 1];
            const zq cs = qmul(tau, qadd(c0, qmul(c1, v2)));
            if (lane <= rmaxw) {
                Wn[lane * QLD + pw] = qsub(c0, cs);
                Wn[lane * QLD + pw + 1] = qsub(c1, qmulc(cs, v2));
            }
            if (WITHU) {
                const zq u0 = Un[lane * QLD + pw], u1 = Un[lane * QLD + pw + 1];
                const zq us = qmul(tau, qadd(u0, qmul(u1, v2)));
                if (lane < wd) {
                    Un[lane * QLD + pw] = qsub(u0, us);
                    Un[lane * QLD + pw + 1] = qsub(u1, qmulc(us, v2));
                }
            }
        }
        __syncthreads();
    }
}

// Complex Givens rotation with a REAL cosine:  Q = [[c, s], [-conj(s), c]],  Q (a0; a1) = (rho; 0).
// Applying it to a pair of complex numbers costs 12 fused multiply-adds (the 2-element Householder form: 18).
// A pair whose squared moduli would underflow (|a|^2 < 2^-800) is rescaled by a power of two first: c and s are invariant,
// rho scales back.  Without it such a pair got the identity and was never annihilated -- a constant matrix, whose
// Hessenberg form cascades down to blocks of 1e-190, did not converge (round 3, tools/stress_paths.py).
__device__ __forceinline__ void zgivens(zq a0, zq a1, double& c, zq& sn, zq& rho) {
    const double tiny = 1e-300;
    double n0 = fma(a0.re, a0.re, a0.im * a0.im);
    double n1 = fma(a1.re, a1.re, a1.im * a1.im);
    double nn = n0 + n1;
    int back = 0;
    if (nn < 1.4996968138956309e-241) {                                // rare, per lane
        const double big = fmax(fmax(fabs(a0.re), fabs(a0.im)), fmax(fabs(a1.re), fabs(a1.im)));
        if (big > 0.0) {
            back = ilogb(big);
            a0 = {scalbn(a0.re, -back), scalbn(a0.im, -back)};
            a1 = {scalbn(a1.re, -back), scalbn(a1.im, -back)};
            n0 = fma(a0.re, a0.re, a0.im * a0.im);
            n1 = fma(a1.re, a1.re, a1.im * a1.im);
            nn = n0 + n1;
        }
    }
    const double rn = fast_rsqrt(fmax(nn, tiny));                       // 1 / norm
    const double r0 = fast_rsqrt(fmax(n0, tiny));                       // 1 / |a0|
    const bool z0 = n0 < tiny;
    const bool zz = nn < tiny;                                          // nothing to rotate: identity
    const zq sg = qsel(z0, zq{1.0, 0.0}, qscale(a0, r0));                   // a0 / |a0|
    c = zz ? 1.0 : (z0 ? 0.0 : n0 * r0 * rn);
    sn = qsel(zz, zq{0.0, 0.0}, qscale(qmulc(sg, a1), rn));                 // sg conj(a1) / norm
    rho = zz ? a0 : qscale(sg, nn * rn);
    if (back != 0) rho = {scalbn(rho.re, back), scalbn(rho.im, back)};
}
// rows:    (x0; x1) <- Q (x0; x1)
__device__ __forceinline__ void zrot_rows(double c, zq sn, zq x0, zq x1, zq& y0, zq& y1) {
    y0 = {fma(c, x0.re, fma(sn.re, x1.re, -sn.im * x1.im)), fma(c, x0.im, fma(sn.re, x1.im, sn.im * x1.re))};
    y1 = {fma(c, x1.re, -fma(sn.re, x0.re, sn.im * x0.im)), fma(c, x1.im, -fma(sn.re, x0.im, -sn.im * x0.re))};
}
// columns: (x0, x1) <- (x0, x1) Q^H
__device__ __forceinline__ void zrot_cols(double c, zq sn, zq x0, zq x1, zq& y0, zq& y1) {
    y0 = {fma(c, x0.re, fma(sn.re, x1.re, sn.im * x1.im)), fma(c, x0.im, fma(sn.re, x1.im, -sn.im * x1.re))};
    y1 = {fma(c, x1.re, -fma(sn.re, x0.re, -sn.im * x0.im)), fma(c, x1.im, -fma(sn.re, x0.im, sn.im * x0.re))};
}

// The chase as it runs now (the first version above computed every 2-element Householder reflector in all 64 lanes
// of its wave -- 28 % of the issued fp64 instructions -- on the critical path of every step):
//   * rotations instead of reflectors (12 instead of 18 multiply-adds per element pair);
//   * wave = bulge for the bulk work: row operation on the window and column operation on U in phase 1, column
//     operation on the window in phase 2;
//   * the rotation of step t+1 is computed DURING phase 2 of step t by one lane per bulge (lanes 0..3 of waves 0..3)
//     from three numbers the row phase leaves in a mailbox: the two entries of row p+1 it has just produced and the
//     subdiagonal entry below them.  The two window entries that become (rho, 0) are written by that lane and skipped
//     by the bulk column operation, so the scalar chain and the bulk update never touch the same element.
// prm: [2][QM][3] = ({c, -}, s, rho) per bulge, double-buffered by step parity;  mbx: [2][QM][3].
// rho of step t is put into the window by the bulge's own wave at the start of phase 1 (the wave of the bulge behind
// takes it from the table), so no window element is written by one wave and read by another inside a phase.
template <bool WITHU>
__device__ __forceinline__ void zchase3(zq* __restrict__ Wn, zq* __restrict__ Un, zq* __restrict__ prm,
                                        zq* __restrict__ mbx, const zq* __restrict__ shifts, int s, int wd, int l, int i,
                                        int t0, int t1, int lane, int wv, const bool prof, long long (&cst)[4]) {
    long long tl = prof ? (long long)__builtin_readcyclecounter() : 0;
#define CSTAMP(k)                                                  \
    if (prof) {                                                    \
        const long long now_ = __builtin_readcyclecounter();       \
        cst[k] += now_ - tl;                                       \
        tl = now_;                                                 \
    }
    const int bg = 4 * (wv & 3) + (lane & 3);                          // look-ahead lane -> bulge
    const bool lwave = wv < 4;
    if (lwave) {                                                       // prologue: rotations of step t0 from the window
        const int tb = t0 - 2 * bg;
        const int p = l + tb;
        const bool act = tb >= 0 && p <= i - 1;
        const int pw = act ? p - s : 1;
        const bool intro = act && p == l;
        const int pc = (pw >= 1) ? pw : 1;
        const zq a0 = qsel(intro, qsub(Wn[pw * QLD + pw], shifts[bg]), Wn[pc * QLD + pc - 1]);
        const zq a1 = qsel(intro, Wn[(pw + 1) * QLD + pw], Wn[(pc + 1) * QLD + pc - 1]);
        double c;
        zq sn, rho;
        zgivens(a0, a1, c, sn, rho);
        if (act && lane < 4) {
            prm[((t0 & 1) * QM + bg) * 3 + 0] = {c, 0.0};
            prm[((t0 & 1) * QM + bg) * 3 + 1] = sn;
            prm[((t0 & 1) * QM + bg) * 3 + 2] = rho;
            if (bg + 1 < QM) mbx[((t0 & 1) * QM + bg + 1) * 3 + 2] = rho;
        }
    }
    __syncthreads();
    const int b = wv;
    zq ucar = {0.0, 0.0};
    bool ucv = false;                                                  // ucar holds column p of U
    for (int t = t0; t <= t1; ++t) {
        const int par = t & 1;
        const int tb = t - 2 * b;
        const int p = l + tb;
        const bool act = tb >= 0 && p <= i - 1;                        // wave-uniform
        const int pw = p - s;
        double c = 1.0;
        zq sn = {0.0, 0.0};
        // ---- phase 1: rows p, p+1 of the window (lane = column), columns p, p+1 of U (lane = row), mailbox ----
        if (act) {
            c = prm[(par * QM + b) * 3 + 0].re;
            sn = prm[(par * QM + b) * 3 + 1];
            const zq rho = prm[(par * QM + b) * 3 + 2];
            const zq r0 = Wn[pw * QLD + lane], r1 = Wn[(pw + 1) * QLD + lane];
            zq y0, y1;
            zrot_rows(c, sn, r0, r1, y0, y1);
            // column p-1 holds the bulge this rotation annihilates: lane p-1 stores (rho, 0) with the same two stores
            const bool bl_lane = tb >= 1 && lane == pw - 1;
            if ((lane >= pw && lane < wd) || bl_lane) {
                Wn[pw * QLD + lane] = qsel(bl_lane, rho, y0);
                Wn[(pw + 1) * QLD + lane] = qsel(bl_lane, zq{0.0, 0.0}, y1);
            }
            // mailbox for the look-ahead lane: the new (p+1, p), (p+1, p+1) and, unless the bulge in front supplies its
            // rho, the subdiagonal entry (p+2, p+1)
            const bool ahead = b >= 1 && p + 2 <= i - 1;               // a bulge sits at p + 2 this step
            zq mb = y1;
            int nmb = 2;
            if (!ahead) {
                const int rb = (pw + 2 < wd) ? pw + 2 : wd - 1;
                const zq bl = Wn[rb * QLD + pw + 1];
                mb = qsel(lane == pw + 2, bl, mb);
                nmb = 3;
            }
            if (lane >= pw && lane < pw + nmb) mbx[(par * QM + b) * 3 + (lane - pw)] = mb;
        }
        CSTAMP(0)
        __syncthreads();
        CSTAMP(1)
        // ---- phase 2: columns p, p+1 of the window (lane = row); next rotation by the look-ahead lanes ----
        // the bulk loads are issued first so that, in the four waves that also carry look-ahead lanes, their LDS latency
        // runs under the scalar chain
        const int pwc = act ? pw : 0;
        const zq c0 = Wn[lane * QLD + pwc], c1 = Wn[lane * QLD + pwc + 1];
        zq ul = {0.0, 0.0}, u1 = {0.0, 0.0};
        if (WITHU) {
            ul = Un[lane * QLD + pwc];
            u1 = Un[lane * QLD + pwc + 1];
        }
        if (lwave) {
            __builtin_amdgcn_s_setprio(3);                             // the scalar chain is the critical path of the step
            const int tb1 = t + 1 - 2 * bg;
            const int p1 = l + tb1;
            const bool act1 = t < t1 && tb1 >= 0 && p1 <= i - 1;
            const bool intro = act1 && tb1 == 0;
            const int pwn = act1 ? p1 - s : 1;
            const double lc = prm[(par * QM + bg) * 3 + 0].re;
            const zq ls = prm[(par * QM + bg) * 3 + 1];
            const zq x00 = mbx[(par * QM + bg) * 3 + 0], x01 = mbx[(par * QM + bg) * 3 + 1], x11 = mbx[(par * QM + bg) * 3 + 2];
            const zq i0 = Wn[pwn * QLD + pwn], i1 = Wn[(pwn + 1) * QLD + pwn];
            // column operation on rows p+1 (entries x00, x01) and p+2 (entries 0, x11), first column only
            zq a0 = {fma(lc, x00.re, fma(ls.re, x01.re, ls.im * x01.im)), fma(lc, x00.im, fma(ls.re, x01.im, -ls.im * x01.re))};
            zq a1 = {fma(ls.re, x11.re, ls.im * x11.im), fma(ls.re, x11.im, -ls.im * x11.re)};
            a0 = qsel(intro, qsub(i0, shifts[bg]), a0);
            a1 = qsel(intro, i1, a1);
            double nc;
            zq nsn, nrho;
            zgivens(a0, a1, nc, nsn, nrho);
            if (act1 && lane < 4) {
                prm[((par ^ 1) * QM + bg) * 3 + 0] = {nc, 0.0};
                prm[((par ^ 1) * QM + bg) * 3 + 1] = nsn;
                prm[((par ^ 1) * QM + bg) * 3 + 2] = nrho;
                if (bg + 1 < QM) mbx[((par ^ 1) * QM + bg + 1) * 3 + 2] = nrho;   // "below" of the bulge behind, next step
            }
            __builtin_amdgcn_s_setprio(0);
        }
        if (act) {
            const int rmaxw = ((p + 2 < i) ? p + 2 : i) - s;
            zq y0, y1;
            zrot_cols(c, sn, c0, c1, y0, y1);
            if (lane <= rmaxw) {
                Wn[lane * QLD + pw] = y0;
                Wn[lane * QLD + pw + 1] = y1;
            }
            if (WITHU) {
                // column p of U is still in this wave's registers from the previous step (no other bulge touches it)
                const zq u0 = qsel(ucv, ucar, ul);
                zq v0, v1;
                zrot_cols(c, sn, u0, u1, v0, v1);
                ucv = t < t1 && p + 1 <= i - 1;                        // the bulge moves on inside this window
                ucar = v1;
                if (lane < wd) {
                    Un[lane * QLD + pw] = v0;
                    if (!ucv) Un[lane * QLD + pw + 1] = v1;
                }
            }
        }
        CSTAMP(2)
        __syncthreads();
        CSTAMP(3)
    }
#undef CSTAMP
}

// zchase3 for the INTERIOR stages of a sweep -- every one of the 16 bulges is inside the window for every step of the
// stage, none is being introduced and none leaves the active block:
//     t0 >= 2 (QM - 1) + 1,   l + t1 + 3 <= i,   s = l + t0 - 2 (QM - 1) - 1,   wd = QW,
// so bulge b sits at window row pw = 2 (QM - 1) + 1 - 2 b + (t - t0) and every flag of the general routine (act, intro,
// ahead, bl_lane, the clamps of pw, rmaxw, wd) is a constant.  The general body issues ~400 instructions per wave and
// step, 73 of them fp64 arithmetic; with four waves per SIMD the chase is bound by exactly that count (16 waves x
// instructions per step / 4 SIMDs ~ the 3,250 cycles a step was measured at).  This body keeps the protocol of zchase3
// (prm / mbx tables double-buffered by step parity, look-ahead by lanes 0..3 of waves 0..3) and drops everything else:
// pointers advance by constants, U's shared column stays in a register for the whole stage, and the annihilated bulge
// column goes through the row rotation like every other column (lane pw-1 stores (rho~, eps) instead of a selected
// (rho, 0): rounding-level entries below the sub-diagonal, which the next rotation of that column absorbs).
#ifdef EGS_ZCHASE_STAMPS
constexpr bool kZSteadyStamps = true;
#else
constexpr bool kZSteadyStamps = false;
#endif
template <bool PROF>
__device__ __forceinline__ void zchase3_steady(zq* __restrict__ Wn, zq* __restrict__ Un, zq* __restrict__ prm,
                                               zq* __restrict__ mbx, int t0, int t1, int lane, int wv,
                                               long long (&cst)[4]) {
    long long tl = PROF ? (long long)__builtin_readcyclecounter() : 0;
#define SSTAMP(k)                                                  \
    if (PROF) {                                                    \
        const long long now_ = __builtin_readcyclecounter();       \
        cst[k] += now_ - tl;                                       \
        tl = now_;                                                 \
    }
    constexpr int P0 = 2 * (QM - 1) + 1;                               // window row of bulge 0 at step t0
    const int bg = 4 * (wv & 3) + (lane & 3);                          // look-ahead lane -> bulge
    const bool lwave = wv < 4;
    if (lwave) {                                                       // rotations of step t0 from the window
        const int pc = P0 - 2 * bg;
        double c;
        zq sn, rho;
        zgivens(Wn[pc * QLD + pc - 1], Wn[(pc + 1) * QLD + pc - 1], c, sn, rho);
        if (lane < 4) {
            prm[((t0 & 1) * QM + bg) * 3 + 0] = {c, 0.0};
            prm[((t0 & 1) * QM + bg) * 3 + 1] = sn;
            prm[((t0 & 1) * QM + bg) * 3 + 2] = rho;
            if (bg + 1 < QM) mbx[((t0 & 1) * QM + bg + 1) * 3 + 2] = rho;
        }
    }
    __syncthreads();
    const int b = wv;
    int pw = P0 - 2 * b;
    zq* rowp = Wn + pw * QLD + lane;                                   // (pw, lane), advances by QLD
    zq* colp = Wn + lane * QLD + pw;                                   // (lane, pw), advances by 1
    zq* ucol = Un + lane * QLD + pw;
    zq ucar = ucol[0];                                                 // column pw of U: carried from step to step
    for (int t = t0; t <= t1; ++t) {
        const int par = t & 1;
        const zq* pr = prm + (par * QM + b) * 3;
        // ---- phase 1: rows pw, pw+1 of the window (lane = column), mailbox ----
        const double c = pr[0].re;
        const zq sn = pr[1];
        const zq r0 = rowp[0], r1 = rowp[QLD];
        zq y0, y1;
        zrot_rows(c, sn, r0, r1, y0, y1);
        if (lane >= pw - 1) {
            rowp[0] = y0;
            rowp[QLD] = y1;
        }
        {
            zq* mb = mbx + (par * QM + b) * 3;
            if (b == 0) {                                              // nobody in front: the sub-diagonal entry itself
                const zq bl = Wn[(pw + 2) * QLD + pw + 1];
                const zq v = qsel(lane == pw + 2, bl, y1);
                if (lane >= pw && lane < pw + 3) mb[lane - pw] = v;
            } else {
                if (lane >= pw && lane < pw + 2) mb[lane - pw] = y1;
            }
        }
        SSTAMP(0)
        __syncthreads();
        SSTAMP(1)
        // ---- phase 2: columns pw, pw+1 of the window and of U (lane = row); next rotation by the look-ahead lanes ----
        const zq c0 = colp[0], c1 = colp[1];
        const zq u1 = ucol[1];
        if (lwave) {
            __builtin_amdgcn_s_setprio(3);
            const zq* lp = prm + (par * QM + bg) * 3;
            const zq* lm = mbx + (par * QM + bg) * 3;
            const double lc = lp[0].re;
            const zq ls = lp[1];
            const zq x00 = lm[0], x01 = lm[1], x11 = lm[2];
            const zq a0 = {fma(lc, x00.re, fma(ls.re, x01.re, ls.im * x01.im)), fma(lc, x00.im, fma(ls.re, x01.im, -ls.im * x01.re))};
            const zq a1 = {fma(ls.re, x11.re, ls.im * x11.im), fma(ls.re, x11.im, -ls.im * x11.re)};
            double nc;
            zq nsn, nrho;
            zgivens(a0, a1, nc, nsn, nrho);
            if (t < t1 && lane < 4) {
                zq* np = prm + ((par ^ 1) * QM + bg) * 3;
                np[0] = {nc, 0.0};
                np[1] = nsn;
                np[2] = nrho;
                if (bg + 1 < QM) mbx[((par ^ 1) * QM + bg + 1) * 3 + 2] = nrho;
            }
            __builtin_amdgcn_s_setprio(0);
        }
        {
            zq y0c, y1c;
            zrot_cols(c, sn, c0, c1, y0c, y1c);
            if (lane <= pw + 2) {
                colp[0] = y0c;
                colp[1] = y1c;
            }
            zq v0, v1;
            zrot_cols(c, sn, ucar, u1, v0, v1);
            ucol[0] = v0;
            ucar = v1;
        }
        SSTAMP(2)
        __syncthreads();
        SSTAMP(3)
        ++pw;
        rowp += QLD;
        colp += 1;
        ucol += 1;
    }
    ucol[0] = ucar;                                                    // column pw(t1) + 1 of U
    __syncthreads();
#undef SSTAMP
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the chase with every wave computing its OWN next rotation.
//
// zchase3 gave the rotation of step t+1 to four "look-ahead" waves (four bulges each, one lane per bulge), fed through
// a mailbox and a parameter table in LDS.  That put the ~150-instruction scalar chain on 4 lanes of 4 waves and cost every
// wave and step three parameter reads and one mailbox store, the look-ahead waves five reads and four stores more: 32 of
// the 112 ds_write_b128 of a chain step (13 LDS cycles each -- the resource the step is bound by, profiles/r03_c5_*).
// But the two numbers the next rotation is computed from are RESULTS OF THE WAVE'S OWN COLUMN OPERATION: with the bulge
// at window row pw, the column operation on columns pw, pw+1 (lane = row) leaves the new H(pw+1, pw) in lane pw+1 and the
// new bulge H(pw+2, pw) in lane pw+2 of its first result.  Two lane broadcasts (8 v_readlane_b32) and a rotation computed
// on wave-uniform values replace mailbox, table and look-ahead lanes: no cross-wave traffic besides the window itself,
// 5 LDS reads + 5 stores per wave and step instead of 8-13 + 6-10, all waves symmetric (no priority games).  The rotation
// is computed at the START of the next row phase, behind the row loads: a phase begins with all 16 waves queueing on the
// LDS pipe, so every wave has a few hundred cycles of load latency to fill.
// The arithmetic is that of zchase3 / zchase3_steady, element for element.
// ---------------------------------------------------------------------------------------------------

// zgivens on WAVE-UNIFORM operands.  Fast path (one reciprocal square root): with d = 1 / sqrt(|a0|^2 (|a0|^2 + |a1|^2)),
//     c = |a0|^2 d,   s = a0 conj(a1) d,   rho = a0 (|a0|^2 + |a1|^2) d;
// operands whose squares leave the safe range take zgivens (uniform branch, rare: sub-diagonal entries below 1e-70 of a
// matrix scaled to max |entry| in [1, 2)).
template <bool RHO>
__device__ __forceinline__ void zgivens_u(zq a0, zq a1, double& c, zq& sn, zq& rho) {
    const double n0 = fma(a0.re, a0.re, a0.im * a0.im);
    const double n1 = fma(a1.re, a1.re, a1.im * a1.im);
    const double nn = n0 + n1;
    if (n0 >= 1e-140 && nn <= 1e140) {                                  // uniform
        const double d = fast_rsqrt(n0 * nn);
        c = n0 * d;
        const zq t = qmulc(a0, a1);
        sn = {t.re * d, t.im * d};
        if (RHO) {
            const double g = nn * d;
            rho = {a0.re * g, a0.im * g};
        }
    } else {
        zgivens(a0, a1, c, sn, rho);
    }
}

__device__ __forceinline__ zq qbcast_u(zq x, int src) {                 // src wave-uniform
    const int s = __builtin_amdgcn_readfirstlane(src);
    return {__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x.re), s),
                             __builtin_amdgcn_readlane(__double2loint(x.re), s)),
            __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x.im), s),
                             __builtin_amdgcn_readlane(__double2loint(x.im), s))};
}

// General body: bulges being introduced, leaving the active block, clipped windows.  Same window protocol as zchase3:
// phase 1 = rows p, p+1 of the window (lane = column), phase 2 = columns p, p+1 of the window and of U (lane = row).
template <bool WITHU>
__device__ __forceinline__ void zchase4(zq* __restrict__ Wn, zq* __restrict__ Un, const zq* __restrict__ shifts, int s,
                                        int wd, int l, int i, int t0, int t1, int lane, int wv, const bool prof,
                                        long long (&cst)[4]) {
    long long tl = prof ? (long long)__builtin_readcyclecounter() : 0;
#define CSTAMP(k)                                                  \
    if (prof) {                                                    \
        const long long now_ = __builtin_readcyclecounter();       \
        cst[k] += now_ - tl;                                       \
        tl = now_;                                                 \
    }
    const int b = wv;
    double c = 1.0;
    zq sn = {0.0, 0.0}, rho = {0.0, 0.0};
    zq a0n = {0.0, 0.0}, a1n = {0.0, 0.0};                             // pair the coming step's rotation annihilates
    bool have = false;                                                 // (a0n, a1n) valid
    zq ucar = {0.0, 0.0};
    bool ucv = false;                                                  // ucar holds column p of U
    for (int t = t0; t <= t1; ++t) {
        const int tb = t - 2 * b;
        const int p = l + tb;
        const bool act = tb >= 0 && p <= i - 1;                        // wave-uniform
        const int pw = p - s;
        // ---- phase 1: rows p, p+1 of the window (lane = column) ----
        if (act) {
            const zq r0 = Wn[pw * QLD + lane], r1 = Wn[(pw + 1) * QLD + lane];   // in flight under the rotation
            if (!have) {
                // first step of the stage (pair in column p-1 of the bulge's own rows: nobody else writes them in this
                // phase) or introduction of the bulge (first column of H - shift I)
                const bool intro = tb == 0;
                const int pc = intro ? pw : pw - 1;
                a0n = Wn[pw * QLD + pc];
                a1n = Wn[(pw + 1) * QLD + pc];
                if (intro) a0n = qsub(a0n, shifts[b]);
            }
            zgivens_u<true>(a0n, a1n, c, sn, rho);
            zq y0, y1;
            zrot_rows(c, sn, r0, r1, y0, y1);
            // column p-1 holds the bulge this rotation annihilates: lane p-1 stores (rho, 0) with the same two stores
            const bool bl_lane = tb >= 1 && lane == pw - 1;
            if ((lane >= pw && lane < wd) || bl_lane) {
                Wn[pw * QLD + lane] = qsel(bl_lane, rho, y0);
                Wn[(pw + 1) * QLD + lane] = qsel(bl_lane, zq{0.0, 0.0}, y1);
            }
        }
        CSTAMP(0)
        __syncthreads();
        CSTAMP(1)
        // ---- phase 2: columns p, p+1 of the window and of U (lane = row); the pair of the wave's next rotation ----
        have = false;
        if (act) {
            const int rmaxw = ((p + 2 < i) ? p + 2 : i) - s;
            const zq c0 = Wn[lane * QLD + pw], c1 = Wn[lane * QLD + pw + 1];
            zq ul = {0.0, 0.0}, u1 = {0.0, 0.0};
            if (WITHU) {
                if (!ucv) ul = Un[lane * QLD + pw];
                u1 = Un[lane * QLD + pw + 1];
            }
            zq y0, y1;
            zrot_cols(c, sn, c0, c1, y0, y1);
            if (lane <= rmaxw) {
                Wn[lane * QLD + pw] = y0;
                Wn[lane * QLD + pw + 1] = y1;
            }
            const bool more = t < t1 && p + 1 <= i - 1;                // the bulge moves on inside this window
            if (more) {
                // new H(p+1, p) and the new bulge H(p+2, p): lanes pw+1, pw+2 of the first result column
                a0n = qbcast_u(y0, pw + 1);
                a1n = qbcast_u(y0, pw + 2);
                have = true;
            }
            if (WITHU) {
                // column p of U is still in this wave's registers from the previous step (no other bulge touches it)
                const zq u0 = qsel(ucv, ucar, ul);
                zq v0, v1;
                zrot_cols(c, sn, u0, u1, v0, v1);
                ucv = more;
                ucar = v1;
                if (lane < wd) {
                    Un[lane * QLD + pw] = v0;
                    if (!more) Un[lane * QLD + pw + 1] = v1;
                }
            }
        }
        CSTAMP(2)
        __syncthreads();
        CSTAMP(3)
    }
#undef CSTAMP
}

// Interior stages (conditions and constants of zchase3_steady): every flag of the general body is a constant.
template <bool PROF>
__device__ __forceinline__ void zchase4_steady(zq* __restrict__ Wn, zq* __restrict__ Un, int t0, int t1, int lane, int wv,
                                               long long (&cst)[4]) {
    long long tl = PROF ? (long long)__builtin_readcyclecounter() : 0;
#define SSTAMP(k)                                                  \
    if (PROF) {                                                    \
        const long long now_ = __builtin_readcyclecounter();       \
        cst[k] += now_ - tl;                                       \
        tl = now_;                                                 \
    }
    constexpr int P0 = 2 * (QM - 1) + 1;                               // window row of bulge 0 at step t0
    int pw = P0 - 2 * wv;
    zq* rowp = Wn + pw * QLD + lane;                                   // (pw, lane), advances by QLD
    zq* colp = Wn + lane * QLD + pw;                                   // (lane, pw), advances by 1
    zq* ucol = Un + lane * QLD + pw;
    zq a0n = Wn[pw * QLD + pw - 1], a1n = Wn[(pw + 1) * QLD + pw - 1];   // own rows: no barrier needed
    zq ucar = ucol[0];                                                 // column pw of U: carried from step to step
    for (int t = t0; t <= t1; ++t) {
        double c;
        zq sn, rho;
        // ---- phase 1: rows pw, pw+1 of the window (lane = column); column pw-1 goes through the rotation too ----
        {
            const zq r0 = rowp[0], r1 = rowp[QLD];                      // in flight while the rotation is computed
            zgivens_u<false>(a0n, a1n, c, sn, rho);
            zq y0, y1;
            zrot_rows(c, sn, r0, r1, y0, y1);
            if (lane >= pw - 1) {
                rowp[0] = y0;
                rowp[QLD] = y1;
            }
        }
        SSTAMP(0)
        __syncthreads();
        SSTAMP(1)
        // ---- phase 2: columns pw, pw+1 of the window and of U (lane = row); pair of the next rotation ----
        {
            const zq c0 = colp[0], c1 = colp[1];
            const zq u1 = ucol[1];
            zq y0c, y1c;
            zrot_cols(c, sn, c0, c1, y0c, y1c);
            if (lane <= pw + 2) {
                colp[0] = y0c;
                colp[1] = y1c;
            }
            a0n = qbcast_u(y0c, pw + 1);                                // (unused after step t1)
            a1n = qbcast_u(y0c, pw + 2);
            zq v0, v1;
            zrot_cols(c, sn, ucar, u1, v0, v1);
            ucol[0] = v0;
            ucar = v1;
        }
        SSTAMP(2)
        __syncthreads();
        SSTAMP(3)
        ++pw;
        rowp += QLD;
        colp += 1;
        ucol += 1;
    }
    ucol[0] = ucar;                                                    // column pw(t1) + 1 of U
    __syncthreads();
#undef SSTAMP
}

// Deferred update of one side by the accumulated U (both as  out(f, g) = sum_k X(k, f) U'(k, g)  on the matrix cores):
//   LEFT = false: rows [f0, f0 + nf) above the window:    H(f0 + f, s + g) <- sum_k H(f0 + f, s + k) U(k, g)
//   LEFT = true : columns [f0, f0 + nf) right of it:      H(s + g, f0 + f) <- sum_k conj(U(k, g)) H(s + k, f0 + f)
// X is staged in the (free) window array, 64 x 64 at a time; the next chunk's loads fly during the products.
template <bool LEFT>
__device__ __forceinline__ void zapply_u(zq* __restrict__ H, int n, zq* __restrict__ Xn, const zq* __restrict__ Un, int s,
                                         int wd, int f0, int nf, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    const int ft = wv >> 2, gt = wv & 3;
    const int cl = lane & 15, kk = lane >> 4;
    const int ksteps = (wd + 3) / 4;
    const int nchunk = (nf + 63) / 64;
    zq pre[4];
    auto gaddr = [&](int e, int fbase) -> int64_t {
        // element e of a chunk: LEFT: f = e >> 6, k = e & 63 ; else k = e >> 6, f = e & 63
        const int hi = e >> 6, lo = e & 63;
        const int f = LEFT ? hi : lo, kx = LEFT ? lo : hi;
        const int fc = (fbase + f < f0 + nf) ? fbase + f : f0 + nf - 1;
        const int kc = (kx < wd) ? kx : wd - 1;
        return LEFT ? ((int64_t)fc * n + (s + kc)) : ((int64_t)(s + kc) * n + fc);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) pre[u] = H[gaddr(tid + u * kZL, f0)];
    for (int ch = 0; ch < nchunk; ++ch) {
        const int fbase = f0 + ch * 64;
        const int nfc = (f0 + nf - fbase < 64) ? f0 + nf - fbase : 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * kZL;
            const int hi = e >> 6, lo = e & 63;
            const int f = LEFT ? hi : lo, kx = LEFT ? lo : hi;
            const bool ok = f < nfc && kx < wd;
            Xn[kx * QLD + f] = qsel(ok, pre[u], zq{0.0, 0.0});
        }
        __syncthreads();
        if (ch + 1 < nchunk) {
#pragma unroll
            for (int u = 0; u < 4; ++u) pre[u] = H[gaddr(tid + u * kZL, fbase + 64)];
        }
        // complex product with three real ones (Gauss):  k1 = (xr + xi) ur,  k2 = xr (ui - ur),  k3 = xi (ur + ui);
        // re = k1 - k3,  im = k1 + k2  -- the fp64 matrix pipe is what bounds this phase, a quarter of it is saved
        v4d k1 = {0.0, 0.0, 0.0, 0.0}, k2 = {0.0, 0.0, 0.0, 0.0}, k3 = {0.0, 0.0, 0.0, 0.0};
        const bool tact = ft * 16 < nfc && gt * 16 < wd;
        if (tact) {
            // U is banded (bulges move down: k - g <= QM, g - k <= QNBS): skip the k-steps that only meet zeros
            int tlo = (gt * 16 - QNBS) >> 2;
            if (tlo < 0) tlo = 0;
            int thi = (gt * 16 + 15 + QM + 4) >> 2;
            if (thi > ksteps) thi = ksteps;
            for (int t = tlo; t < thi; ++t) {
                const zq x = Xn[(4 * t + kk) * QLD + ft * 16 + cl];
                const zq u = Un[(4 * t + kk) * QLD + gt * 16 + cl];
                const double ui = LEFT ? -u.im : u.im;
                k1 = mfma64(x.re + x.im, u.re, k1);
                k2 = mfma64(x.re, ui - u.re, k2);
                k3 = mfma64(x.im, u.re + ui, k3);
            }
        }
        const v4d dr = k1 - k3, di = k1 + k2;
#ifdef EGS_ZAPPLY_VIA_LDS
        __syncthreads();
        if (tact) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Xn[(gt * 16 + cl) * QLD + ft * 16 + kk + 4 * q] = {dr[q], di[q]};
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + u * kZL;
            const int hi = e >> 6, lo = e & 63;
            const int f = LEFT ? hi : lo, g = LEFT ? lo : hi;
            if (f < nfc && g < wd) {
                const int64_t ad = LEFT ? ((int64_t)(fbase + f) * n + (s + g)) : ((int64_t)(s + g) * n + (fbase + f));
                H[ad] = Xn[g * QLD + f];
            }
        }
        __syncthreads();
#else
        // Round 4: the result tile goes from the accumulators straight to memory.  Register q of lane (cl, kk) holds
        // out(f = 16 ft + kk + 4 q, g = 16 gt + cl): per store instruction the four kk lanes of a column write 64
        // contiguous bytes (LEFT = false: f runs along a column of H) or the sixteen cl lanes 256 (LEFT = true: g does) --
        // good enough for 64 KB per chunk, and the transposition through the window array with its two barriers (and
        // 1,100 LDS cycles) per chunk is gone.
        if (tact) {
            const int g = gt * 16 + cl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = ft * 16 + kk + 4 * q;
                if (f < nfc && g < wd) {
                    const int64_t ad = LEFT ? ((int64_t)(fbase + f) * n + (s + g)) : ((int64_t)(s + g) * n + (fbase + f));
                    H[ad] = {dr[q], di[q]};
                }
            }
        }
        __syncthreads();                                              // every wave has read X: the next chunk may be staged
#endif
    }
}

// prof: NULL, or double[nmat][12] receiving shader-clock cycles per phase (scan, shift generator / small blocks,
// window load + store, chase, deferred updates, single-window chase), the sweep / stage counts, and the split of the
// chase as wave 0 sees it (row phase, barrier, column phase, barrier).
// PROF = false (the product instantiation): `prof` is ignored and none of the accumulators exists -- as run-time
// optional state they were eleven 64-bit values live across the whole kernel, spilled at entry and reloaded inside the
// sweep loop (profiles/r04_kernel_resources.json: 31 scratch instructions in loops).
template <bool PROF>
__global__ __launch_bounds__(kZL) void zmsqr_kernel(double* __restrict__ work, int n, double* __restrict__ w,
                                                    int32_t* __restrict__ info_out, double* __restrict__ prof_arg,
                                                    int zgeneral_only) {
    double* const prof = PROF ? prof_arg : nullptr;
    extern __shared__ __attribute__((aligned(16))) double dynq[];
    __shared__ int s_idx;
    __shared__ int32_t s_info;
    __shared__ int32_t s_binfo[QM];
    // the wave index is uniform by construction; as a scalar, what is derived from it stays out of the vector registers
    const int tid0 = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    int tid = tid0, lane = tid0 & 63;
    const int64_t mat = blockIdx.x;
    if (info_out[mat] != 0) return;                                   // non-finite input, flagged by the reduction
    zq* Wn = reinterpret_cast<zq*>(dynq);        // [QW][QLD]
    zq* Un = Wn + QW * QLD;                      // [QW][QLD]
    zq* Ss = Un + QW * QLD;                      // [QSMALL][QSLD]
    zq* shifts = Ss + QSMALL * QSLD;             // [QM]
    zq* sw = shifts + QM;                        // [QSMALL] eigenvalues of a small block
    zq* prm = sw + QSMALL;                       // [2][QM][3] rotation parameters of the chase
    zq* mbx = prm + 6 * QM;                      // [2][QM][3] look-ahead mailbox
    // Deflated blocks of <= QSMALL rows are independent problems for ONE wavefront each.  Solved where they appear they
    // cost the whole workgroup that wave's latency (1.98 of 41.7 Mcycles per C5 matrix, fifteen waves waiting); so they
    // are only RECORDED (start row, order) and solved after the iteration, sixteen at a time, one per wave, each on its
    // own slab of the then idle window arrays (round 5).  Nothing touches their rows / columns in between: the
    // eigenvalue-only iteration confines every update to the active block above them.
    int* dlist = reinterpret_cast<int*>(mbx + 6 * QM);    // [kZDeferMax][2]
    int ndefer = 0;
    zq* H = reinterpret_cast<zq*>(work) + mat * (int64_t)n * n;
    zq* wout = reinterpret_cast<zq*>(w) + mat * (int64_t)n;
    const int sexp = (int)w[mat * 2 * (int64_t)n];                   // scaling exponent left by the Hessenberg stage
    const double rescale = (sexp != 0) ? scalbn(1.0, sexp) : 1.0;
    __syncthreads();                                                  // everybody has read it before slot 0 can be written
#define HG(r, c) H[(int64_t)(c) * n + (r)]
    const double ulp = 2.220446049250313e-16;
    const double safmin = 2.2250738585072014e-308;
    // absolute: the Hessenberg stage scaled the matrix to max |entry| in [1, 2) and leaves columns whose norm is below
    // 2^-400 (4e-121) unreduced (their squares underflow): everything down there is negligible and has to deflate at once,
    // or the iteration grinds on blocks that are not even Hessenberg (c ones(n, n), n >= 160: hundreds of sweeps, one
    // matrix in ten flagged as not converged).  1e-100 of the largest entry is 84 orders below the rounding level.
    const double smlnum = 1e-100;
    (void)safmin;
    int info = 0;
    int i = n - 1;
    int stall = 0;
    long long pacc0 = 0, pacc1 = 0, pacc2 = 0, pacc3 = 0, pacc4 = 0, pacc5 = 0, plast = 0;
    int nsweep = 0, nstage = 0;
    long long cst[4] = {0, 0, 0, 0};
#define ZSTAMP(acc)                                            \
    if (prof) {                                                \
        const long long now_ = __builtin_readcyclecounter();   \
        acc += now_ - plast;                                   \
        plast = now_;                                          \
    }
    if (prof) plast = __builtin_readcyclecounter();
    while (i >= 0) {
        // The chase / update bodies derive a dozen per-lane LDS addresses from the thread index; hoisted out of this loop they
        // stay live -- spilled -- across it (128 registers per lane at 16 waves).  An opaque per-sweep copy keeps their
        // derivation (a few integer instructions) inside the sweep.
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        // ---- deflation scan: largest k in (0, i] with a negligible subdiagonal ----
        if (tid == 0) s_idx = 0;
        __syncthreads();
        for (int k = 1 + tid; k <= i; k += kZL) {
            bool small = false;
            const double hkk1 = qabs1(HG(k, k - 1));
            if (hkk1 <= smlnum) {
                small = true;
            } else {
                const zq hkk = HG(k, k), hk1k1 = HG(k - 1, k - 1);
                double tst = qabs1(hk1k1) + qabs1(hkk);
                if (tst == 0.0) {
                    if (k - 2 >= 0) tst += qabs1(HG(k - 1, k - 2));
                    if (k + 1 <= n - 1) tst += qabs1(HG(k + 1, k));
                }
                if (hkk1 <= ulp * tst) {
                    const double hk1k = qabs1(HG(k - 1, k));
                    const double ab = fmax(hkk1, hk1k), ba = fmin(hkk1, hk1k);
                    const double df = qabs1(qsub(hk1k1, hkk));
                    const double aa = fmax(qabs1(hkk), df), bb = fmin(qabs1(hkk), df);
                    const double sm = aa + ab;
                    if (ba * (ab / sm) <= fmax(smlnum, ulp * (bb * (aa / sm)))) small = true;
                }
            }
            if (small) atomicMax(&s_idx, k);
        }
        __syncthreads();
        const int l = s_idx;
        if (l > 0 && tid == 0) HG(l, l - 1) = {0.0, 0.0};
        __syncthreads();
        if (l == i) {
            if (tid == 0) wout[i] = qscale(HG(i, i), rescale);
            --i;
            stall = 0;
            continue;
        }
        const int na = i - l + 1;
        ZSTAMP(pacc0)
        if (na <= QSMALL) {
#ifndef EGS_ZSMALL_INLINE
            // ---- small deflated block: recorded, solved after the iteration (one wave per block, sixteen at a time) ----
            if (tid == 0) {
                dlist[2 * ndefer] = l;
                dlist[2 * ndefer + 1] = na;
            }
            ++ndefer;                                                 // <= n / 2 entries (every block has >= 2 rows)
            i = l - 1;
            stall = 0;
            ZSTAMP(pacc1)
            continue;
#else
            // ---- small deflated block: one wavefront finishes it (rounds 2-4; A/B switch) ----
            for (int e = tid; e < na * na; e += kZL) {
                const int c = e / na, r = e - c * na;
                Ss[r * QSLD + c] = qsel(r <= c + 1, HG(l + r, l + c), zq{0.0, 0.0});
            }
            if (tid == 0) s_info = 0;
            __syncthreads();
            if (wv == 0) zhqr_wave(ZViewT<0, false>{Ss, QSLD}, na, lane, reinterpret_cast<double*>(sw), &s_info);
            __syncthreads();
            if (s_info != 0) {
                info = l + s_info;
                break;
            }
            if (tid < na) wout[l + tid] = qscale(sw[tid], rescale);
            i = l - 1;
            stall = 0;
            __syncthreads();
            ZSTAMP(pacc1)
            continue;
#endif
        }
        if (++stall > 300) {
            info = i + 1;
            break;
        }
        // ---- shifts ----
#ifdef EGS_ZSHIFT_REUSE
        // experiment (profiles/r05_variants.md): every EGS_ZSHIFT_REUSE-th sweep of a stall sequence keeps the previous
        // sweep's shifts instead of running the one-wave shift generator while fifteen waves wait
        if (stall > 1 && (stall % EGS_ZSHIFT_REUSE) == 0 && stall % 6 != 0) {
            // shifts[] still holds the previous sweep's
        } else
#endif
        if (stall % 6 == 0) {
            if (tid < QNS) {
                const int j = i - QNS + 1 + tid;
                const zq d = HG(j, j);
                sw[tid] = {d.re + 0.75 * qabs1(HG(j, j - 1)), d.im};
            }
            __syncthreads();
        } else {
            for (int e = tid; e < QNS * QNS; e += kZL) {
                const int c = e / QNS, r = e - c * QNS;
                Ss[r * QSLD + c] = qsel(r <= c + 1, HG(i - QNS + 1 + r, i - QNS + 1 + c), zq{0.0, 0.0});
            }
            if (tid == 0) s_info = 0;
            __syncthreads();
            if (wv == 0) zhqr_wave<ZViewT<0, false>, true>(ZViewT<0, false>{Ss, QSLD}, QNS, lane, reinterpret_cast<double*>(sw), &s_info);
            __syncthreads();
            if (s_info != 0) {                                        // shift generator stalled: fall back to the diagonal
                if (tid < QNS) sw[tid] = HG(i - QNS + 1 + tid, i - QNS + 1 + tid);
                __syncthreads();
            }
        }
#ifdef EGS_ZSHIFT_REUSE
        if (!(stall > 1 && (stall % EGS_ZSHIFT_REUSE) == 0 && stall % 6 != 0))
#endif
        if (tid < QM) shifts[tid] = sw[tid * QNS / QM];
        __syncthreads();
#ifdef EGS_ZPROF_GEN
        ZSTAMP(pacc0)            // profiling variant: the shift generator goes to slot 0 (with the scan), slot 1 keeps the small blocks
#else
        ZSTAMP(pacc1)
#endif
        ++nsweep;
        const int tmax = (i - 1 - l) + 2 * (QM - 1);
        if (na <= QW) {
            // ---- the whole active block fits one window: chase in place, nothing to update outside ----
            for (int e = tid; e < QW * QW; e += kZL) {
                const int c = e >> 6, r = e & 63;
                const bool ok = r < na && c < na;
                Wn[r * QLD + c] = qsel(ok, HG(l + (r < na ? r : 0), l + (c < na ? c : 0)), zq{0.0, 0.0});
            }
            __syncthreads();
            #ifdef EGS_ZCHASE_LOOKAHEAD
            zchase3<false>(Wn, Un, prm, mbx, shifts, l, na, l, i, 0, tmax, lane, wv, false, cst);
#else
            zchase4<false>(Wn, Un, shifts, l, na, l, i, 0, tmax, lane, wv, false, cst);
#endif
            for (int e = tid; e < QW * QW; e += kZL) {
                const int c = e >> 6, r = e & 63;
                if (r < na && c < na) HG(l + r, l + c) = Wn[r * QLD + c];
            }
            __syncthreads();
            ZSTAMP(pacc5)
            continue;
        }
        for (int kst = 0; kst * QNBS <= tmax; ++kst) {
            const int t0 = kst * QNBS;
            const int t1 = (t0 + QNBS - 1 < tmax) ? t0 + QNBS - 1 : tmax;
            int s = l + t0 - 2 * (QM - 1) - 1;
            if (s < l) s = l;
            int e1 = l + t1 + 2;
            if (e1 > i) e1 = i;
            const int wd = e1 - s + 1;
            for (int e = tid; e < QW * QW; e += kZL) {
                const int c = e >> 6, r = e & 63;
                const bool ok = r < wd && c < wd;
                Wn[r * QLD + c] = qsel(ok, HG(s + (r < wd ? r : 0), s + (c < wd ? c : 0)), zq{0.0, 0.0});
                Un[r * QLD + c] = zq{(r == c) ? 1.0 : 0.0, 0.0};
            }
            __syncthreads();
            ZSTAMP(pacc2)
            ++nstage;
            if (t0 >= 2 * (QM - 1) + 1 && l + t1 + 3 <= i && wd == QW && !zgeneral_only) {
                // (EGS_ZCHASE_STAMPS builds stamp this body too: 26.2 against 29.8 Mcycles of chase per C5 matrix,
                // profiles/r03_c5_phase_cycles.json)
#ifdef EGS_ZCHASE_LOOKAHEAD
                zchase3_steady<kZSteadyStamps>(Wn, Un, prm, mbx, t0, t1, lane, wv, cst);
            } else
                zchase3<true>(Wn, Un, prm, mbx, shifts, s, wd, l, i, t0, t1, lane, wv, prof != nullptr, cst);
#else
                zchase4_steady<kZSteadyStamps>(Wn, Un, t0, t1, lane, wv, cst);
            } else
                zchase4<true>(Wn, Un, shifts, s, wd, l, i, t0, t1, lane, wv, prof != nullptr, cst);
#endif
            ZSTAMP(pacc3)
            for (int e = tid; e < QW * QW; e += kZL) {
                const int c = e >> 6, r = e & 63;
                if (r < wd && c < wd) HG(s + r, s + c) = Wn[r * QLD + c];
            }
            __syncthreads();
            ZSTAMP(pacc2)
            if (s > l) zapply_u<false>(H, n, Wn, Un, s, wd, l, s - l, tid);
            if (e1 < i) zapply_u<true>(H, n, Wn, Un, s, wd, e1 + 1, i - e1, tid);
            ZSTAMP(pacc4)
        }
    }
#ifndef EGS_ZSMALL_INLINE
    // ---- the recorded small blocks: wave w solves block 16 round + w on slab w of the window arrays ----
    // (also when the iteration above gave up on a block: the recorded ones lie below it and are still valid problems)
    if (ndefer > 0) {
        __syncthreads();                                              // the list is complete and the windows are idle
        if (tid == 0) s_info = 0;
        if (tid < QM) s_binfo[tid] = 0;
        __syncthreads();
        zq* slab = (wv < 8 ? Wn : Un) + (wv & 7) * (QSMALL * QSLD);  // 8 slabs of 16 x 17 per window array
        for (int b0 = 0; b0 < ndefer; b0 += QM) {
            const int b = b0 + wv;
            if (b < ndefer) {
                const int bl = dlist[2 * b], bn = dlist[2 * b + 1];
                for (int e = lane; e < bn * bn; e += kWave) {
                    const int c = e / bn, r = e - c * bn;
                    slab[r * QSLD + c] = qsel(r <= c + 1, HG(bl + r, bl + c), zq{0.0, 0.0});
                }
                wave_lds_sync();
                // eigenvalues go straight to their slots of w (scaled back); a block that fails flags the matrix
                zhqr_wave(ZViewT<0, false>{slab, QSLD}, bn, lane, reinterpret_cast<double*>(wout + bl), &s_binfo[wv], rescale);
                wave_lds_sync();
                if (lane == 0 && s_binfo[wv] != 0) {
                    atomicMax(&s_info, bl + s_binfo[wv]);
                    s_binfo[wv] = 0;
                }
            }
        }
        __syncthreads();
        if (s_info > info) info = s_info;
        ZSTAMP(pacc1)
    }
#endif
    if (info != 0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int j = tid; j < info; j += kZL) wout[j] = {qnan, qnan};
    }
    if (tid == 0) info_out[mat] = info;
    if (prof && tid == 0) {
        double* pr = prof + mat * 12;
        pr[0] = (double)pacc0;
        pr[1] = (double)pacc1;
        pr[2] = (double)pacc2;
        pr[3] = (double)pacc3;
        pr[4] = (double)pacc4;
        pr[5] = (double)pacc5;
        pr[6] = (double)nsweep;
        pr[7] = (double)nstage;
        for (int x = 0; x < 4; ++x) pr[8 + x] = (double)cst[x];   // chase detail, wave 0: row phase, barrier, column phase, barrier
    }
#undef ZSTAMP
#undef HG
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static size_t zhess_lds_bytes(int n, int nb) {
    const int ldv = zl_np(n) + 1;
    return ((size_t)2 * nb * ldv + (size_t)nb * nb + kZL + 3 * nb) * sizeof(zq);
}

static int zhess_nb(int n) { return n <= 256 ? 16 : (n <= 512 ? 8 : 4); }

size_t eig_zlarge_workspace_bytes(int64_t nmat, int n) { return (size_t)nmat * n * n * sizeof(zq); }

int zhess_blocked_launch(const void* a, bool in_real, int64_t nmat, int n, int lda, int64_t stride, double* w,
                         int32_t* info, void* workspace, hipStream_t stream) {
    const int nb = zhess_nb(n);
    const size_t dyn = zhess_lds_bytes(n, nb);
    const int balance = getenv("EGS_NO_BALANCE") ? 0 : 1;
    const double* ap = static_cast<const double*>(a);
    double* wk = static_cast<double*>(workspace);
    auto launch = [&](auto kern) -> int {
        EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)dyn));
        hipLaunchKernelGGL(kern, dim3((unsigned)nmat), dim3(kZL), dyn, stream, ap, in_real ? 1 : 0, n, lda, stride, w,
                           info, wk, balance);
        EGS_LAUNCH_CHECK();
        return 0;
    };
    if (nb == 16) return launch(zhess_blocked_kernel<16, 2>);
    if (nb == 8) return launch(zhess_blocked_kernel<8, 2>);
    return launch(zhess_blocked_kernel<4, 4>);
}

int zmsqr_launch(int64_t nmat, int n, double* w, int32_t* info, void* workspace, double* prof, hipStream_t stream) {
    const size_t dyn = ((size_t)2 * QW * QLD + QSMALL * QSLD + QM + QSMALL + 6 * QM + 6 * QM) * sizeof(zq) +
                       (size_t)2 * 512 * sizeof(int);                 // + the list of deferred small blocks (n <= 1024)
    const int general_only = getenv("EGS_ZCHASE_GENERAL") ? 1 : 0;   // A/B switch: the general chase body in every stage
    if (prof != nullptr) {
        EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(zmsqr_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(zmsqr_kernel<true>, dim3((unsigned)nmat), dim3(kZL), dyn, stream, static_cast<double*>(workspace),
                           n, w, info, prof, general_only);
    } else {
        EGS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(zmsqr_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(zmsqr_kernel<false>, dim3((unsigned)nmat), dim3(kZL), dyn, stream, static_cast<double*>(workspace),
                           n, w, info, prof, general_only);
    }
    EGS_LAUNCH_CHECK();
    return 0;
}

int eig_zlarge_launch(const void* a, bool in_real, int64_t nmat, int n, int lda, int64_t stride, double* w,
                      int32_t* info, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (nmat == 0 || n == 0) return 0;
    if (workspace_bytes < eig_zlarge_workspace_bytes(nmat, n)) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, eig_zlarge_workspace_bytes(nmat, n));
        return -9;
    }
    const int st = zhess_blocked_launch(a, in_real, nmat, n, lda, stride, w, info, workspace, stream);
    if (st) return st;
    return zmsqr_launch(nmat, n, w, info, workspace, nullptr, stream);
}

}  // namespace egs
