// K5 (real, n = 64), first stage (device functions shared by hess_reg_kernel and the fused eigenvalue kernel): Householder reduction to upper-Hessenberg form with the MATRIX IN REGISTERS.
// One wavefront per matrix, 64 fp64 elements per lane, no LDS traffic for the matrix itself -- LDS only carries the
// three 64-vectors of a step (1.5 KB per wave), so residency is limited by registers (2 waves per SIMD) instead of by
// LDS (the LDS-resident hess_real_kernel keeps ONE wave per SIMD and spends its time waiting for LDS round trips).
//
// Lane grid.  q = lane bits 2-3 (4 values), c = lane bits 0-1 and 4-5 (16 values).  Lane (q, c) owns the elements
//     A(4R + q, 16C + c),   R = 0..15, C = 0..3                                     (register A[R][C])
// so a ROW sum (over columns) is 4 local terms + a reduction over the 16 lanes that share q, and a COLUMN sum (over
// rows) is 16 local terms + a reduction over the 4 lanes that share c.  The 16-lane reduction is two
// v_mfma_f64_4x4x4_4b_f64 with a ones operand: that instruction contracts over lane bits 4-5 and moves operand bits
// 0-1 to result bits 4-5 (mapping measured with tools/probe/mfma444.hip), so applying it twice sums bits 0-1 and 4-5
// and leaves the total in all 16 lanes.  The 4-lane reduction is two DPP row rotations (8, then 4).  Vectors live in
// LDS in the order rho(i) = (i & 3) * 16 + (i >> 2): the 16 entries a lane needs for its rows are then contiguous
// (eight ds_read_b128), the 4 entries for its columns are immediate offsets from one per-lane address.
//
// Per column j (xGEHD2 step, rank-2 form  H A H = A - w v^T - v y^T  as in hess_real_kernel):
//   owners of column j store it to LDS (8 ds_write_b128) -> lane i reads x_i, norm by the two reductions, reflector,
//   v back to LDS -> every lane reads v for its rows and columns -> p = A v (row sums), q = A^T v (column sums),
//   w and y in registers -> 2 FMAs per element.  Column blocks left of the panel and row blocks above it are skipped
//   at compile time (the panel index J is a template parameter).  The annihilated part of column j is left to the
//   update (it becomes rounding-level noise that no later step reads) and is written as exact zeros at the end.
//
// Replaces, for n = 64, the first stage behind np.linalg.eigvals at analysis/eval_eig.py:296 (BASELINE config C2).
#pragma once

#include "egs_common.h"

namespace egs {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int kRegVec = 3 * 64;             // doubles of LDS per wave: three 64-vectors
constexpr int kRegNonFinite = -2147483647;  // same sentinel as eig_real_lds.hip: the input holds NaN / Inf

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// sum over the four lanes that differ in lane bits 2-3 (result in all four)
__device__ __forceinline__ double sum_q(double x) {
    x += dpp_mov<0x128>(x);                 // row_ror:8
    x += dpp_mov<0x124>(x);                 // row_ror:4
    return x;
}

// sum over the 16 lanes that share lane bits 2-3 (result in all 16)
__device__ __forceinline__ double sum_c(double x) {
    const double t = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);
    return __builtin_amdgcn_mfma_f64_4x4x4f64(t, 1.0, 0.0, 0, 0, 0);
}

// All-lanes maximum.  The partner addresses are derived from the caller's lane index (the fused kernel's per-matrix
// OPAQUE copy): __shfl_xor derives them from the hardware lane id, which is invariant across the per-matrix loop -- the six
// index vectors were hoisted out of it and spilled across the whole kernel.
__device__ __forceinline__ double wave_max(double v, int lane) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int src = (lane ^ off) << 2;
        const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
        const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
        v = fmax(v, __hiloint2double(hi, lo));
    }
    return v;
}

struct RegLane {
    int lane, q, c;
    int rho_lane;      // rho(lane): where entry `lane` of a vector lives
    int col_base;      // rho(16 C + c) = col_base + 4 C
};

// off-diagonal 1-norms of the rows and columns of the window [k0, l0] (xGEBAL's c and r), row sums replicated over the
// lanes of a row group (rs[R] for row 4R + q), column sums over those of a column (cs[C] for column 16C + c)
__device__ __forceinline__ void offdiag_norms(const double (&A)[16][4], const RegLane& L, int k0, int l0,
                                              double (&rs)[16], double (&cs)[4]) {
    const int d = L.c - L.q;                // row == col  <=>  4R - 16C == c - q
#pragma unroll
    for (int C = 0; C < 4; ++C) cs[C] = 0.0;
#pragma unroll
    for (int R = 0; R < 16; ++R) {
        const int row = 4 * R + L.q;
        const bool rin = row >= k0 && row <= l0;
        double acc = 0.0;
#pragma unroll
        for (int C = 0; C < 4; ++C) {
            const int col = 16 * C + L.c;
            const bool use = rin && col >= k0 && col <= l0 && (4 * R - 16 * C != d);
            const double t = use ? fabs(A[R][C]) : 0.0;
            acc += t;
            cs[C] += t;
        }
        rs[R] = acc;
    }
#pragma unroll
    for (int R = 0; R < 16; ++R) rs[R] = sum_c(rs[R]);
#pragma unroll
    for (int C = 0; C < 4; ++C) cs[C] = sum_q(cs[C]);
}

// xGEBAL's permutation search on a row-major copy of the matrix in global memory (rare path: some row or column has
// an all-zero off-diagonal part).  Lane = row for the row scans, lane = column for the column scans, exactly the scheme
// of hess_real_kernel; the copy is accessed through volatile pointers with a device-scope fence between phases.
__device__ __forceinline__ void permute_isolated(volatile double* s, int lane, int& k0, int& l0) {
    constexpr int n = 64;
    for (int guard = 0; guard < n; ++guard) {       // rows whose off-diagonal part in columns 0..l0 is zero
        bool rz = false;
        if (lane <= l0) {
            rz = true;
            for (int j = 0; j <= l0; ++j)
                if (j != lane && s[lane * n + j] != 0.0) rz = false;
        }
        const unsigned long long m = __ballot(rz);
        if (m == 0ULL || l0 == 0) break;
        const int r = uniform(63 - __builtin_clzll(m));
        if (r != l0) {
            if (lane <= l0) {                       // swap columns r <-> l0 over rows 0..l0
                const double t = s[lane * n + r];
                s[lane * n + r] = s[lane * n + l0];
                s[lane * n + l0] = t;
            }
            __threadfence();
            {                                       // swap rows r <-> l0 over all columns
                const double t = s[r * n + lane];
                s[r * n + lane] = s[l0 * n + lane];
                s[l0 * n + lane] = t;
            }
            __threadfence();
        }
        --l0;
    }
    for (int guard = 0; guard < n; ++guard) {       // columns whose off-diagonal part in rows k0..l0 is zero
        bool cz = false;
        if (lane >= k0 && lane <= l0) {
            cz = true;
            for (int i = k0; i <= l0; ++i)
                if (i != lane && s[i * n + lane] != 0.0) cz = false;
        }
        const unsigned long long m = __ballot(cz);
        if (m == 0ULL || k0 >= l0) break;
        const int c = uniform(__builtin_ctzll(m));
        if (c != k0) {
            if (lane <= l0) {
                const double t = s[lane * n + c];
                s[lane * n + c] = s[lane * n + k0];
                s[lane * n + k0] = t;
            }
            __threadfence();
            if (lane >= k0) {
                const double t = s[c * n + lane];
                s[c * n + lane] = s[k0 * n + lane];
                s[k0 * n + lane] = t;
            }
            __threadfence();
        }
        ++k0;
    }
}

// The 16 steps (14 in the last panel) of column block J.
template <int J>
__device__ __forceinline__ void hess_panel(double (&A)[16][4], const RegLane& L, double* __restrict__ xs) {
    double* __restrict__ vs = xs + 64;          // v of the current step (xs: the column)
    constexpr int R0 = 4 * J;                       // rows above 16 J carry v = 0
    const int jend = (J == 3) ? 14 : 16;
    for (int jc = 0; jc < jend; ++jc) {
        const int j = 16 * J + jc;
        // ---- column j to LDS: its four owner lanes store their 16 rows contiguously ----
        if (L.c == jc) {
            v2d* dst = reinterpret_cast<v2d*>(xs + L.q * 16);
#pragma unroll
            for (int t = 0; t < 8; ++t) dst[t] = (v2d){A[2 * t][J], A[2 * t + 1][J]};
        }
        wave_lds_sync();
        const double xi = xs[L.rho_lane];           // lane i holds x_i = A(i, j)
        const double alpha = xs[((j + 1) & 3) * 16 + ((j + 1) >> 2)];
        const double xnorm2 = sum_q(sum_c((L.lane > j + 1) ? xi * xi : 0.0));
        // nothing to annihilate (wave-uniform): the step runs with tau = 0, which leaves A unchanged bit for bit -- a
        // branch around the step would make every register of A a loop-carried phi and costs more than the idle step
        // (a column whose squared norm is below 2^-800 counts as annihilated too: the matrix is scaled to max |entry| in
        // [1, 2), and between ~1e-162 and ~1e-150 the squares are denormal or gone while the entries are not, which
        // would make tau and v inconsistent -- a reflector that is not orthogonal, applied to the whole matrix)
        const bool live = xnorm2 != 0.0 && fma(alpha, alpha, xnorm2) >= 1.4996968138956309e-241;
        const double tiny = 1e-300;
        const double nrm2 = fma(alpha, alpha, xnorm2);
        const double rn = fast_rsqrt(fmax(nrm2, tiny));
        const double nrm = nrm2 * rn;
        const double apn = fabs(alpha) + nrm;                                    // |alpha - beta|
        const double tau = live ? apn * rn : 0.0;                                 // (beta - alpha) / beta
        const double scal = live ? copysign(fast_rcp(fmax(apn, tiny)), alpha) : 0.0;   // 1 / (alpha - beta)
        const double vi = (L.lane > j + 1) ? xi * scal : ((L.lane == j + 1) ? 1.0 : 0.0);
        vs[L.rho_lane] = vi;
        wave_lds_sync();
        // v for this lane's columns in registers; v for its rows (16 contiguous entries) is streamed from LDS where it
        // is needed: holding it would push the step over the 256-register budget of two waves per SIMD
        const v2d* vr = reinterpret_cast<const v2d*>(vs + L.q * 16);
        double vcol[4], y[4];
#pragma unroll
        for (int C = J; C < 4; ++C) {
            vcol[C] = vs[L.col_base + 4 * C];
            y[C] = 0.0;
        }
        // ---- left:  y = tau A^T v (column sums over the rows that carry v) ----
#pragma unroll
        for (int t = R0 / 2; t < 8; ++t) {
            const v2d u = vr[t];
#pragma unroll
            for (int C = J; C < 4; ++C) {
                y[C] = fma(A[2 * t][C], u.x, y[C]);
                y[C] = fma(A[2 * t + 1][C], u.y, y[C]);
            }
        }
#pragma unroll
        for (int C = J; C < 4; ++C) y[C] = tau * sum_q(y[C]);
        // ---- A <- A - v y^T, then right: w = tau A v (row sums), A <- A - w v^T; four rows at a time so that only
        // four row sums are in flight ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            double w[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = 2 * g + h;
                if (2 * t >= R0) {
                    const v2d u = vr[t];
#pragma unroll
                    for (int C = J; C < 4; ++C) {
                        A[2 * t][C] = fma(-u.x, y[C], A[2 * t][C]);
                        A[2 * t + 1][C] = fma(-u.y, y[C], A[2 * t + 1][C]);
                    }
                }
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int C = J; C < 4; ++C) {
                    acc0 = fma(A[2 * t][C], vcol[C], acc0);
                    acc1 = fma(A[2 * t + 1][C], vcol[C], acc1);
                }
                w[2 * h] = acc0;
                w[2 * h + 1] = acc1;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = tau * sum_c(w[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int C = J; C < 4; ++C) A[4 * g + r][C] = fma(-w[r], vcol[C], A[4 * g + r][C]);
        }
        wave_lds_sync();
    }
}

__device__ __forceinline__ void reg_lane_init(RegLane& L, int lane) {
    L.lane = lane;
    L.q = (lane >> 2) & 3;
    L.c = (lane & 3) | ((lane >> 4) << 2);
    L.rho_lane = (lane & 3) * 16 + (lane >> 2);
    L.col_base = (L.c & 3) * 16 + (L.c >> 2);
}

// Load one 64 x 64 matrix into registers, scale it like xGEEV when `scaling` (returns the power-of-two exponent that
// the eigenvalues have to be multiplied back by, or kRegNonFinite), balance it like xGEBAL when `balance`, and reduce
// it to Hessenberg form.  `vecs` = kRegVec doubles of LDS owned by this wave; `scratch` = 64 x 64 doubles of global
// memory owned by this wave (touched only by the rare permutation path of the balancing).
__device__ __forceinline__ int hess_reg_reduce(double (&A)[16][4], const RegLane& L, const double* __restrict__ src,
                                               int lda, double* __restrict__ vecs, double* __restrict__ scratch,
                                               bool scaling, int balance) {
    constexpr int n = 64;
    double* xs = vecs;
    double* ys = xs + 64;
    double* zs = ys + 64;
    // Inside a per-matrix loop the 16 row offsets (4R + q) * lda and the element addresses of the scratch copy are loop
    // invariants; hoisted, they occupy ~150 registers across the whole reduction and are spilled.  Opaque copies keep
    // their computation here (16 multiply-adds per matrix) and in the rare path.
    asm volatile("" : "+s"(lda));
    asm volatile("" : "+s"(scratch));          // wave-uniform (one slot per wave): stays in scalar registers
    double amax = 0.0;
    bool finite = true;
#pragma unroll
    for (int R = 0; R < 16; ++R)
#pragma unroll
        for (int C = 0; C < 4; ++C) A[R][C] = src[(int64_t)(4 * R + L.q) * lda + 16 * C + L.c];
#pragma unroll
    for (int R = 0; R < 16; ++R)
#pragma unroll
        for (int C = 0; C < 4; ++C) {
            amax = fmax(amax, fabs(A[R][C]));
            finite = finite && (fabs(A[R][C]) <= 1.7976931348623157e308);   // false for NaN and Inf
        }
    int e = 0;
    // Scaling by a power of two to max |entry| in [1, 2) (exact; the QR stage multiplies the eigenvalues back).  xGEEV
    // only does this outside [1e-140, 1e140]; here every matrix is normalised so that the QR sweep can call a bulge
    // column negligible by an ABSOLUTE threshold (larfg3_coeffs in eig_real_lds.hip) before its squares underflow.
    if (scaling) {
        amax = wave_max(amax, L.lane);
        if (__ballot(!finite) != 0ULL) e = kRegNonFinite;
        else if (amax > 0.0) e = ilogb(amax);
    }
    {
        // two exact factors: for denormal input (e < -1023) 2^-e itself is out of range
        const int me = (e != kRegNonFinite) ? -e : 0;
        const int m1 = me > 1000 ? 1000 : me;
        const double sc = scalbn(1.0, m1);
#pragma unroll
        for (int R = 0; R < 16; ++R)
#pragma unroll
            for (int C = 0; C < 4; ++C) A[R][C] *= sc;
        if (__builtin_expect(me > 1000, 0)) {              // wave-uniform, denormal input only
            const double sc2 = scalbn(1.0, me - 1000);
#pragma unroll
            for (int R = 0; R < 16; ++R)
#pragma unroll
                for (int C = 0; C < 4; ++C) A[R][C] *= sc2;
        }
    }
    if (e == kRegNonFinite) balance = 0;            // nothing sensible to balance; the QR stage reports failure
    // ---- xGEBAL: permutation of isolated eigenvalues (rare), then power-of-two scaling of the remaining block ----
    if (balance) {
        int k0 = 0, l0 = n - 1;
        bool any_iso;
        {
            double rs[16], cs[4];
            offdiag_norms(A, L, 0, n - 1, rs, cs);
            bool iso = false;
#pragma unroll
            for (int R = 0; R < 16; ++R) iso = iso || (rs[R] == 0.0);
#pragma unroll
            for (int C = 0; C < 4; ++C) iso = iso || (cs[C] == 0.0);
            any_iso = __ballot(iso) != 0ULL;
        }
        if (any_iso) {
            // through this wave's global-memory scratch, row-major.  ONE running pointer, bumped by four rows per step behind
            // an opaque copy: with computed indices the compiler materialises eight 64-bit bases (the store offsets exceed
            // the 12-bit immediate) on top of the 128 registers of A and spills -- in a branch almost never taken, but the
            // spill slots and their reloads were the fused kernel's last scratch instructions
            {
                double* sp = scratch + L.q * n + L.c;
#pragma unroll
                for (int R = 0; R < 16; ++R) {
#pragma unroll
                    for (int C = 0; C < 4; ++C) sp[16 * C] = A[R][C];
                    sp += 4 * n;
                    asm volatile("" : "+v"(sp));
                }
            }
            __threadfence();
            permute_isolated(scratch, L.lane, k0, l0);
            __threadfence();
            {
                volatile const double* vs = scratch + L.q * n + L.c;
#pragma unroll
                for (int R = 0; R < 16; ++R) {
#pragma unroll
                    for (int C = 0; C < 4; ++C) A[R][C] = vs[16 * C];
                    vs += 4 * n;
                    asm volatile("" : "+v"(vs));
                }
            }
        }
        const bool win = L.lane >= k0 && L.lane <= l0;
        for (int pass = 0; pass < 16; ++pass) {
            double rs[16], cs[4];                   // (recomputed in pass 0 too: keeping them live costs spills)
            offdiag_norms(A, L, k0, l0, rs, cs);
            // row / column norms of index i to lane i: row groups write their 16 sums, columns their 4
            if (L.c == 0) {
                v2d* d2 = reinterpret_cast<v2d*>(xs + L.q * 16);
#pragma unroll
                for (int t = 0; t < 8; ++t) d2[t] = (v2d){rs[2 * t], rs[2 * t + 1]};
            }
            if (L.q == 0) {
#pragma unroll
                for (int C = 0; C < 4; ++C) ys[L.col_base + 4 * C] = cs[C];
            }
            wave_lds_sync();
            const double r = xs[L.rho_lane];
            double c = ys[L.rho_lane];
            double f = 1.0;
            if (win && c != 0.0 && r != 0.0 && c < 1e300 && r < 1e300) {
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
            if (__ballot(f != 1.0) == 0ULL) break;
            // A(r, c) <- A(r, c) * f_c / f_r   (1 / f_r is exact: powers of two)
            zs[L.rho_lane] = f;
            wave_lds_sync();
            {
                const v2d* fr = reinterpret_cast<const v2d*>(zs + L.q * 16);
                double fc[4];
#pragma unroll
                for (int C = 0; C < 4; ++C) fc[C] = zs[L.col_base + 4 * C];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const v2d u = fr[t];
                    const double i0 = 1.0 / u.x, i1 = 1.0 / u.y;
#pragma unroll
                    for (int C = 0; C < 4; ++C) {
                        A[2 * t][C] = (A[2 * t][C] * fc[C]) * i0;
                        A[2 * t + 1][C] = (A[2 * t + 1][C] * fc[C]) * i1;
                    }
                }
            }
            wave_lds_sync();
        }
    }

    hess_panel<0>(A, L, xs);
    hess_panel<1>(A, L, xs);
    hess_panel<2>(A, L, xs);
    hess_panel<3>(A, L, xs);

    return e;
}

}  // namespace egs
