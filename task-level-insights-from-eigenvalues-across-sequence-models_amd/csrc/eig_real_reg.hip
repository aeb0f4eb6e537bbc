// K5 (real, n = 64), first stage as a kernel of its own: hess_reg_reduce (hess_reg.h, matrix in registers) per
// wavefront, result stored densely.  Used by egs_hessenberg_batched and by the two-kernel A/B path of the solver; the
// eigenvalue driver itself runs the fused kernel of eig_real_lds.hip.
#include <stdlib.h>

#include "hess_reg.h"

namespace egs {

constexpr int kRegWaves = 4;                // matrices (waves) per workgroup

__global__ __launch_bounds__(kRegWaves* kWave, 2) void hess_reg_kernel(const double* __restrict__ a, int64_t nmat,
                                                                        int lda, int64_t stride,
                                                                        double* __restrict__ hout,
                                                                        int32_t* __restrict__ scale_exp, int balance) {
    extern __shared__ __attribute__((aligned(16))) double smem_reg[];
    constexpr int n = 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);       // uniform by construction: a scalar
    const int64_t mat = (int64_t)blockIdx.x * kRegWaves + wave;
    if (mat >= nmat) return;                        // whole wave exits; no workgroup barriers are used
    RegLane L;
    reg_lane_init(L, threadIdx.x & (kWave - 1));
    double* dst = hout + mat * (int64_t)n * n;
    double A[16][4];
    // the output slot of this matrix doubles as the scratch of the permutation path
    // always normalised inside (max |entry| in [1, 2), exact): without an exponent output (egs_hessenberg_batched: the
    // caller wants a matrix similar to its input) the result is scaled back when it is stored -- entries of 1e-220
    // otherwise gave a wrong reduction (their squares underflow), entries of 1e+220 an overflow
    const int e = hess_reg_reduce(A, L, a + mat * stride, lda, smem_reg + (size_t)wave * kRegVec, dst, true, balance);
    if (scale_exp != nullptr && L.lane == 0) scale_exp[mat] = e;
    const int back = (scale_exp == nullptr && e != kRegNonFinite) ? e : 0;
    // the Hessenberg matrix, dense, with exact zeros below the subdiagonal
#pragma unroll
    for (int R = 0; R < 16; ++R)
#pragma unroll
        for (int C = 0; C < 4; ++C) {
            const int row = 4 * R + L.q, col = 16 * C + L.c;
            dst[(int64_t)row * n + col] = (row > col + 1) ? 0.0 : (back ? scalbn(A[R][C], back) : A[R][C]);
        }
}

int hess_reg_launch(const double* a, int64_t nmat, int lda, int64_t stride, double* h, int32_t* scale_exp, int balance,
                    hipStream_t stream) {
    if (nmat == 0) return 0;
    const unsigned grid = (unsigned)((nmat + kRegWaves - 1) / kRegWaves);
    hipLaunchKernelGGL(hess_reg_kernel, dim3(grid), dim3(kRegWaves * kWave), kRegWaves * kRegVec * sizeof(double), stream,
                       a, nmat, lda, stride, h, scale_exp, balance);
    EGS_LAUNCH_CHECK();
    return 0;
}

}  // namespace egs
