// Lane mapping of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by one-hot operands.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* a, const double* b, double* o) {
    o[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}
int main() {
    double *a, *b, *o;
    (void)hipHostMalloc(&a, 512); (void)hipHostMalloc(&b, 512); (void)hipHostMalloc(&o, 512);
    // for each (la, lb): which output lanes light up
    for (int la = 0; la < 64; ++la) {
        for (int lb = 0; lb < 64; ++lb) {
            for (int i = 0; i < 64; ++i) { a[i] = 0; b[i] = 0; }
            a[la] = 1; b[lb] = 1;
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o);
            (void)hipDeviceSynchronize();
            for (int i = 0; i < 64; ++i) if (o[i] != 0.0) printf("A@%d B@%d -> D@%d\n", la, lb, i);
        }
    }
    // all-ones B: which lanes does output lane l sum over?
    for (int l = 0; l < 64; ++l) { a[l] = (double)(1ull << (l % 16)) + 65536.0 * (l / 16); b[l] = 1.0; }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o);
    (void)hipDeviceSynchronize();
    for (int l = 0; l < 64; ++l) printf("ones: D@%d = %.0f\n", l, o[l]);
    return 0;
}
