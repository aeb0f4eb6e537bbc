// Measures the relative error of the gfx950 f64 reciprocal / rsqrt seeds and of 0/1/2 Newton refinements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double e = __builtin_fma(-v, r0, 1.0);
    double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-v, r1, 1.0);
    double r2 = __builtin_fma(r1, e, r1);
    double q0 = __builtin_amdgcn_rsq(v);
    double h = 0.5 * v;
    double f = __builtin_fma(-h * q0, q0, 0.5);
    double q1 = __builtin_fma(q0, f, q0);
    f = __builtin_fma(-h * q1, q1, 0.5);
    double q2 = __builtin_fma(q1, f, q1);
    out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2;
    out[6 * i + 3] = q0; out[6 * i + 4] = q1; out[6 * i + 5] = q2;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::exp((u - 0.5) * 40.0);
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 48);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    std::vector<double> o(6 * (size_t)n);
    hipMemcpy(o.data(), dout, n * 48, hipMemcpyDeviceToHost);
    double err[6] = {0};
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / x[i], q = 1.0L / sqrtl((long double)x[i]);
        for (int j = 0; j < 3; ++j) err[j] = std::fmax(err[j], (double)fabsl((o[6 * (size_t)i + j] - t) / t));
        for (int j = 3; j < 6; ++j) err[j] = std::fmax(err[j], (double)fabsl((o[6 * (size_t)i + j] - q) / q));
    }
    printf("rcp  seed %.3e  1 newton %.3e  2 newton %.3e\n", err[0], err[1], err[2]);
    printf("rsq  seed %.3e  1 newton %.3e  2 newton %.3e\n", err[3], err[4], err[5]);
    return 0;
}
