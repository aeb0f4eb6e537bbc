// Dependent-chain latencies (one wavefront alone on its SIMD) of the instructions the n = 64 QR sweep is built from,
// in shader cycles (s_memtime), gfx950.  Each test runs a chain of N dependent operations and reports cycles / op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N 256
#define T0() const uint64_t t0 = __builtin_readcyclecounter()
#define T1(slot) do { const uint64_t t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[slot] = (double)(t1 - t0) / N; } while (0)
#define KEEP(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ double bcast(double x, int s) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), s);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), s);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__global__ void lat(double* out, const double* in, int sidx, double* sink) {
    __shared__ double lds_all[8 * 1024];
    const int lane = threadIdx.x & 63;              // every wave runs the same chains on its own 8 KB of LDS
    double* lds = lds_all + (threadIdx.x >> 6) * 1024;
    double a = in[lane], b = in[64 + lane], c = in[128 + lane];
    for (int i = lane; i < 1024; i += 64) lds[i] = in[i & 255];
    __syncthreads();
    double acc = 0.0;
    {   // 0: empty timing
        T0();
        T1(0);
    }
    {   // 1: dependent v_fma_f64
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = __builtin_fma(x, b, c); KEEP(x); }
        T1(1);
        acc += x;
    }
    {   // 2: 4 independent v_fma_f64 chains (cycles per instruction)
        double x0 = a, x1 = b, x2 = c, x3 = a + b;
        T0();
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            x0 = __builtin_fma(x0, b, c); x1 = __builtin_fma(x1, b, c); x2 = __builtin_fma(x2, b, c); x3 = __builtin_fma(x3, b, c);
            KEEP(x0); KEEP(x1); KEEP(x2); KEEP(x3);
        }
        T1(2);
        acc += x0 + x1 + x2 + x3;
    }
    {   // 3: dependent v_rsq_f64
        double x = a + 2.0;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_rsq(x); KEEP(x); }
        T1(3);
        acc += x;
    }
    {   // 4: dependent v_rcp_f64
        double x = a + 2.0;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_rcp(x); KEEP(x); }
        T1(4);
        acc += x;
    }
    {   // 5: independent v_rsq_f64 (issue cost)
        double x0 = a + 2.0, x1 = b + 2.0, x2 = c + 2.0, x3 = a + 3.0;
        T0();
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            x0 = __builtin_amdgcn_rsq(x0); x1 = __builtin_amdgcn_rsq(x1); x2 = __builtin_amdgcn_rsq(x2); x3 = __builtin_amdgcn_rsq(x3);
            KEEP(x0); KEEP(x1); KEEP(x2); KEEP(x3);
        }
        T1(5);
        acc += x0 + x1 + x2 + x3;
    }
    {   // 6: readlane (2 x b32, uniform dynamic index) -> v_fma_f64 with the SGPR pair as operand
        double x = a;
        int s = sidx;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { const double u = bcast(x, s); x = __builtin_fma(u, b, c); KEEP(x); }
        T1(6);
        acc += x;
    }
    {   // 7: LDS store -> load of another lane's value (one round trip per op)
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            lds[lane] = x;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            x = lds[lane ^ 1] + 1.0;
            asm volatile("" ::: "memory");
            KEEP(x);
        }
        T1(7);
        acc += x;
    }
    {   // 8: ds_bpermute_b32 x 2 (one double) chain
        double x = a;
        const int addr = ((lane + 1) & 63) * 4;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x));
            const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
            x = __hiloint2double(hi, lo);
            KEEP(x);
        }
        T1(8);
        acc += x;
    }
    {   // 9: DPP wave_shr:1 (0x138) of a double (2 v_mov_b32_dpp) chain
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = dpp_mov<0x138>(x); KEEP(x); }
        T1(9);
        acc += x;
    }
    {   // 10: DPP row_shr:1 (0x111) of a double chain
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = dpp_mov<0x111>(x); KEEP(x); }
        T1(10);
        acc += x;
    }
    {   // 11: v_mfma_f64_4x4x4 dependent through the accumulator
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, x, 0, 0, 0); KEEP(x); }
        T1(11);
        acc += x;
    }
    {   // 12: v_mfma_f64_4x4x4 dependent through the A operand
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_mfma_f64_4x4x4f64(x, c, b, 0, 0, 0); KEEP(x); }
        T1(12);
        acc += x;
    }
    {   // 13: LDS pointer chase, ds_read_b64 (broadcast address)
        int idx = sidx;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { const double v = lds[idx]; idx = (__double2loint(v) & 1023); asm volatile("" : "+v"(idx)); }
        T1(13);
        acc += idx;
    }
    {   // 14: dependent v_cndmask_b32 pair (double select)
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = (lane == sidx + (i & 3)) ? b : x; KEEP(x); }
        T1(14);
        acc += x;
    }
    {   // 15: dependent v_add_f64
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = x + b; KEEP(x); }
        T1(15);
        acc += x;
    }
    {   // 16: rsq + cubic Newton (fast_rsqrt) chain
        double x = a + 2.0;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double r = __builtin_amdgcn_rsq(x);
            const double e = __builtin_fma(-x * r, r, 1.0);
            const double p = __builtin_fma(0.375, e, 0.5);
            x = __builtin_fma(r * e, p, r) + 2.0;
            KEEP(x);
        }
        T1(16);
        acc += x;
    }
    {   // 17: LDS store (b64) + load of own value, no cross-lane (store->load forwarding cost)
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            lds[lane + 64] = x;
            asm volatile("" ::: "memory");
            x = lds[lane + 64] + 1.0;
            asm volatile("" ::: "memory");
            KEEP(x);
        }
        T1(17);
        acc += x;
    }
    {   // 18: v_readfirstlane pair -> fma
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
            const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
            x = __builtin_fma(__hiloint2double(hi, lo), b, c);
            KEEP(x);
        }
        T1(18);
        acc += x;
    }
    {   // 19: 8 independent fma (issue only)
        double x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = a + j;
        T0();
#pragma unroll
        for (int i = 0; i < N / 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { x[j] = __builtin_fma(x[j], b, c); }
#pragma unroll
            for (int j = 0; j < 8; ++j) KEEP(x[j]);
        }
        T1(19);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += x[j];
    }
    {   // 20: ds_swizzle? skipped.  DPP row_bcast15 (0x142) chain
        double x = a;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = dpp_mov<0x142>(x); KEEP(x); }
        T1(20);
        acc += x;
    }
    {   // 21: v_mov_b32 x 2 chain (32-bit VALU dependent latency x 2)
        int x = lane;
        T0();
#pragma unroll
        for (int i = 0; i < N; ++i) { x = x + sidx; asm volatile("" : "+v"(x)); }
        T1(21);
        acc += x;
    }
    sink[threadIdx.x] = acc;   // 512 threads at most, 4096 slots
}

int main() {
    double *out, *in, *sink;
    (void)hipHostMalloc(&out, 64 * 8);
    (void)hipHostMalloc(&in, 256 * 8);
    (void)hipMalloc(&sink, 4096 * 8);
    for (int i = 0; i < 256; ++i) in[i] = 1.0 + 1e-3 * (i % 17);
    for (int i = 0; i < 64; ++i) out[i] = -1;
    const char* names[] = {"empty", "dep v_fma_f64", "4 indep v_fma_f64 (per instr)", "dep v_rsq_f64", "dep v_rcp_f64",
                           "4 indep v_rsq_f64 (per instr)", "readlane x2 -> fma", "LDS store->other-lane load (+add)",
                           "bpermute x2 (double)", "dpp wave_shr:1 x2", "dpp row_shr:1 x2", "mfma f64 4x4x4 dep C",
                           "mfma f64 4x4x4 dep A", "LDS pointer chase b64", "cndmask x2 dep", "dep v_add_f64",
                           "fast_rsqrt chain (+add)", "LDS store->own load (+add)", "readfirstlane x2 -> fma",
                           "8 indep fma (per instr)", "dpp row_bcast15 x2", "dep v_add_u32"};
    for (int waves = 1; waves <= 2; ++waves) {
        // waves = 2: 8 waves in the workgroup -> two per SIMD, both running the same chains
        hipLaunchKernelGGL(lat, dim3(1), dim3(waves == 1 ? 64 : 512), 0, 0, out, in, 3, sink);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(lat, dim3(1), dim3(waves == 1 ? 64 : 512), 0, 0, out, in, 3, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        printf("--- %d wave(s) per SIMD ---\n", waves);
        for (int i = 0; i < 22; ++i) printf("%2d %-40s %8.2f cycles/op\n", i, names[i], out[i]);
    }
    return 0;
}
