// Probe for gfx950: do fp64 VALU operations (v_fma_f64) co-execute with fp64 MFMAs issued by ANOTHER
// wavefront of the same SIMD, or do they share the DP pipeline?  One workgroup of 8 waves on one CU
// (2 waves per SIMD): waves 0-3 run role A, waves 4-7 role B.  Roles: 0 idle, 1 MFMA stream,
// 2 independent v_fma_f64 stream, 3 v_readlane + dependent FMA (the pivot-elimination pattern),
// 4 ds_read stream.  Reports cycles per instruction of each role alone and paired.
// Diagnostic tool only: not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ double role_mfma(int iters) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    return s;
}
__device__ __forceinline__ double role_fma(int iters) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-12;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    return s;
}
__device__ __forceinline__ double role_readlane(int iters) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    double p = 1.0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int lo = __builtin_amdgcn_readlane(__double2loint(p), i), hi = __builtin_amdgcn_readlane(__double2hiint(p), i);
            acc[i] = fma(-__hiloint2double(hi, lo), p, acc[i]);
        }
        p += 1e-9;
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    return s;
}
__device__ __forceinline__ double role_f32(int iters) {
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-9f + i;
    float a = 1.0f + threadIdx.x * 1e-9f, b = 1e-12f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(acc[i], a, b);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    return (double)s;
}
__device__ __forceinline__ double run_role(int role, int iters) {
    switch (role) {
        case 1: return role_mfma(iters);
        case 2: return role_fma(iters);
        case 3: return role_readlane(iters);
        case 4: return role_f32(iters);
        default: return 0.0;
    }
}
__global__ void __launch_bounds__(512) probe(double* out, long long* cyc, int roleA, int roleB, int iters, int prioB) {
    const int wv = threadIdx.x >> 6;
    const int role = wv < 4 ? roleA : roleB;
    if (wv >= 4 && prioB) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const long long c0 = clock64();
    const double s = run_role(role, iters);
    const long long c1 = clock64();
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wv] = c1 - c0;
}
int main() {
    double* out; long long* cyc;
    CK(hipMalloc(&out, 512 * 8)); CK(hipMalloc(&cyc, 8 * 8));
    const char* names[] = {"idle", "mfma_f64", "fma_f64", "readlane+fma_f64", "fma_f32"};
    const int per_iter[] = {0, 8, 8, 24, 8};
    const int iters = 4000;
    for (int prio = 0; prio <= 1; ++prio)
    for (int a = 1; a <= 4; ++a)
        for (int b = 0; b <= 4; ++b) {
            if (prio && !(a == 1 && b >= 2)) continue;   // priority runs: vector roles against the MFMA stream only
            if (prio) printf("[role B at s_setprio 3] ");
            probe<<<1, 512>>>(out, cyc, a, b, iters, prio);   // warm
            probe<<<1, 512>>>(out, cyc, a, b, iters, prio);
            CK(hipDeviceSynchronize());
            long long h[8];
            CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
            printf("A=%-17s B=%-17s  A: %6.1f clk/instr", names[a], names[b], (double)h[0] / (iters * per_iter[a]));
            if (b) printf("   B: %6.1f clk/instr", (double)h[4] / (iters * per_iter[b]));
            printf("\n");
        }
    return 0;
}
