// Probe (gfx950): cycles per row update of the 16-pivot chain's inner loop, one wave per SIMD:
//   A: 2 x v_readlane_b32 + v_fma_f64 with an SGPR-pair multiplier (the kernel's form)
//   B: v_fmac_f64_dpp row_newbcast:i (multiplier broadcast inside each row of 16 lanes)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int I> __device__ __forceinline__ void fmac_dpp(double& acc, double src, double piv) {
    // acc -= bcast(src, lane I of each row) * piv   (src is negated beforehand)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(piv), "n"(I));
}
__global__ void __launch_bounds__(256) probe(double* out, long long* cyc, int mode, int iters) {
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        double piv = a[0] * 0.999;
        if (mode == 0) {
#pragma unroll
            for (int i = 2; i < 16; ++i) { const double s = readlane_d(piv, i); a[i] = fma(-s, piv, a[i]); }
        } else {
            const double np = -piv;
            fmac_dpp<2>(a[2], np, piv); fmac_dpp<3>(a[3], np, piv); fmac_dpp<4>(a[4], np, piv); fmac_dpp<5>(a[5], np, piv);
            fmac_dpp<6>(a[6], np, piv); fmac_dpp<7>(a[7], np, piv); fmac_dpp<8>(a[8], np, piv); fmac_dpp<9>(a[9], np, piv);
            fmac_dpp<10>(a[10], np, piv); fmac_dpp<11>(a[11], np, piv); fmac_dpp<12>(a[12], np, piv); fmac_dpp<13>(a[13], np, piv);
            fmac_dpp<14>(a[14], np, piv); fmac_dpp<15>(a[15], np, piv);
        }
        a[0] = a[15] * 1e-3 + 1.0;
    }
    const long long c1 = clock64();
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = c1 - c0;
}
int main() {
    double* out; long long* cyc;
    CK(hipMalloc(&out, 256 * 8)); CK(hipMalloc(&cyc, 4 * 8));
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 256>>>(out, cyc, mode, 2000);
        probe<<<1, 256>>>(out, cyc, mode, 2000);
        CK(hipDeviceSynchronize());
        long long h[4]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
        double o[4]; CK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %.1f cycles per 14 row updates = %.1f per update   (check %.6f)\n", mode, mode ? "v_fmac_f64_dpp row_newbcast" : "2 v_readlane + v_fma_f64",
               (double)h[0] / 2000, (double)h[0] / 2000 / 14, o[1]);
    }
    return 0;
}
