// Probe for gfx950: does fp64 vector work HIDE under fp64 MFMAs of the SAME wave when the two are interleaved in program
// order?  One wave per SIMD (64-thread workgroups, 512 registers), per iteration NM = 28 MFMAs on independent accumulators
// and NV = 42 v_fma_f64 (independent, or one dependent chain).  Arrangements:
//   0: the vector block, then the MFMA block (what hipcc emits for the Gram loops of posterior_wave_impl.h);
//   1: one or two vector instructions behind every MFMA;
//   2: MFMAs only;  3: vector instructions only.
// Prints shader cycles per iteration (s_memtime) for each.  Diagnostic tool only: not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

#define MFMA(c, a, b) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define FMA(x, y, z) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z))

template <int ARR, bool DEP>
__global__ void __launch_bounds__(64, 1) probe(double* out, long long* cyc, int iters) {
    const int lane = threadIdx.x;
    d4 acc[28];
    for (int t = 0; t < 28; ++t) acc[t] = d4{0, 0, 0, 0};
    double v[14], a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
    for (int i = 0; i < 14; ++i) v[i] = 1.0 + 1e-6 * (i + lane);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (ARR == 0) {
#pragma unroll
            for (int q = 0; q < 42; ++q) { if (DEP) FMA(v[0], a, b); else FMA(v[q % 14], a, b); }
#pragma unroll
            for (int t = 0; t < 28; ++t) MFMA(acc[t], a, b);
        } else if (ARR == 1) {
#pragma unroll
            for (int t = 0; t < 28; ++t) {
                MFMA(acc[t], a, b);
                if (DEP) { FMA(v[0], a, b); if (t & 1) FMA(v[0], a, b); }
                else { FMA(v[(3 * t) % 14], a, b); if (t & 1) FMA(v[(3 * t + 1) % 14], a, b); }
            }
        } else if (ARR == 2) {
#pragma unroll
            for (int t = 0; t < 28; ++t) MFMA(acc[t], a, b);
        } else {
#pragma unroll
            for (int q = 0; q < 42; ++q) { if (DEP) FMA(v[0], a, b); else FMA(v[q % 14], a, b); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7");
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int t = 0; t < 28; ++t) s += acc[t][0] + acc[t][3];
    for (int i = 0; i < 14; ++i) s += v[i];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ARR, bool DEP>
void run(const char* name, double* out, long long* cyc, long long* h, int blocks, int iters) {
    hipLaunchKernelGGL((probe<ARR, DEP>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<ARR, DEP>), dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
    printf("%-58s %8.1f cycles / iteration (28 MFMAs = 1792 at 64 each)   kernel %.3f ms\n", name, mean / iters, ms);
}

int main() {
    const int blocks = 1024, iters = 2000;
    double* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(double) * blocks * 64)); CK(hipMalloc(&cyc, sizeof(long long) * blocks));
    long long* h = (long long*)malloc(sizeof(long long) * blocks);
    run<2, false>("28 MFMAs only", out, cyc, h, blocks, iters);
    run<3, false>("42 independent v_fma_f64 only", out, cyc, h, blocks, iters);
    run<3, true>("42 dependent v_fma_f64 only", out, cyc, h, blocks, iters);
    run<0, false>("42 independent fma, THEN 28 MFMAs", out, cyc, h, blocks, iters);
    run<1, false>("28 MFMAs with 42 independent fma INTERLEAVED", out, cyc, h, blocks, iters);
    run<0, true>("42 dependent fma, THEN 28 MFMAs", out, cyc, h, blocks, iters);
    run<1, true>("28 MFMAs with 42 dependent fma INTERLEAVED", out, cyc, h, blocks, iters);
    return 0;
}
