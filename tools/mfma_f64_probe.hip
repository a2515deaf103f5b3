// Probe for gfx950: (1) lane maps of v_mfma_f64_16x16x4_f64 checked with exact integer
// data and an ASYMMETRIC B, (2) issue rate of the f64 MFMA and of v_fma_f64, (3) device props.
// Diagnostic tool only: not part of the product path.  Build: see tools/Makefile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// D(16x16) = A(16x4) * B(4x16).  Assumed maps (cdna_hip_programming.md §3):
//   A operand: lane l supplies A[i = l&15][k = l>>4]
//   B operand: lane l supplies B[k = l>>4][j = l&15]
//   D: register r of lane l = D[row = (l>>4) + 4r][col = l&15]
__global__ void layout_kernel(const double* A, const double* B, double* D) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ void mfma_rate_kernel(double* out, long long* cyc, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    long long t0 = wall_clock64();
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64();
    long long t1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}

template <int NACC>
__global__ void fma_rate_kernel(double* out, long long* cyc, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-12;
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    long long c1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}

// dependent chain of rsqrt + divisions, to price the pivot step of a Cholesky
__global__ void sqrt_chain_kernel(double* out, long long* cyc, int iters) {
    double x = 2.0 + threadIdx.x;
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) x = 1.0 / sqrt(x) + 1.5;
    long long c1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s arch=%s CUs=%d clock=%d kHz memclk=%d kHz bus=%d LDS/block=%zu regs/block=%d L2=%d\n",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.memoryBusWidth,
           p.sharedMemPerBlock, p.regsPerBlock, p.l2CacheSize);
    printf("totalGlobalMem=%.1f GiB maxSharedMemoryPerMultiProcessor=%zu\n",
           p.totalGlobalMem / 1073741824.0, p.maxSharedMemoryPerMultiProcessor);

    // ---- layout ----
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (i + 1) * 10 + k;       // asymmetric
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k + 1) * 100 + j * 3 + 1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k)
        R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dD);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
    printf("layout check (A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4r][l&15]): %s (%d mismatches)\n",
           bad ? "FAIL" : "PASS", bad);
    if (bad) {
        // try to identify: print first rows
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) printf("%g/%g ", D[i * 16 + j], R[i * 16 + j]); printf("\n"); }
    }

    // ---- rates ----
    double* dout; long long* dcyc; long long hc[2];
    CK(hipMalloc(&dout, 8 * 1024 * 1024)); CK(hipMalloc(&dcyc, 16));
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN_MFMA(NACC, GRID, BLOCK) do { \
        mfma_rate_kernel<NACC><<<GRID, BLOCK>>>(dout, dcyc, 10); CK(hipDeviceSynchronize()); \
        CK(hipEventRecord(e0)); \
        mfma_rate_kernel<NACC><<<GRID, BLOCK>>>(dout, dcyc, iters); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        CK(hipMemcpy(hc, dcyc, 16, hipMemcpyDeviceToHost)); \
        double n = (double)iters * NACC; \
        double flops = n * 2048.0 * (double)(GRID) * ((BLOCK) / 64); \
        printf("mfma_f64 NACC=%d grid=%d block=%d: %.1f clk/mfma/wave (clock64), %.1f ticks(100MHz)/mfma, %.3f ms, %.2f TFLOP/s\n", \
               NACC, GRID, BLOCK, hc[0] / n, hc[1] / n, ms, flops / ms * 1e-9); } while (0)
    RUN_MFMA(1, 1, 64);
    RUN_MFMA(4, 1, 64);
    RUN_MFMA(8, 1, 64);
    RUN_MFMA(4, 1, 256);
    RUN_MFMA(4, 256, 256);
    RUN_MFMA(4, 512, 256);
    RUN_MFMA(7, 1024, 256);
    RUN_MFMA(4, 2048, 256);
#define RUN_FMA(NACC, GRID, BLOCK) do { \
        fma_rate_kernel<NACC><<<GRID, BLOCK>>>(dout, dcyc, 10); CK(hipDeviceSynchronize()); \
        CK(hipEventRecord(e0)); \
        fma_rate_kernel<NACC><<<GRID, BLOCK>>>(dout, dcyc, iters); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        CK(hipMemcpy(hc, dcyc, 8, hipMemcpyDeviceToHost)); \
        double n = (double)iters * NACC; \
        double flops = n * 128.0 * (double)(GRID) * ((BLOCK) / 64); \
        printf("v_fma_f64 NACC=%d grid=%d block=%d: %.2f clk/fma/wave, %.3f ms, %.2f TFLOP/s\n", \
               NACC, GRID, BLOCK, hc[0] / n, ms, flops / ms * 1e-9); } while (0)
    RUN_FMA(1, 1, 64);
    RUN_FMA(8, 1, 64);
    RUN_FMA(8, 1, 256);
    RUN_FMA(8, 2048, 256);
    RUN_FMA(16, 2048, 512);
    sqrt_chain_kernel<<<1, 64>>>(dout, dcyc, 1000);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("dependent 1/sqrt(x)+c chain: %.1f clk per step\n", hc[0] / 1000.0);
    return bad ? 1 : 0;
}
