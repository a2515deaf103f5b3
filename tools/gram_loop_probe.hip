// Probe for gfx950: what bounds the inner loop "LDS fragments -> fp64 MFMAs -> barrier" that every Gram kernel
// here is built on?  One workgroup of 4 waves per CU slot, 4 workgroups per CU (16 waves/CU, 4 per SIMD), no
// global memory at all: each k-step reads 1 A + 4 B fragments from LDS (ds_read_b64) and issues 4 MFMAs; 4
// k-steps per "chunk", then (optionally) a workgroup barrier.  Variants: (0) reads then MFMAs per k-step, as
// the kernels do; (1) operands of the NEXT k-step read before the MFMAs of this one (software pipelining);
// (2) = 0 without the barrier; (3) = 1 without the barrier.  Reports the MFMA pipe utilisation.
// Diagnostic tool only: not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
constexpr int LDX = 144, CH = 16;

template <int VARIANT>
__global__ void __launch_bounds__(256) probe(double* out, int chunks) {
    __shared__ double lds[2 * CH * LDX];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fr = lane & 15, fq = lane >> 4;
    for (int i = tid; i < 2 * CH * LDX; i += 256) lds[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    d4 acc[4];
    for (int b = 0; b < 4; ++b) acc[b] = d4{0, 0, 0, 0};
    constexpr bool PIPE = (VARIANT & 1) != 0, BAR = VARIANT < 2;
    for (int ch = 0; ch < chunks; ++ch) {
        const double* lb = lds + (ch & 1) * CH * LDX + fq * LDX + fr;
        if (!PIPE) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double a = lb[4 * s4 * LDX + 16 * wv];
                double bb[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) bb[b] = lb[4 * s4 * LDX + 64 + 16 * b];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb[b], acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            double a = lb[16 * wv], bb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) bb[b] = lb[64 + 16 * b];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                double an = 0, bn[4] = {0, 0, 0, 0};
                if (s4 < 3) {
                    an = lb[4 * (s4 + 1) * LDX + 16 * wv];
#pragma unroll
                    for (int b = 0; b < 4; ++b) bn[b] = lb[4 * (s4 + 1) * LDX + 64 + 16 * b];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb[b], acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a = an;
#pragma unroll
                for (int b = 0; b < 4; ++b) bb[b] = bn[b];
            }
        }
        if (BAR) __syncthreads();
    }
    double s = 0;
    for (int b = 0; b < 4; ++b) s += acc[b][0] + acc[b][1] + acc[b][2] + acc[b][3];
    out[blockIdx.x * 256 + tid] = s;
}

// 8 tiles per wave (32 x 64 strip: 2 A fragments x 4 B fragments, 6 LDS reads per 8 MFMAs), 16 staged rows of
// 192 columns per chunk, 53 KB of LDS -> 3 workgroups per CU
constexpr int LDX8 = 208;
template <bool BAR>
__global__ void __launch_bounds__(256) probe8(double* out, int chunks) {
    __shared__ double lds[2 * CH * LDX8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fr = lane & 15, fq = lane >> 4;
    for (int i = tid; i < 2 * CH * LDX8; i += 256) lds[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    d4 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = d4{0, 0, 0, 0};
    for (int ch = 0; ch < chunks; ++ch) {
        const double* lb = lds + (ch & 1) * CH * LDX8 + fq * LDX8 + fr;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const double a0 = lb[4 * s4 * LDX8 + 32 * wv], a1 = lb[4 * s4 * LDX8 + 32 * wv + 16];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double bb = lb[4 * s4 * LDX8 + 128 + 16 * b];
                acc[0][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bb, acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bb, acc[1][b], 0, 0, 0);
            }
        }
        if (BAR) __syncthreads();
    }
    double s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <bool BAR>
void run8(double* out, int blocks, int chunks, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe8<BAR><<<blocks, 256>>>(out, chunks);
    CK(hipEventRecord(e0));
    probe8<BAR><<<blocks, 256>>>(out, chunks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)blocks * 4 * chunks * 32;
    printf("%-46s %8.3f ms   MFMA pipe %.0f %% (at 2.4 GHz)\n", name, ms, 100 * mfma * 64.0 / (1024.0 * ms * 1e-3 * 2.4e9));
}

template <int V>
void run(double* out, int blocks, int chunks, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<V><<<blocks, 256>>>(out, chunks);
    CK(hipEventRecord(e0));
    probe<V><<<blocks, 256>>>(out, chunks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = (double)blocks * 4 * chunks * 16;                 // MFMAs issued
    const double util = mfma * 64.0 / (1024.0 * ms * 1e-3 * 2.4e9);      // 1024 SIMDs at 2.4 GHz
    printf("%-46s %8.3f ms   MFMA pipe %.0f %% (at 2.4 GHz)\n", name, ms, 100 * util);
}

int main() {
    double* out;
    const int blocks = 256 * 4 * 8, chunks = 200;       // 8 rounds of 4 workgroups per CU
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    run<0>(out, blocks, chunks, "reads then MFMAs per k-step, barrier per chunk");
    run<1>(out, blocks, chunks, "next k-step's reads before the MFMAs, barrier");
    run<2>(out, blocks, chunks, "reads then MFMAs per k-step, no barrier");
    run<3>(out, blocks, chunks, "next k-step's reads before the MFMAs, no barrier");
    run8<true>(out, 256 * 3 * 8, chunks, "8 tiles per wave, barrier per chunk (3 wg/CU)");
    run8<false>(out, 256 * 3 * 8, chunks, "8 tiles per wave, no barrier (3 wg/CU)");
    return 0;
}
