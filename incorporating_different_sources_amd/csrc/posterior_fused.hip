// posterior_fused.hip - dispatcher of the register-tile fused kernel over the tile count
// NT = ceil((k+1)/16); the kernels themselves are instantiated in posterior_fused_nt.hip.
#include "posterior_kernels.h"
#include <stdlib.h>

#define TP_MAX_NT 15   // NT = 16 would need 163 KiB of LDS (two 32-row staging buffers of 272 doubles)

#define TP_DECL(NT) hipError_t tp_fused_launch_nt##NT(const tp_kargs_t&, int, hipStream_t, tp_launch_info_t*, int*);
TP_DECL(1) TP_DECL(2) TP_DECL(3) TP_DECL(4) TP_DECL(5) TP_DECL(6) TP_DECL(7) TP_DECL(8)
TP_DECL(9) TP_DECL(10) TP_DECL(11) TP_DECL(12) TP_DECL(13) TP_DECL(14) TP_DECL(15)

int tp_fused_max_assets(void) { return 16 * TP_MAX_NT - 1; }

bool tp_use_wave_kernel(int nt, int choice) {
    if (choice >= 0) return choice != 0;
    // measured on MI355X (tools/sweep_k.py, both kernels in one run, gpurun_out/r03f/sweep.log and the round's later runs):
    // one wave per window wins at every tile count it is built for: +10 % (k = 8) .. +55 % (k = 55), +24 % at k = 100,
    // +22..31 % at 8 tiles (k = 112..127: four of the 36 tiles live in VGPRs), +15 % .. -2 % at 9 tiles (k = 128..143).
    // Ten tiles per side (440 accumulator registers) no longer fit the register file next to the row pipeline.
    return nt >= 1 && nt <= 9;
}

hipError_t tp_fused_launch(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info,
                           int* want_occupancy) {
    const int nt = (a.k + 1 + 15) / 16;
    switch (nt) {
#define TP_CASE(NT) case NT: return tp_fused_launch_nt##NT(a, grid, stream, info, want_occupancy);
        TP_CASE(1) TP_CASE(2) TP_CASE(3) TP_CASE(4) TP_CASE(5) TP_CASE(6) TP_CASE(7) TP_CASE(8)
        TP_CASE(9) TP_CASE(10) TP_CASE(11) TP_CASE(12) TP_CASE(13) TP_CASE(14) TP_CASE(15)
        default: return hipErrorInvalidValue;
    }
}

// Block-window sums of the shared daily Gram: Q_L[b0] = G[b0] + ... + G[b0 + L - 1] for every block position b0 and each
// block count L the batch needs.  Elementwise (16 bytes per thread): a thread owns one pair of doubles for a run of
// TP_WINSUM_RUN consecutive positions - the first sum of the run is taken in full (ascending blocks), the following ones
// slide (add the entering block, subtract the leaving one), so the rounding of Q_L[b0] depends on the panel and on b0
// alone.  HBM / L2-bound: reads 2 slots and writes 1 per position.
// Non-finite panel values (a NaN row; an infinite price ratio that the front-end clamps to +-DBL_MAX, whose square
// overflows) must poison exactly the windows that CONTAIN them: a sliding difference would carry Inf - Inf = NaN into
// every later position of the run.  So a position whose previous sum is not finite is summed in full again - while the
// bad block is inside the window that is the same NaN / Inf, once it has left it is the clean sum.
typedef double tp_d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) tp_window_sums_kernel(const tp_d2* __restrict__ G, tp_d2* __restrict__ Q, int nblk,
                                                             long long slot_pairs, int4 Ls, int n_L) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= slot_pairs) return;
    const int li = blockIdx.z;
    const int L = li == 0 ? Ls.x : li == 1 ? Ls.y : li == 2 ? Ls.z : Ls.w;
    const int npos = nblk - L + 1;                          // positions b0 with b0 + L <= nblk
    const int bs = blockIdx.y * TP_WINSUM_RUN;
    if (li >= n_L || L < 1 || bs >= npos) return;
    const int be = bs + TP_WINSUM_RUN < npos ? bs + TP_WINSUM_RUN : npos;
    auto full = [&](int b0) {
        tp_d2 acc = G[(long long)b0 * slot_pairs + e];
        for (int b = b0 + 1; b < b0 + L; ++b) acc += G[(long long)b * slot_pairs + e];
        return acc;
    };
    tp_d2 sum = full(bs);
    tp_d2* out = Q + (long long)li * nblk * slot_pairs;
    out[(long long)bs * slot_pairs + e] = sum;
    for (int b0 = bs + 1; b0 < be; ++b0) {
        if (__builtin_isfinite(sum[0]) && __builtin_isfinite(sum[1]))
            sum += G[(long long)(b0 + L - 1) * slot_pairs + e] - G[(long long)(b0 - 1) * slot_pairs + e];
        else
            sum = full(b0);
        out[(long long)b0 * slot_pairs + e] = sum;
    }
}

hipError_t tp_window_sums_launch(const double* G, double* Q, int nblk, size_t slot_doubles, const int* L, int n_L,
                                 hipStream_t stream) {
    if (n_L < 1 || nblk < 1) return hipSuccess;
    const long long slot_pairs = (long long)(slot_doubles / 2);
    const int runs = (nblk + TP_WINSUM_RUN - 1) / TP_WINSUM_RUN;
    const int4 Ls = make_int4(L[0], n_L > 1 ? L[1] : 0, n_L > 2 ? L[2] : 0, n_L > 3 ? L[3] : 0);
    hipLaunchKernelGGL(tp_window_sums_kernel, dim3((unsigned)((slot_pairs + 255) / 256), (unsigned)runs, (unsigned)n_L), dim3(256), 0,
                       stream, (const tp_d2*)G, (tp_d2*)Q, nblk, slot_pairs, Ls, n_L);
    return hipGetLastError();
}
