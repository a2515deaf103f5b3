// posterior_fused.hip - dispatcher of the register-tile fused kernel over the tile count
// NT = ceil((k+1)/16); the kernels themselves are instantiated in posterior_fused_nt.hip.
#include "posterior_kernels.h"
#include <stdlib.h>

#define TP_MAX_NT 15   // NT = 16 would need 163 KiB of LDS (two 32-row staging buffers of 272 doubles)

#define TP_DECL(NT) hipError_t tp_fused_launch_nt##NT(const tp_kargs_t&, int, hipStream_t, tp_launch_info_t*, int*);
TP_DECL(1) TP_DECL(2) TP_DECL(3) TP_DECL(4) TP_DECL(5) TP_DECL(6) TP_DECL(7) TP_DECL(8)
TP_DECL(9) TP_DECL(10) TP_DECL(11) TP_DECL(12) TP_DECL(13) TP_DECL(14) TP_DECL(15)

int tp_fused_max_assets(void) { return 16 * TP_MAX_NT - 1; }

int tp_pick_wave_kernel(int nt, int choice) {
    if (choice >= 0) return choice;
    // measured on MI355X (tools/sweep_k.py, the kernels in one run): one wave per window wins at every tile count it is
    // built for: +10 % (k = 8) .. +55 % (k = 55), +24 % at k = 100, +22..31 % at 8 tiles, +15 % .. -2 % at 9 tiles (round 2).
    // Ten tiles per side (440 accumulator registers) no longer fit one wave's register file next to the row pipeline:
    // two waves per window up to 12 tiles per side, four up to 15 (posterior_wave2_impl.h, round 3).
    return nt <= 9 ? 1 : 2;
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
// block count L the batch needs.  Elementwise (16 bytes per thread).  ADDITIONS ONLY - round 2 slid the sums (add the
// entering block, subtract the leaving one), and a subtraction carries whatever the leaving block held into every later
// position of the run: one NaN row, an infinite price ratio (clamped to +-DBL_MAX by the front-end: its square overflows)
// or merely a huge finite outlier (1e200 x 0.01 = 1e198 absorbs every normal term for good) spoiled windows that do not
// contain the row.  Now a thread owns one pair of doubles for a group of R = min(L, 16) consecutive positions
// b0 = g .. g + R - 1 and splits every window at the group's end:
//     [b0, b0 + L)  =  [b0, g + R)  u  [g + R, g + L)  u  [g + L, b0 + L)
//                      suffix S[i]     middle M (per group)   prefix P[i]       (b0 = g + i)
// S runs backwards over the group's own blocks, P forwards over the blocks behind the middle: every block is read once
// as part of a suffix and once as part of a prefix (the traffic of the sliding form: 2 slot reads + 1 write per position
// for L <= 16) and Q_L[b0] = (S[i] + M) + P[i] contains the blocks of ITS window and nothing else; its rounding depends on
// the panel, b0 and L alone.
typedef double tp_d2 __attribute__((ext_vector_type(2)));
// abs0: the ABSOLUTE block index of table position 0 (the large-k path builds tables for the blocks of the sub-batch in flight
// only).  Groups are cut in absolute positions - multiples of the group length - so Q_L of a block position is the same
// sequence of additions whichever range a table covers; the first group of a table may start below position 0 (its
// positions and blocks there are skipped: a position p >= 0 only needs blocks >= p).
__global__ void __launch_bounds__(256) tp_window_sums_kernel(const tp_d2* __restrict__ G, tp_d2* __restrict__ Q, int nblk,
                                                             long long slot_pairs, int4 Ls, int n_L, long long abs0) {
    constexpr int R = TP_WINSUM_RUN;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= slot_pairs) return;
    const int li = blockIdx.z;
    const int L = li == 0 ? Ls.x : li == 1 ? Ls.y : li == 2 ? Ls.z : Ls.w;
    if (li >= n_L || L < 1) return;
    const int npos = nblk - L + 1;                          // positions b0 with b0 + L <= nblk
    const int Rr = L < R ? L : R;                           // group length
    const int g = (int)(((long long)blockIdx.y) * Rr - abs0 % Rr);          // table position of the group's first block (>= -Rr + 1)
    if (g >= npos) return;
    const tp_d2* Ge = G + e;
    tp_d2* out = Q + (long long)li * nblk * slot_pairs + e;
    tp_d2 S[R];
    tp_d2 run = tp_d2{0.0, 0.0};
#pragma unroll
    for (int i = R - 1; i >= 0; --i) {                      // suffix sums of the group's own blocks (g + i <= nblk - 1)
        if (i < Rr && g + i >= 0) run += Ge[(long long)(g + i) * slot_pairs];
        S[i] = run;
    }
    tp_d2 M = tp_d2{0.0, 0.0};
    for (int b = g + Rr; b < g + L; ++b) M += Ge[(long long)b * slot_pairs];      // empty for L <= 16
    if (g >= 0) out[(long long)g * slot_pairs] = S[0] + M;
    tp_d2 P = tp_d2{0.0, 0.0};
#pragma unroll
    for (int i = 1; i < R; ++i) {
        if (i < Rr && g + i < npos) {
            P += Ge[(long long)(g + L + i - 1) * slot_pairs];
            if (g + i >= 0) out[(long long)(g + i) * slot_pairs] = (S[i] + M) + P;
        }
    }
}

hipError_t tp_window_sums_launch(const double* G, double* Q, int nblk, size_t slot_doubles, const int* L, int n_L,
                                 hipStream_t stream, long long abs0) {
    if (n_L < 1 || nblk < 1) return hipSuccess;
    const long long slot_pairs = (long long)(slot_doubles / 2);
    int groups = 1;                                         // the largest group count among the tables
    for (int i = 0; i < n_L; ++i) {
        const int Rr = L[i] < TP_WINSUM_RUN ? L[i] : TP_WINSUM_RUN;
        const int npos = nblk - L[i] + 1;
        if (Rr >= 1 && npos >= 1) {
            const long long gcount = (abs0 % Rr + npos + Rr - 1) / Rr;
            if (gcount > groups) groups = (int)gcount;
        }
    }
    const int4 Ls = make_int4(L[0], n_L > 1 ? L[1] : 0, n_L > 2 ? L[2] : 0, n_L > 3 ? L[3] : 0);
    hipLaunchKernelGGL(tp_window_sums_kernel, dim3((unsigned)((slot_pairs + 255) / 256), (unsigned)groups, (unsigned)n_L), dim3(256), 0,
                       stream, (const tp_d2*)G, (tp_d2*)Q, nblk, slot_pairs, Ls, n_L, abs0);
    return hipGetLastError();
}
