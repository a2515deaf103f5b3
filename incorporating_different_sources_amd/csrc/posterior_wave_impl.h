#pragma once
// posterior_wave_impl.h - ONE wavefront per rolling window (register-tile path, small tile counts).
//
// The multi-wave kernel of posterior_fused_impl.h spreads the tiles of a window over 4 waves: every staged row
// goes global -> registers -> LDS -> MFMA operand, every phase ends at a workgroup barrier, and during the
// serial phases (the 16-pivot elimination of a diagonal tile, the back substitution) three of the four waves
// of a window are parked (61 % of all wave cycles, profiles/r02_final_pmc_stall_fused_k100.txt).
//
// Here a window is ONE wave that owns the whole upper triangle of the bordered matrix (NT (NT+1)/2 accumulator
// tiles = 224 registers at NT = 7, in the 512-entry unified VGPR/AGPR file at one wave per SIMD):
//   * the MFMA operands of a 4-row k-step are loaded straight from the panels: lane (fq, fr) reads row 4s+fq,
//     column 16i+fr - one 8-byte load per 16-column group IS the A/B operand of every tile in that tile row /
//     column.  No LDS staging, no barrier, three k-steps of loads in flight;
//   * register r of an accumulator tile holds rows 4r..4r+3 of the tile in exactly the MFMA operand layout, so
//     the block row R_jJ = M A_jJ and the trailing update A_IJ -= R_jI' R_jJ of the blocked Cholesky take their
//     operands from the accumulators themselves - the factorisation touches LDS only for the 16x16 diagonal
//     tile (to turn it into one column per lane for the pivot chain) and for M = R_jj^-T;
//   * the back substitution runs along block ROWS: lane-local products over the tiles of the row, one 16-lane
//     DPP reduction per register, and w_I = M' z by MFMA with z as the A operand - no partial sums through LDS;
//   * no __syncthreads anywhere.
// Same arithmetic as the multi-wave kernel per matrix element (same k-step order, same elimination), ref:LINE
// cites /root/reference/src/portfolio_calculations.py as there.
#include "posterior_fused_impl.h"

namespace {

template <int NT_, bool LEAN_ = true>
struct WCfg {
    static constexpr int NT = NT_;
    static constexpr int KP = 16 * NT;
    static constexpr int NTILES = NT * (NT + 1) / 2;
    static constexpr int MLD = 17;                               // row stride of an M block (conflict-free both ways)
    static constexpr int OFF_M = 0;                              // [NT][16][MLD]  M_j = R_jj^-T, row-major
    static constexpr int OFF_DG = OFF_M + NT * 16 * MLD;         // [16][16] diagonal tile handed to the pivot chain
    static constexpr int OFF_IDT = OFF_DG + 256;                 // [16][16] identity
    static constexpr int OFF_VEC = OFF_IDT + 256;                // [KP] column sums / Jeffreys t / y
    static constexpr int LDS_DOUBLES = OFF_VEC + KP;
    static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
    // General (index) layout only, behind the fixed part: the panel rows of the pass in flight - and the daily pass's
    // per-row risk-free adjustments - staged in LDS once per pass: [rows] doubles, then [rows] ints, rows = wave_idx_rows().
    // Read from global memory inside the row loop they were a DEPENDENT load per k-step, and since s_waitcnt vmcnt counts
    // in order, waiting for the youngest load (the index) drained the whole three-k-step pipeline of row loads every step
    // (round 3: 30 % of the wave cycles of the index layout parked in s_waitcnt, 1.26 vs 1.04 ms at configs[1]).  LDS reads
    // are counted by lgkmcnt and are fetched one k-step ahead.  Sized per batch (dynamic LDS); batches whose passes would
    // not fit WAVE_LDS_LIMIT stay on the multi-wave kernel (wave_idx_fits).
    static constexpr int OFF_SUB = LDS_DOUBLES;
};
constexpr int WAVE_LDS_LIMIT = 64 * 1024;
// rows of the longest pass of a batch, rounded up to whole k-steps of 64 lanes' staging loads
__host__ __device__ inline int wave_idx_rows(int n_r, int m, bool conj) {
    const int r = (conj && m > n_r) ? m : n_r;
    return (r + 63) & ~63;
}
__host__ __device__ inline int wave_idx_bytes(int n_r, int m, bool conj) { return wave_idx_rows(n_r, m, conj) * 12; }

// row-major index of upper-triangle tile (I, J), I <= J - the numbering of the shared Gram tables
constexpr int wtile(int NT, int I, int J) { return I * NT - I * (I - 1) / 2 + (J - I); }

// The MFMAs of the Gram loops are inline assembly, which the compiler's hazard recogniser does not see: a
// v_mfma_f64_16x16x4 result may be read by anything but the SrcC of the next MFMA on the same registers only 19
// wait states after issue (what hipcc inserts behind the builtin).  Every pass over rows ends here; the asm
// "modifies" every tile, so no later use of an accumulator can be scheduled in front of the wait.
// Where a tile lives.  The first 32 tiles fill the AGPR half of the register file (256 registers); tile counts of 8 and
// 9 per side (36 / 45 tiles) keep the rest in VGPRs - MFMA accumulators may be either.
constexpr bool wave_tile_in_agpr(int t) { return t < 32; }

// s_nop 1: a VGPR written by a vector instruction may be read as an MFMA operand two wait states later at the earliest
// (hipcc puts the same s_nop in front of the builtin); the asm carries it because the compiler cannot see the hazard.
template <int T>
__device__ __forceinline__ void wave_mfma_agpr(d4& c, double a, double b) {
    if constexpr (wave_tile_in_agpr(T)) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void wave_settle14(d4& c0, d4& c1, d4& c2, d4& c3, d4& c4, d4& c5, d4& c6, d4& c7, d4& c8, d4& c9,
                                              d4& c10, d4& c11, d4& c12, d4& c13) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7), "+a"(c8),
                 "+a"(c9), "+a"(c10), "+a"(c11), "+a"(c12), "+a"(c13));
}
template <int T>
__device__ __forceinline__ void wave_settle1(d4& c0) {
    if constexpr (wave_tile_in_agpr(T)) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(c0));
    else asm volatile("s_nop 15\n\ts_nop 7" : "+v"(c0));
}
// Entering a row loop: hand the loop-carried accumulators over as AGPR values (an empty asm that "modifies" them), so
// that the loop's phi nodes become AGPR phis; otherwise the accumulators live in VGPRs across the back edge and are
// copied into AGPRs for the asm MFMAs on every iteration.
template <int T>
__device__ __forceinline__ void wave_pin1(d4& c0) {
    if constexpr (wave_tile_in_agpr(T)) asm volatile("" : "+a"(c0));
    else asm volatile("" : "+v"(c0));
}
template <int NT>
__device__ __forceinline__ void wave_pin(d4 (&acc)[WCfg<NT>::NTILES]) {
    static_for<0, WCfg<NT>::NTILES>([&](auto tc) __attribute__((always_inline)) { wave_pin1<decltype(tc)::value>(acc[decltype(tc)::value]); });
}
template <int NT>
__device__ __forceinline__ void wave_settle(d4 (&acc)[WCfg<NT>::NTILES]) {
    constexpr int NTL = WCfg<NT>::NTILES;
    static_for<0, (NTL + 13) / 14>([&](auto gc) __attribute__((always_inline)) {
        constexpr int b = 14 * decltype(gc)::value;
        if constexpr (b + 14 <= NTL && wave_tile_in_agpr(b + 13)) {
            wave_settle14(acc[b], acc[b + 1], acc[b + 2], acc[b + 3], acc[b + 4], acc[b + 5], acc[b + 6], acc[b + 7], acc[b + 8],
                          acc[b + 9], acc[b + 10], acc[b + 11], acc[b + 12], acc[b + 13]);
        } else {
            static_for<b, (b + 14 < NTL ? b + 14 : NTL)>([&](auto tc) __attribute__((always_inline)) {
                wave_settle1<decltype(tc)::value>(acc[decltype(tc)::value]);
            });
        }
    });
}

struct WRows {
    const double* base;     // panel
    long long ld;           // leading dimension (doubles)
    const int* ridx;        // explicit rows of this window, or nullptr
    long long first;        // first row (contiguous)
    const double* sub_row;  // per-row subtrahend (rf_adj) or nullptr
    int count;              // rows
    int count0;             // contiguous, two row ranges: staged rows r >= count0 are panel rows first + r + jump
    int jump;
    bool off32 = false;     // general layout: row * 8 ld + 8 column fits 32 bits for every row of the PANEL (host-checked)
};

// One pass over the rows of a window: acc(I,J) += rows[:, I]' rows[:, J], 4 rows per k-step.
//  HF:   intraday rows, shifted by `shift` (the window's first row, or its column means), column k carries
//        u_r = (y_r - shift).w0 and - when `ones` - column k+1 carries ones (one-pass centring, phase C)
//  !HF:  daily rows minus the per-row risk-free adjustment, column k carries ones (t = X'1, ref:222)
//  lazy_mask (HF): `shift` and `w0v` arrive as RAW loads (padding columns hold column k-1's values); they are zeroed
//  beyond column k here, AFTER the first two k-steps' row loads have been issued - masking them at the call site made the
//  wave wait for the shift row before it could ask for the first panel rows (two memory round trips in a row, and this
//  wave has nothing else to run meanwhile)
//  !LEAN: lds_rows[r] / lds_sub[r] hold the panel row and the subtrahend of row r of this pass (staged by wave_stage_rows)
template <int NT, bool HF, bool LEAN>
__device__ __forceinline__ void wave_gram(const WRows& src, const long long (&coff)[NT], int k, int lane,
                                          double (&shift)[NT], double (&w0v)[NT], bool ones, bool lazy_mask,
                                          d4 (&acc)[WCfg<NT>::NTILES], const int* lds_rows, const double* lds_sub,
                                          double (&csum)[NT], double& usum) {
    constexpr int kI = NT - 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int kc = k - 16 * kI;
    const bool cvl = fr < kc;
    const double border = HF ? ((ones && fr == kc + 1) ? 1.0 : 0.0) : ((fr == kc) ? 1.0 : 0.0);
    const int nks = (src.count + 3) >> 2;
    const bool has_sub = !HF && src.sub_row != nullptr;

    // column offsets (doubles) of this lane in the NT column groups; padding columns re-read column k-1
    // general layout: the pass's panel rows (and subtrahends) come from LDS; the values of the NEXT load are fetched right
    // behind the current load's requests, a whole k-step of MFMAs ahead of their use
    int row_pref = 0;
    double sub_pref = 0.0;
    auto prefetch = [&](int ks) __attribute__((always_inline)) {
        int r = 4 * ks + fq;
        r = r < src.count ? r : src.count - 1;
        row_pref = lds_rows[r];
        if (has_sub) sub_pref = lds_sub[r];
    };
    if constexpr (!LEAN) prefetch(0);
    // 32-bit addressing: a wave-uniform 64-bit base (the window's first row; general layout: the panel) + row * (8 ld) +
    // 8 column, ONE 24-bit multiply-add per operand (the 64-bit row * ld product cost 15 vector instructions per k-step).
    // The contiguous layout always qualifies (tp_layout_is_lean); the general layout when the panel is below 4 GiB
    // (tp_kargs_t::panel_off32 / hf_off32 bit 0), else it keeps 64-bit addresses.
    const char* ub = (const char*)(LEAN ? src.base + src.first * src.ld : src.base);
    const unsigned ld8 = (unsigned)src.ld * 8u;
    unsigned c8[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) c8[i] = 8u * (unsigned)coff[i];
    auto load = [&](double (&v)[NT], double& sub, int ks) __attribute__((always_inline)) {
        int r = 4 * ks + fq;
        r = r < src.count ? r : src.count - 1;                      // rows past the end re-read the last row (masked below)
        if (LEAN || src.off32) {
            unsigned row;
            if constexpr (LEAN) row = (unsigned)(r + (r >= src.count0 ? src.jump : 0));
            else row = (unsigned)row_pref;
            if constexpr (LEAN) {
                // ungathered columns: group i sits 128 i bytes behind group 0 (immediate offsets); only the last group clamps
                const unsigned r0 = __umul24(row, ld8) + c8[0];
#pragma unroll
                for (int i = 0; i < NT - 1; ++i) v[i] = *(const double*)(ub + (size_t)r0 + 128 * i);
                v[NT - 1] = *(const double*)(ub + (size_t)(__umul24(row, ld8) + c8[NT - 1]));
            } else {
                const unsigned ro = __umul24(row, ld8);
#pragma unroll
                for (int i = 0; i < NT; ++i) v[i] = *(const double*)(ub + (size_t)(ro + c8[i]));
            }
        } else {
            const double* p = src.base + (long long)row_pref * src.ld;
#pragma unroll
            for (int i = 0; i < NT; ++i) v[i] = p[coff[i]];
        }
        sub = 0.0;
        if constexpr (!LEAN) {
            if (has_sub) sub = sub_pref;
            prefetch(ks + 1);                                       // loads are issued for ks = 0, 1, 2, ... in this order
        } else {
            if (has_sub) sub = src.sub_row[r];                       // contiguous rows with a risk-free adjustment (rare)
        }
    };
    // MASK: the k-step may hold rows past the end (only the last three k-steps of a pass are built with it)
    auto step = [&](double (&v)[NT], double sub, int ks, auto maskc) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(maskc)::value != 0;
        if (HF) {
#pragma unroll
            for (int i = 0; i < NT; ++i) v[i] -= shift[i];
        } else if (has_sub) {
#pragma unroll
            for (int i = 0; i < NT; ++i) v[i] -= sub;                // ref:57
        }
        v[kI] = cvl ? v[kI] : border;
        if (MASK) {
            const bool rv = 4 * ks + fq < src.count;
#pragma unroll
            for (int i = 0; i < NT; ++i) v[i] = rv ? v[i] : 0.0;
        }
        if (HF) {
            double z = 0.0;
#pragma unroll
            for (int i = 0; i < NT; ++i) z = fma(v[i], w0v[i], z);
            z = rowgroup_sum16(z);
            if (!ones) {
                // k + 1 = 0 (mod 16): no spare column for the ones of the one-pass centring - the column sums of the
                // shifted rows (and the sum of u) are kept by vector adds instead: NT + 1 per k-step next to the MFMAs
                // (round 2 ran a separate pass for the column MEANS at these sizes: a second dependent trip to memory)
#pragma unroll
                for (int i = 0; i < NT; ++i) csum[i] += v[i];
                usum += z;
            }
            if (fr == kc) v[kI] = z;                                 // u_r = (y_r - shift).w0
        }
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int J = decltype(Jc)::value;
                constexpr int t = wtile(NT, I, J);
                // accumulators pinned to the AGPR half of the register file: left to itself the allocator moved the
                // whole accumulator set between VGPRs and AGPRs inside this loop (497 v_accvgpr moves per 84 MFMAs)
                wave_mfma_agpr<t>(acc[t], v[I], v[J]);
            });
        });
    };

    if (nks <= 0) return;
    // Three register sets rotate through load -> (two k-steps of MFMAs) -> use.  The main loop runs whole triples of
    // full k-steps as ONE basic block (loads past the end re-read the last row); at most three k-steps remain.
    double va[NT], vb[NT], vc[NT];
    double sa = 0.0, sb = 0.0, sc = 0.0;
    load(va, sa, 0);
    load(vb, sb, 1);
    if (HF && lazy_mask) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const bool cv = 16 * i + fr < k;
            shift[i] = cv ? shift[i] : 0.0;
            w0v[i] = cv ? w0v[i] : 0.0;
        }
    }
    wave_pin<NT>(acc);
    int ks = 0;
#pragma nounroll
    for (; 4 * (ks + 3) <= src.count; ks += 3) {
        load(vc, sc, ks + 2);
        step(va, sa, ks, ic<0>{});
        load(va, sa, ks + 3);
        step(vb, sb, ks + 1, ic<0>{});
        load(vb, sb, ks + 4);
        step(vc, sc, ks + 2, ic<0>{});
    }
    if (ks < nks) {
        if (ks + 2 < nks) load(vc, sc, ks + 2);
        step(va, sa, ks, ic<1>{});
        if (ks + 1 < nks) step(vb, sb, ks + 1, ic<1>{});
        if (ks + 2 < nks) step(vc, sc, ks + 2, ic<1>{});
    }
    wave_settle<NT>(acc);
}

// General layout: panel row (and subtrahend) of every row of the pass into LDS, 64 rows per instruction.
__device__ __forceinline__ void wave_stage_rows(const WRows& src, int lane, int* li, double* lsb) {
    for (int i = lane; i < src.count; i += 64) {
        li[i] = src.ridx ? src.ridx[i] : (int)(src.first + i);
        if (src.sub_row) lsb[i] = src.sub_row[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// MODE 0: conjugate, 1: Jeffreys - the plain product paths, compiled without the read-back / custom right-hand side /
// shift branches: every branch that merges two versions of the accumulators costs register copies or spills here (one
// kernel with all of them decided at run time, MODE 2, needs 2.2 KB of scratch per lane against 44 bytes, and spill code
// next to the inline-assembly MFMAs is exactly what the hazard argument of wave_settle cannot cover).  MODE 2 is therefore
// not instantiated: batches that ask for a matrix read-back, a custom right-hand side, tp_batch_keep_rhs or a shift run
// on the multi-wave kernel (wave_mode / launch_one).
template <int NT, bool LEAN, int MODE>
__device__ __forceinline__ void wave_window_body(const tp_kargs_t& A, double* lds) {
    constexpr bool FULL = MODE == 2;
    using C = WCfg<NT, LEAN>;
    const int lane = threadIdx.x;
    const int fr = lane & 15, fq = lane >> 4;
    const int k = A.k;
    __builtin_assume(k >= 16 * (NT - 1));
    __builtin_assume(k <= 16 * NT - 1);
    constexpr int kI = NT - 1;
    const int kc = k - 16 * kI;
    const int NTB = (kc == 0) ? NT - 1 : NT;
    const bool colv = fr < kc;
    // XCD-aware workgroup -> window map (see posterior_fused_impl.h): one contiguous window range per XCD
    const long long per_xcd = (A.w_count + 7) >> 3;
    const long long wl = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= per_xcd || wl >= A.w_count) return;
    const long long w = A.w_first + wl;
    const int tid0 = lane;
    (void)tid0;

    const int* cols = (!LEAN && A.col_idx) ? A.col_idx + w * k : nullptr;
    // column offsets (doubles) of this lane in the NT column groups, once per window; padding columns re-read column k-1.
    // The gathered columns' indices are requested together (one wait), not one dependent load per select.
    long long coff[NT];
    if (cols != nullptr) {
        int cidx[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int c = 16 * i + fr;
            cidx[i] = cols[c < k ? c : k - 1];
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) coff[i] = (long long)cidx[i];
    } else {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int c = 16 * i + fr;
            coff[i] = (long long)(c < k ? c : k - 1);
        }
    }
    // general layout: staging region of the pass in flight behind the fixed part of the LDS image
    double* idx_sub_lds = LEAN ? nullptr : lds + C::OFF_SUB;
    int* idx_rows_lds = LEAN ? nullptr : (int*)(lds + C::OFF_SUB + wave_idx_rows(A.n_r, A.m, A.strategy == 0));
    d4 acc[C::NTILES];
    static_for<0, C::NTILES>([&](auto tc) __attribute__((always_inline)) { acc[decltype(tc)::value] = d4{0.0, 0.0, 0.0, 0.0}; });
    // AGPR values from the first definition on: where two paths of the kernel meet, the accumulators must arrive as AGPR
    // values on both, or the merge keeps all of them in VGPRs (and spills)
    wave_pin<NT>(acc);

    // identity tile for the pivot chain (read after many waits on this wave's own LDS traffic)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        lds[C::OFF_IDT + e] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
    }

    double n0 = 0.0, cc = 0.0, q0 = 0.0;
    const bool conj = MODE == 0 ? true : MODE == 1 ? false : (A.strategy == 0);
    const int dbg = (FULL && A.dbg_S1 != nullptr && w == A.dbg_w) ? A.dbg_mode : 0;

    auto dump_matrix = [&]() __attribute__((always_inline)) {
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int J = decltype(Jc)::value;
                constexpr int t = wtile(NT, I, J);
                const int gj = 16 * J + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = 16 * I + fq + 4 * r;
                    if (gi < k && gj < k) {
                        A.dbg_S1[(long long)gi * k + gj] = acc[t][r];
                        A.dbg_S1[(long long)gj * k + gi] = acc[t][r];
                    }
                    if (gi < k && gj == k) A.dbg_S1[(long long)k * k + gi] = acc[t][r];
                }
            });
        });
    };
    // value of element (row kc, column kc) of tile (kI, kI): the corner of the bordered matrix.  All four registers are
    // read and the SCALAR results selected: a run-time choice among the four vector registers came out of hipcc as a
    // branch ladder that left the value undefined for kc >= 12 in some instantiations (k = 31, 63: garbage q0).
    auto corner = [&]() __attribute__((always_inline)) {
        constexpr int t = wtile(NT, kI, kI);
        const int ln = __builtin_amdgcn_readfirstlane(16 * (kc & 3) + kc);
        const double x0 = readlane_d(acc[t][0], ln), x1 = readlane_d(acc[t][1], ln);
        const double x2 = readlane_d(acc[t][2], ln), x3 = readlane_d(acc[t][3], ln);
        const int rr = kc >> 2;
        return rr == 0 ? x0 : rr == 1 ? x1 : rr == 2 ? x2 : x3;
    };

    TP_MARK(0);
    if (conj && dbg != 2) {
        n0 = A.n0[w];
        WRows hs;
        hs.base = A.hf_panel; hs.ld = A.hf_ld;
        hs.ridx = (!LEAN && A.hf_row_idx) ? A.hf_row_idx + w * (long long)A.m : nullptr;
        hs.first = A.hf_start ? A.hf_start[w] : 0;
        hs.sub_row = nullptr;
        hs.count = A.hf_count ? A.hf_count[w] : A.m;
        hs.count0 = 0x7fffffff; hs.jump = 0;
        hs.off32 = (A.hf_off32 & 1) != 0;
        // ---- phase A: the shift row of the one-pass centred scatter (see posterior_fused_impl.h) = the window's first row
        const bool ones = kc < 15;        // a spare column k+1 carries ones; otherwise the sums are kept by vector adds (wave_gram)
        double shift[NT], w0v[NT], csum[NT];
        double usum = 0.0;
        {
            const long long row0 = hs.ridx ? (long long)hs.ridx[0] : hs.first;
            const double* p0 = hs.base + row0 * (long long)hs.ld;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int c = 16 * i + fr;
                const int cl = c < k ? c : k - 1;
                const double sv = p0[coff[i]];
                const double wv = A.w0[w * k + cl];
                shift[i] = sv;               // raw: zeroed beyond column k inside wave_gram (lazy_mask)
                w0v[i] = wv;
                csum[i] = 0.0;
            }
        }
        TP_MARK(1);
        // ---- phase B: Gram of the shifted intraday rows
        // the shift row is the window's FIRST row: shifted it is exactly zero and adds nothing to any sum, so the pass
        // starts at row 1 (77 intraday rows: 19 k-steps instead of 20); the means of phase C still divide by all rows
        const int hf_rows_all = hs.count;
        if (!hs.ridx) { hs.first += 1; hs.count -= 1; }
        else { hs.ridx += 1; hs.count -= 1; }
        if constexpr (!LEAN) wave_stage_rows(hs, lane, idx_rows_lds, idx_sub_lds);      // (WCfg::OFF_SUB)
        wave_gram<NT, true, LEAN>(hs, coff, k, lane, shift, w0v, ones, true, acc, idx_rows_lds, idx_sub_lds, csum, usum);
        hs.count = hf_rows_all;
        TP_MARK(2);
        // ---- phase C: rank-one term of the centring (one-pass form); q0, c, scaling (ref:333, 415-418).  ONE pass over
        // the tiles, a tile row at a time: vector instructions cannot read AGPRs, every tile visits the VGPR half on its
        // way - all 224 registers at once would spill, and a spill reload costs this lone wave a full memory round trip.
        const double invm = 1.0 / (double)hs.count;
        double tj[NT];
        double cz = corner();                                       // z'z = w0'C w0 (before the rank-one term)
        if (ones) {
            // column k+1 holds t_i = sum_r (y_r - s)_i for the asset columns and sum_r u_r in row k
            static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
                constexpr int I = decltype(Ic)::value;
                constexpr int t = wtile(NT, I, kI);
                if (fr == kc + 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        lds[C::OFF_VEC + 16 * I + fq + 4 * r] = (I < kI || fq + 4 * r <= kc) ? acc[t][r] : 0.0;
                }
            });
        } else {
            // the same vector from the vector-add sums: the four row groups of a lane column meet by shuffles
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                double sc_ = csum[i];
                sc_ += __shfl_xor(sc_, 16, 64);
                sc_ += __shfl_xor(sc_, 32, 64);
                if (fq == 0) lds[C::OFF_VEC + 16 * i + fr] = (16 * i + fr < k) ? sc_ : 0.0;
            }
            double su = usum;
            su += __shfl_xor(su, 16, 64);
            su += __shfl_xor(su, 32, 64);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) lds[C::OFF_VEC + k] = su;                // row k: sum_r u_r (over the zero the loop above put there)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int J = 0; J < NT; ++J) tj[J] = lds[C::OFF_VEC + 16 * J + fr];          // column k: sum u; zero beyond
        {
            const double tk = lds[C::OFF_VEC + k];
            cz = fma(-(tk * invm), tk, cz);
        }
        const double mm = (double)hs.count;
        const double sc = n0 * (mm / (mm - 1.0));
        q0 = sc * cz;
        const double a = n0 + k + 2;
        cc = (2 * n0) / (a + sqrt(a * a + 4 * n0 * q0));
        // S0 = sc * C on the real columns, c * sc * C w0 in the border column, zero beyond
        const double fcol = colv ? sc : ((fr == kc) ? cc * sc : 0.0);
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            double ti[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ti[r] = -(lds[C::OFF_VEC + 16 * I + fq + 4 * r] * invm);
            static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int J = decltype(Jc)::value;
                constexpr int t = wtile(NT, I, J);
                d4 x = acc[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = fma(ti[r], tj[J], x[r]);
                if constexpr (J < kI) {
                    x *= sc;
                } else if constexpr (I < kI) {
                    x *= fcol;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] *= (fq + 4 * r < kc) ? fcol : 0.0;
                }
                acc[t] = x;
                wave_pin1<t>(acc[t]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_wave_barrier();
        if (dbg == 1) { dump_matrix(); return; }
    }
    TP_MARK(3);
    // ---- phase D: daily Gram (ref:180) + t in the border column (ref:222)
    {
        WRows ds;
        ds.base = A.panel; ds.ld = A.panel_ld;
        ds.ridx = (!LEAN && A.row_idx) ? A.row_idx + w * (long long)A.n_r : nullptr;
        ds.first = A.start ? A.start[w] : 0;
        ds.sub_row = A.rf_adj ? A.rf_adj + w * (long long)A.n_r : nullptr;
        ds.count = A.n_rows ? A.n_rows[w] : A.n_r;
        ds.count0 = 0x7fffffff; ds.jump = 0;
        ds.off32 = (A.panel_off32 & 1) != 0;
        double none[NT] = {};
        bool shared = false;
        const double* q = nullptr;
        if constexpr (LEAN) {
            // shared block-window sums (posterior_fused_impl.h, phase D): only the rows in front of the first whole
            // aligned 16-row block and behind the last one go through the MFMAs
            constexpr int BLK = 16;
            const long long b0 = (ds.first + BLK - 1) / BLK, b1 = (ds.first + ds.count) / BLK;
            const int Lw = (int)(b1 - b0);
            const int li = Lw == A.winsum_L[0] ? 0 : Lw == A.winsum_L[1] ? 1 : Lw == A.winsum_L[2] ? 2 : Lw == A.winsum_L[3] ? 3 : -1;
            shared = A.winsum != nullptr && Lw > 0 && li >= 0;
            if (shared) {
                ds.count0 = (int)(BLK * b0 - ds.first);
                ds.jump = (int)(BLK * b1 - ds.first) - ds.count0;
                ds.count = ds.count0 + (int)(ds.first + ds.count - BLK * b1);
                q = A.winsum + ((long long)li * A.prefix_nblk + b0) * ((long long)C::NTILES * 256);
            }
        }
        TP_MARK(32);
        // ONE table slot Q_L[b0]: [tile][2][64 lanes][2] doubles, two 16-byte reads per tile.  The reads run one group of
        // tiles ahead of the additions.
        typedef double d2 __attribute__((ext_vector_type(2)));
        constexpr int GRP = 7, NG = (C::NTILES + GRP - 1) / GRP;
        const d2* pq = (const d2*)q + lane;
        d2 qa[GRP][2], qb[GRP][2];
        auto qload = [&](d2 (&v2)[GRP][2], auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            static_for<0, GRP>([&](auto ec) __attribute__((always_inline)) {
                constexpr int e = decltype(ec)::value;
                constexpr int t = g * GRP + e;
                if constexpr (t < C::NTILES) {
                    v2[e][0] = pq[(long long)t * 128];
                    v2[e][1] = pq[(long long)t * 128 + 64];
                }
            });
        };
        auto qadd = [&](d2 (&v2)[GRP][2], auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            static_for<0, GRP>([&](auto ec) __attribute__((always_inline)) {
                constexpr int e = decltype(ec)::value;
                constexpr int t = g * GRP + e;
                if constexpr (t < C::NTILES) {
                    d4 x = acc[t];
                    x[0] += v2[e][0][0]; x[1] += v2[e][0][1];
                    x[2] += v2[e][1][0]; x[3] += v2[e][1][1];
                    acc[t] = x;
                    wave_pin1<t>(acc[t]);
                }
            });
        };
        // (every load of the intraday pass has completed: its staging region is free for the daily pass)
        if constexpr (!LEAN) wave_stage_rows(ds, lane, idx_rows_lds, idx_sub_lds);
        double nosum = 0.0;
        wave_gram<NT, false, LEAN>(ds, coff, k, lane, none, none, false, false, acc, idx_rows_lds, idx_sub_lds, none, nosum);
        TP_MARK(33);
        if (LEAN && shared) {
            // (issuing the first group in front of the edge rows' loop was measured and dropped: the loop's counted
            // vmcnt waits then also wait for the table reads - 0.648 vs 0.593 ms at configs[1])
            qload(qa, ic<0>{});
            static_for<0, NG>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g + 1 < NG) {
                    if constexpr (g % 2 == 0) qload(qb, ic<g + 1>{}); else qload(qa, ic<g + 1>{});
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (g % 2 == 0) qadd(qa, gc); else qadd(qb, gc);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
    TP_MARK(34);
    // rows >= k of the bordered matrix are never pivots: clear them (they hold 1'X, n_r, ...)
    static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int I = decltype(Ic)::value;
        constexpr int t = wtile(NT, I, kI);
        if constexpr (I == kI) {
            d4 x = acc[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = (fq + 4 * r >= kc) ? 0.0 : x[r];
            acc[t] = x;
            wave_pin1<t>(acc[t]);
        }
    });
    __builtin_amdgcn_sched_barrier(0);
    if (dbg == 2) { dump_matrix(); return; }

    if (!conj) {
        // ---- phase E: J = T - t t'/N (ref:600-601); t stays in the border column (ref:606 rhs)
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            constexpr int t = wtile(NT, I, kI);
            if (fr == kc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[C::OFF_VEC + 16 * I + fq + 4 * r] = acc[t][r];
            }
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const double invN = !FULL ? 1.0 / (double)A.N : A.center_rows == 2 ? 0.0
                          : 1.0 / (double)(A.center_rows ? (A.n_rows ? A.n_rows[w] : A.n_r) : A.N);
        const double sh_d = (FULL && A.shift) ? A.shift[2 * w] : 0.0;
        const double sh_e = (FULL && A.shift) ? A.shift[2 * w + 1] : 0.0;
        double tj[NT];
#pragma unroll
        for (int J = 0; J < NT; ++J) tj[J] = lds[C::OFF_VEC + 16 * J + fr];
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            double ti[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ti[r] = lds[C::OFF_VEC + 16 * I + fq + 4 * r];
            static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int J = decltype(Jc)::value;
                constexpr int t = wtile(NT, I, J);
                d4 x = acc[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool on = (J < kI || colv) && (I < kI || fq + 4 * r < kc);
                    const double add = sh_e + ((I == J && fq + 4 * r == fr) ? sh_d : 0.0);
                    x[r] += on ? add - invN * (ti[r] * tj[J]) : 0.0;
                }
                acc[t] = x;
                wave_pin1<t>(acc[t]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_wave_barrier();
    }
    if (FULL && A.rhs != nullptr) {
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            constexpr int t = wtile(NT, I, kI);
            if (fr == kc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = 16 * I + fq + 4 * r;
                    const int gl = gi < k ? gi : k - 1;
                    const double x = A.rhs[w * k + gl];
                    acc[t][r] = (gi < k) ? x : 0.0;
                }
            }
            wave_pin1<t>(acc[t]);
        });
    }
    if (FULL && A.out_rhs != nullptr) {
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            constexpr int t = wtile(NT, I, kI);
            if (fr == kc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = 16 * I + fq + 4 * r;
                    if (gi < k) A.out_rhs[w * k + gi] = acc[t][r];
                }
            }
        });
    }
    if (dbg == 3) dump_matrix();

    TP_MARK(4);
    // ---- phase F: blocked upper Cholesky S1 = R'R with the border column riding along (y = R^-T b)
    double badacc = 0.0;
    static_for<0, NT>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if (j < NTB) {
            const int npiv = (k - 16 * j < 16) ? (k - 16 * j) : 16;
            constexpr int tjj = wtile(NT, j, j);
            // (1) diagonal tile -> LDS (row-major) -> one column per lane; lanes 16-31 take the identity's columns
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[C::OFF_DG + (fq + 4 * r) * 16 + fr] = acc[tjj][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int c16 = lane & 15;
            double a[16];
            const double* src = (lane < 16) ? (lds + C::OFF_DG) : (lds + C::OFF_IDT);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = src[i * 16 + c16];
            // (2) 16 pivots: multipliers by v_readlane, rsqrt (v_rsq_f64 + one cubic step) with look-ahead.  Block rows in
            // front of the last one have 16 live pivots at compile time (no selects).  A non-positive or NaN pivot makes
            // its 1/sqrt an infinity or a NaN: 0 * rinv is accumulated per pivot (ONE instruction) and looked at once.
            constexpr bool ALL16 = j < kI;
            double d0 = readlane_d(a[0], 0);
            double rinv = rsqrt_cubic(d0);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (ALL16 || p < npiv) {
                    a[p] *= rinv;
                    badacc = fma(0.0, rinv, badacc);
                    double rinv_next = 1.0;
                    if (p + 1 < 16) {
                        const double s1 = readlane_d(a[p], p + 1);
                        a[p + 1] = fma(-s1, a[p], a[p + 1]);
                        double dn = readlane_d(a[p + 1], p + 1);
                        if (!ALL16) dn = (p + 1 < npiv) ? dn : 1.0;
                        rinv_next = rsqrt_cubic(dn);
                    }
#pragma unroll
                    for (int i = p + 2; i < 16; ++i) {
                        const double sI = readlane_d(a[p], i);
                        a[i] = fma(-sI, a[p], a[i]);
                    }
                    rinv = rinv_next;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // M_j = R_jj^-T (lower triangular), rows past the last pivot zeroed
            if (lane >= 16 && lane < 32) {
#pragma unroll
                for (int i = 0; i < 16; ++i) lds[C::OFF_M + (j * 16 + i) * C::MLD + c16] = (i < npiv) ? a[i] : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (3) block row j: R_jJ = M A_jJ (A operand: M, B operand: the tile's own registers)
            double mop[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mop[r] = lds[C::OFF_M + (j * 16 + fr) * C::MLD + 4 * r + fq];    // M[fr][4r + fq]
            static_for<j, NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int J = decltype(Jc)::value;
                if constexpr (J > j || j == kI) {        // R_jj itself is never used again (the solves use M_j)
                    constexpr int t = wtile(NT, j, J);
                    d4 rj = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) rj = __builtin_amdgcn_mfma_f64_16x16x4f64(mop[r], acc[t][r], rj, 0, 0, 0);
                    acc[t] = rj;
                }
            });
            // (4) trailing update A_IJ -= R_jI' R_jJ, operands straight from the block row's registers
            static_for<j + 1, NT>([&](auto Ic) __attribute__((always_inline)) {
                constexpr int I = decltype(Ic)::value;
                static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(Jc)::value;
                    constexpr int t = wtile(NT, I, J), tI = wtile(NT, j, I), tJ = wtile(NT, j, J);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[tI][r], acc[tJ][r], acc[t], 0, 0, 1);
                });
            });
        }
    });

    TP_MARK(5);
    // ---- phase G: y, q1 = y'y (ref:574), back substitution R w = y along block rows
    double q1 = 0.0;
    double wcol[NT];          // wcol[J] = w[16 J + fr] in every lane group
#pragma unroll
    for (int J = 0; J < NT; ++J) wcol[J] = 0.0;
    {
        double q1p = 0.0;
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            constexpr int t = wtile(NT, I, kI);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double y = (fr == kc && (I < kI || fq + 4 * r < kc)) ? acc[t][r] : 0.0;
                q1p = fma(y, y, q1p);
            }
        });
        q1 = wave_sum64(q1p);
    }
    static_for<0, NT>([&](auto ic) __attribute__((always_inline)) {
        constexpr int Ib = NT - 1 - decltype(ic)::value;
        if (Ib < NTB) {
            constexpr int tb = wtile(NT, Ib, kI);
            // z_r (rows 4r + fq of block Ib) = y - sum_{J > Ib} R_{Ib,J} w_J: lane-local products, ONE reduction per register
            double z[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double u = 0.0;
                static_for<Ib + 1, NT>([&](auto Jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(Jc)::value;
                    constexpr int t = wtile(NT, Ib, J);
                    if (J < NTB) u = fma(acc[t][r], wcol[J], u);
                });
                // the border column's y sits in lane fr == kc of its 16-lane row group: it joins the same reduction
                const double y = (fr == kc && (Ib < kI || fq + 4 * r < kc)) ? acc[tb][r] : 0.0;
                z[r] = rowgroup_sum16(y - u);
            }
            // w_Ib = M' z by MFMA: A[i][kk] = z[4r + kk] (any i), B[kk][n] = M[4r + kk][n]  =>  every row of the result is w'
            d4 wt = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double mrow = lds[C::OFF_M + (Ib * 16 + 4 * r + fq) * C::MLD + fr];
                wt = __builtin_amdgcn_mfma_f64_16x16x4f64(z[r], mrow, wt, 0, 0, 0);
            }
            wcol[Ib] = wt[0];
            __builtin_amdgcn_sched_barrier(0);
        }
    });

    TP_MARK(6);
    // ---- phase H: weights, status, aux (ref:572-575, 836 / 849)
    {
        const double n1 = n0 + (double)A.N;
        const double denom = n1 - q1;
        bool nonfinite = false;
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            const int gi = 16 * I + fr;
            double out;
            if (conj) out = 1.0 / A.gamma * ((n1 + k + 2) * wcol[I] / denom);
            else out = 1.0 / A.gamma * wcol[I];
            if (fq == (I & 3) && gi < k) {
                A.weights[w * k + gi] = out;
                if (!isfinite(out)) nonfinite = true;
            }
        });
        const bool anybad = __any(nonfinite ? 1 : 0) != 0;
        const bool notpd = __any((badacc != badacc) ? 1 : 0) != 0;
        if (lane == 0) {
            int st = TP_KSTATUS_OK;
            if (notpd) st = TP_KSTATUS_NOT_PD;
            else if (anybad) st = TP_KSTATUS_NONFINITE;
            else if (conj && !(denom > 0.0)) st = TP_KSTATUS_BAD_DENOM;
            A.status[w] = st;
            if (A.aux) {
                double* ax = A.aux + w * 8;
                ax[0] = n0; ax[1] = conj ? n1 : 0.0; ax[2] = cc; ax[3] = q0; ax[4] = q1;
                ax[5] = conj ? denom : 0.0; ax[6] = 0.0; ax[7] = 0.0;
            }
        }
    }
    TP_MARK(7);
}

// Windows (= waves) per SIMD the register allocator is asked to keep: the accumulators take 8 registers per tile, the
// row pipeline about 100; two or more waves per SIMD also double the vector issue rate (one wave alone issues an fp64
// vector instruction every ~9 cycles, two get one every ~4.75: tools/coexec_probe.hip).
#ifdef TP_WAVE_OCC
constexpr int wave_occupancy(int) { return TP_WAVE_OCC; }
#else
constexpr int wave_occupancy(int nt) { return nt <= 3 ? 4 : nt <= 5 ? 2 : 1; }
#endif

template <int NT, bool LEAN, int MODE>
__global__ void __launch_bounds__(64, wave_occupancy(NT)) posterior_wave_kernel(const tp_kargs_t A) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    wave_window_body<NT, LEAN, MODE>(A, lds);
}

// LDS bytes of a launch: the fixed image, plus the staging region of the general layout (sized by the batch's passes)
template <int NT, bool LEAN>
inline int wave_lds_bytes(const tp_kargs_t& a) {
    return WCfg<NT, LEAN>::LDS_BYTES + (LEAN ? 0 : wave_idx_bytes(a.n_r, a.m, a.strategy == 0));
}

template <int NT, bool LEAN, int MODE>
hipError_t wave_launch_mode(const tp_kargs_t& a, int grid8, hipStream_t stream) {
    const int lds_bytes = wave_lds_bytes<NT, LEAN>(a);
    if (lds_bytes > WAVE_LDS_LIMIT) return hipErrorNotSupported;     // nothing launched: launch_one falls back to the multi-wave kernel
    static std::atomic<unsigned long long> attr_done{0};      // one bit per device (tp_allow_dynamic_lds)
    { hipError_t e = tp_allow_dynamic_lds(attr_done, posterior_wave_kernel<NT, LEAN, MODE>, WAVE_LDS_LIMIT); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((posterior_wave_kernel<NT, LEAN, MODE>), dim3(grid8), dim3(64), lds_bytes, stream, a);
    return hipGetLastError();
}

template <int NT, bool LEAN>
hipError_t wave_launch_variant(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info) {
    if (info) { info->grid = grid; info->block = 64; info->lds_bytes = wave_lds_bytes<NT, LEAN>(a); info->ntile = NT; }
    const int grid8 = 8 * ((grid + 7) / 8);
    switch (wave_mode(a)) {
        case 0: return wave_launch_mode<NT, LEAN, 0>(a, grid8, stream);
        case 1: return wave_launch_mode<NT, LEAN, 1>(a, grid8, stream);
        default: return hipErrorNotSupported;        // launch_one keeps such batches on the multi-wave kernel
    }
}

}  // namespace
