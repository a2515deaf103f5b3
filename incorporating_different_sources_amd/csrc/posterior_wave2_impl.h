#pragma once
// posterior_wave2_impl.h - a rolling window on NWV = 2 (or 4) wavefronts, one per SIMD of a CU: the one-wave kernel's
// data flow (posterior_wave_impl.h) for universes whose bordered matrix no longer fits ONE wave's registers
// (10 .. 15 tiles per side, 144 <= k <= 239).
//
// The multi-wave kernel these sizes ran on (posterior_fused_impl.h) stages every row global -> registers -> LDS -> MFMA
// operand, ends every phase at a workgroup barrier and spills (0.66 GB of scratch write-back per launch at k = 100; wave
// time 61 % parked at barriers: profiles/r02_final_pmc_stall_fused_k100.txt).  Here
//   * the upper triangle of the bordered matrix is split by tile COLUMN: wave own(J) holds every tile (I, J), I <= J, of
//     its columns in accumulator registers (snake order over the columns from the widest down, so every trailing
//     sub-matrix of the factorisation stays balanced too): 39 + 39 tiles at 12 tiles per side, 30 x 4 at 15;
//   * the Gram passes are the one-wave kernel's: each wave loads the MFMA operands of a 4-row k-step straight from the
//     panels (one 8-byte load per 16-column group IS the A / B operand of every tile in that tile row / column) and
//     feeds its own tiles - no LDS staging, no barrier inside a pass (the waves load the same rows: the second request
//     is an L1 / L2 hit);
//   * the blocked Cholesky exchanges ONE block row per step through LDS: the owner of the diagonal tile runs the
//     16-pivot chain and publishes M = R_jj^-T; every wave turns its tiles of the block row into R_jJ = M A_jJ with
//     the tile's own registers as the B operand and stores them - register r of a tile is rows 4r..4r+3 in MFMA operand
//     layout, so the LDS image is a plain lane-linear dump - and the trailing update A_IJ -= R_jI' R_jJ reads the A
//     operands it does not own from that image.  NO workgroup barrier inside the factorisation: the waves meet through
//     flags in LDS, and the owner of the next diagonal tile runs its pivot chain while the others finish the trailing
//     update (data-flow form, phase F below; one instantiation keeps a barrier pair per block step, w2_dataflow);
//   * the back substitution runs along block rows: lane-local products over a wave's own tiles, one 16-lane DPP
//     reduction per register, the NWV partial vectors meet in LDS (one barrier per block row, double-buffered) and every
//     wave forms w_I = M' z for itself.
// Plain conjugate / Jeffreys batches only (MODE 0 / 1), like the one-wave kernel; read-backs, custom right-hand sides
// and shifts stay on the multi-wave kernel.  ref:LINE cites /root/reference/src/portfolio_calculations.py.
#include "posterior_wave_impl.h"

namespace {

// ---- ownership: tile column J belongs to wave w2_owner(J); snake over the columns from the widest (J = NT-1) down
template <int NT, int NWV>
constexpr int w2_owner(int J) {
    const int p = NT - 1 - J, g = p / NWV, i = p % NWV;
    return (g & 1) ? NWV - 1 - i : i;
}
template <int NT, int NWV>
constexpr int w2_count(int WV) {
    int n = 0;
    for (int J = 0; J < NT; ++J)
        if (w2_owner<NT, NWV>(J) == WV) n += J + 1;
    return n;
}
template <int NT, int NWV>
constexpr int w2_max_tiles() {
    int m = 0;
    for (int w = 0; w < NWV; ++w) m = w2_count<NT, NWV>(w) > m ? w2_count<NT, NWV>(w) : m;
    return m;
}
// slot of tile (I, J) among its owner's tiles (row-major over the owner's tiles); -1 when I > J
template <int NT, int NWV>
constexpr int w2_slot(int I, int J) {
    const int wv = w2_owner<NT, NWV>(J);
    int n = 0;
    for (int i = 0; i < NT; ++i)
        for (int j = i; j < NT; ++j) {
            if (i == I && j == J) return n;
            if (w2_owner<NT, NWV>(j) == wv) ++n;
        }
    return -1;
}
// does wave WV own any tile column J >= I (i.e. does it need the A operand of tile row I in a trailing update)?
template <int NT, int NWV>
constexpr bool w2_owns_from(int WV, int I) {
    for (int J = I; J < NT; ++J)
        if (w2_owner<NT, NWV>(J) == WV) return true;
    return false;
}

// Which instantiations factorise in the data-flow form.  At 11 and 12 tiles per side (39 tiles per wave, seven of them in
// VGPRs) its extra live values made the register allocator move an accumulator tile between the two register files INSIDE
// the row loop of the general-layout conjugate kernel - a read of an MFMA result two instructions after its issue, which
// tools/check_mfma_hazards.py rejects - so that one keeps the barrier form.  TP_WAVE2_DATAFLOW = 0 / 1 forces one form
// everywhere (A/B builds).
#ifdef TP_WAVE2_DATAFLOW
constexpr bool w2_dataflow(int, bool, int) { return TP_WAVE2_DATAFLOW != 0; }
#else
constexpr bool w2_dataflow(int nt, bool lean, int mode) { return !((nt == 11 || nt == 12) && !lean && mode == 0); }
#endif

template <int NT_, int NWV_, bool DF_>
struct W2Cfg {
    static constexpr int NT = NT_, NWV = NWV_;
    static constexpr int KP = 16 * NT;
    static constexpr int NTILES = NT * (NT + 1) / 2;
    static constexpr int NTHREADS = 64 * NWV;
    static constexpr int MLD = 17;
    static constexpr int OFF_M = 0;                              // [NT][16][MLD]  M_j = R_jj^-T, row-major
    static constexpr int OFF_DG = OFF_M + NT * 16 * MLD;         // [16][16] diagonal tile handed to the pivot chain
    static constexpr int OFF_IDT = OFF_DG + 256;                 // [16][16] identity
    static constexpr int OFF_VEC = OFF_IDT + 256;                // [KP] column sums / Jeffreys t
    // block row j in MFMA operand layout: tiles (j, J), j < J < NT-1, at slot J - 1 (tile (j, NT-1) is read by its owner only)
    static constexpr int RB_TILES = NT > 2 ? NT - 2 : 1;
    static constexpr int OFF_RB = OFF_VEC + KP;                  // [RB_TILES][4][64]
    static constexpr int OFF_PART = OFF_RB + RB_TILES * 256;     // [2][NWV][16] partial sums of the back substitution
    static constexpr int OFF_SCAL = OFF_PART + 2 * NWV * 16;     // [16]: 0 = corner z'z, 1 = q1, 4.. = per-wave not-PD flags
    // data-flow factorisation only (w2_dataflow(NT)): a second block-row image and diagonal buffer (steps alternate) and
    // the flags - ints: [NT] "M_j published", then [NT][NWV] "wave w's tiles of block row j published"
    static constexpr bool DF = DF_;
    static constexpr int OFF_RB2 = OFF_SCAL + 16;
    static constexpr int OFF_DG2 = OFF_RB2 + (DF ? RB_TILES * 256 : 0);
    static constexpr int OFF_FLAG = OFF_DG2 + (DF ? 256 : 0);
    static constexpr int LDS_DOUBLES = OFF_FLAG + (DF ? (NT * (1 + NWV) + 1) / 2 : 0);
    static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
    static constexpr int OFF_SUB = LDS_DOUBLES;                  // general layout: staged rows of the pass (wave_idx_rows)
};
constexpr int WAVE2_LDS_LIMIT = 160 * 1024;

template <int T>
__device__ __forceinline__ void w2_touch1(d4& c) {      // an ordered (volatile) "modification" of one tile, no instruction
    if constexpr (wave_tile_in_agpr(T)) asm volatile("" : "+a"(c));
    else asm volatile("" : "+v"(c));
}
template <int N>
__device__ __forceinline__ void w2_pin(d4 (&acc)[N]) {
    static_for<0, N>([&](auto tc) __attribute__((always_inline)) { wave_pin1<decltype(tc)::value>(acc[decltype(tc)::value]); });
}
// end of a pass of inline-assembly MFMAs: 24 wait states once, then every tile is "modified" behind them (asm volatile
// statements keep their order), so that no use of an accumulator can be scheduled in front of the wait
template <int N>
__device__ __forceinline__ void w2_settle(d4 (&acc)[N]) {
    wave_settle1<0>(acc[0]);
    static_for<1, N>([&](auto tc) __attribute__((always_inline)) { w2_touch1<decltype(tc)::value>(acc[decltype(tc)::value]); });
}

// One pass over the rows of a window for wave WV: acc(I, J) += rows[:, I]' rows[:, J] for its own tile columns J.
// Everything else as wave_gram (posterior_wave_impl.h).
template <int NT, int NWV, int WV, bool HF, bool LEAN>
__device__ __forceinline__ void w2_gram(const WRows& src, const long long (&coff)[NT], int k, int lane,
                                        double (&shift)[NT], double (&w0v)[NT], bool ones, bool lazy_mask,
                                        d4 (&acc)[w2_count<NT, NWV>(WV)], const int* lds_rows, const double* lds_sub,
                                        double (&csum)[NT], double& usum) {
    constexpr int NS = w2_count<NT, NWV>(WV);
    constexpr int kI = NT - 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int kc = k - 16 * kI;
    const bool cvl = fr < kc;
    const double border = HF ? ((ones && fr == kc + 1) ? 1.0 : 0.0) : ((fr == kc) ? 1.0 : 0.0);
    const int nks = (src.count + 3) >> 2;
    const bool has_sub = !HF && src.sub_row != nullptr;

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
            prefetch(ks + 1);
        } else {
            if (has_sub) sub = src.sub_row[r];
        }
    };
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
                // shifted rows (and the sum of u) are kept by vector adds instead: NT + 1 per k-step next to the MFMAs.
                // Round 2 ran a separate pass for the column MEANS at these sizes (13.8 % of a window's time at k = 191).
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
                if constexpr (w2_owner<NT, NWV>(J) == WV) {
                    constexpr int t = w2_slot<NT, NWV>(I, J);
                    wave_mfma_agpr<t>(acc[t], v[I], v[J]);
                }
            });
        });
    };

    if (nks <= 0) return;
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
    w2_pin<NS>(acc);
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
    w2_settle<NS>(acc);
}

// ---- data-flow synchronisation of the factorisation (flags in LDS instead of workgroup barriers) -------------------------
// publish: every lane's LDS writes are complete (release fence = s_waitcnt lgkmcnt(0)), then lane 0 raises the flag
__device__ __forceinline__ void w2_publish(int* flag, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *(volatile int*)flag = 1;
}
// wait: spin on the flag (s_sleep between reads), then an acquire fence so that no later LDS read is moved in front of it.
// The spin is BOUNDED (2^20 iterations of >= 64 cycles: tens of milliseconds, ten thousand times the longest legitimate
// wait): a bug must never hang the GPU - the window's results are then garbage and the caller sees it (the bound also
// raises the not-positive-definite flag).
__device__ __forceinline__ bool w2_wait(const int* flag) {
    int spins = 0;
    bool ok = true;
    while (*(const volatile int*)flag == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) { ok = false; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

// general layout: panel row (and subtrahend) of every row of the pass into LDS, by all threads of the workgroup
__device__ __forceinline__ void w2_stage_rows(const WRows& src, int tid, int nthreads, int* li, double* lsb) {
    for (int i = tid; i < src.count; i += nthreads) {
        li[i] = src.ridx ? src.ridx[i] : (int)(src.first + i);
        if (src.sub_row) lsb[i] = src.sub_row[i];
    }
    __syncthreads();
}

template <int NT, int NWV, int WV, bool LEAN, int MODE>
__device__ __forceinline__ void w2_body(const tp_kargs_t& A, double* lds) {
    using C = W2Cfg<NT, NWV, w2_dataflow(NT, LEAN, MODE)>;
    constexpr int NS = w2_count<NT, NWV>(WV);
    constexpr int kI = NT - 1;
    constexpr int OWN_KI = w2_owner<NT, NWV>(kI);            // the wave that holds the border column (b, then y)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int k = A.k;
    __builtin_assume(k >= 16 * (NT - 1));
    __builtin_assume(k <= 16 * NT - 1);
    const int kc = k - 16 * kI;
    const int NTB = (kc == 0) ? NT - 1 : NT;
    const bool colv = fr < kc;
    // XCD-aware workgroup -> window map (posterior_fused_impl.h)
    const long long per_xcd = (A.w_count + 7) >> 3;
    const long long wl = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= per_xcd || wl >= A.w_count) return;
    const long long w = A.w_first + wl;
    const int tid0 = tid;
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
    double* idx_sub_lds = LEAN ? nullptr : lds + C::OFF_SUB;
    int* idx_rows_lds = LEAN ? nullptr : (int*)(lds + C::OFF_SUB + wave_idx_rows(A.n_r, A.m, A.strategy == 0));
    d4 acc[NS];
    static_for<0, NS>([&](auto tc) __attribute__((always_inline)) { acc[decltype(tc)::value] = d4{0.0, 0.0, 0.0, 0.0}; });
    w2_pin<NS>(acc);

    if constexpr (WV == 0) {            // identity tile for the pivot chains (published by the barriers of phase C / E)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i;
            lds[C::OFF_IDT + e] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
        }
        if constexpr (C::DF) {
            int* fl = (int*)(lds + C::OFF_FLAG);                       // the factorisation's flags, likewise
            for (int i = lane; i < NT * (1 + NWV); i += 64) fl[i] = 0;
        }
    }

    double n0 = 0.0, cc = 0.0, q0 = 0.0;
    constexpr bool conj = MODE == 0;

    TP_MARK(0);
    if constexpr (conj) {
        n0 = A.n0[w];
        WRows hs;
        hs.base = A.hf_panel; hs.ld = A.hf_ld;
        hs.ridx = (!LEAN && A.hf_row_idx) ? A.hf_row_idx + w * (long long)A.m : nullptr;
        hs.first = A.hf_start ? A.hf_start[w] : 0;
        hs.sub_row = nullptr;
        hs.count = A.hf_count ? A.hf_count[w] : A.m;
        hs.count0 = 0x7fffffff; hs.jump = 0;
        hs.off32 = (A.hf_off32 & 1) != 0;
        // ---- phase A: the shift row of the one-pass centred scatter = the window's first intraday row; every wave for itself
        const bool ones = kc < 15;        // a spare column k+1 carries ones; otherwise the sums are kept by vector adds (w2_gram)
        double shift[NT], w0v[NT], csum[NT];
        double usum = 0.0;
        {
            const long long row0 = hs.ridx ? (long long)hs.ridx[0] : hs.first;
            const double* p0 = hs.base + row0 * (long long)hs.ld;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int c = 16 * i + fr;
                const int cl = c < k ? c : k - 1;
                shift[i] = p0[coff[i]];
                w0v[i] = A.w0[w * k + cl];
                csum[i] = 0.0;
            }
        }
        TP_MARK(1);
        // ---- phase B: Gram of the shifted intraday rows (row 0, shifted by itself, is exactly zero: skipped)
        const int hf_rows_all = hs.count;
        if (!hs.ridx) { hs.first += 1; hs.count -= 1; }
        else { hs.ridx += 1; hs.count -= 1; }
        if constexpr (!LEAN) w2_stage_rows(hs, tid, C::NTHREADS, idx_rows_lds, idx_sub_lds);
        w2_gram<NT, NWV, WV, true, LEAN>(hs, coff, k, lane, shift, w0v, ones, true, acc, idx_rows_lds, idx_sub_lds, csum, usum);
        hs.count = hf_rows_all;
        TP_MARK(2);
        // ---- phase C: rank-one term of the centring, q0, c, scaling (ref:333, 415-418)
        const double invm = 1.0 / (double)hs.count;
        if constexpr (WV == OWN_KI) {
            // corner element (row kc, column kc of tile (kI, kI)): z'z = w0'C w0 before the rank-one term.  All four
            // registers are read and the scalar results selected (posterior_wave_impl.h: a compiler trap otherwise)
            double cz0;
            {
                constexpr int t = w2_slot<NT, NWV>(kI, kI);
                const int ln = __builtin_amdgcn_readfirstlane(16 * (kc & 3) + kc);
                const double x0 = readlane_d(acc[t][0], ln), x1 = readlane_d(acc[t][1], ln);
                const double x2 = readlane_d(acc[t][2], ln), x3 = readlane_d(acc[t][3], ln);
                const int rr = kc >> 2;
                cz0 = rr == 0 ? x0 : rr == 1 ? x1 : rr == 2 ? x2 : x3;
            }
            if (ones) {
                // column k+1 holds t_i = sum_r (y_r - s)_i for the asset columns and sum_r u_r in row k
                static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
                    constexpr int I = decltype(Ic)::value;
                    constexpr int t = w2_slot<NT, NWV>(I, kI);
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
                if (lane == 0) lds[C::OFF_VEC + k] = su;            // row k: sum_r u_r (after the zero the loop above put there)
            }
            if (lane == 0) lds[C::OFF_SCAL + 0] = cz0;
        }
        __syncthreads();
        double tj[NT];
        double cz = lds[C::OFF_SCAL + 0];
        const double tk = lds[C::OFF_VEC + k];
#pragma unroll
        for (int J = 0; J < NT; ++J) tj[J] = lds[C::OFF_VEC + 16 * J + fr];          // column k: sum u; zero beyond
        cz = fma(-(tk * invm), tk, cz);
        const double mm = (double)hs.count;
        const double sc = n0 * (mm / (mm - 1.0));
        q0 = sc * cz;
        const double a = n0 + k + 2;
        cc = (2 * n0) / (a + sqrt(a * a + 4 * n0 * q0));
        const double fcol = colv ? sc : ((fr == kc) ? cc * sc : 0.0);
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            if constexpr (w2_owns_from<NT, NWV>(WV, I)) {
                double ti[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ti[r] = -(lds[C::OFF_VEC + 16 * I + fq + 4 * r] * invm);
                static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(Jc)::value;
                    if constexpr (w2_owner<NT, NWV>(J) == WV) {
                        constexpr int t = w2_slot<NT, NWV>(I, J);
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
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        });
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
            // shared block-window sums: only the rows in front of the first whole aligned block and behind the last one
            // go through the MFMAs
            constexpr int BLK = TP_PREFIX_BLOCK_ROWS(NT);
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
        if constexpr (!LEAN) w2_stage_rows(ds, tid, C::NTHREADS, idx_rows_lds, idx_sub_lds);
        TP_MARK(32);
        double nosum = 0.0;
        w2_gram<NT, NWV, WV, false, LEAN>(ds, coff, k, lane, none, none, false, false, acc, idx_rows_lds, idx_sub_lds, none, nosum);
        TP_MARK(33);
        if (LEAN && shared) {
            // this wave's tiles of the table slot Q_L[b0]: [tile][2][64 lanes][2] doubles (the table numbers the tiles
            // row-major over the whole triangle), two 16-byte reads per tile, six tiles' reads ahead of the additions
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2* pq = (const d2*)q + lane;
            constexpr int GRP = NS > 32 ? 3 : 6, NG = (NS + GRP - 1) / GRP;
            d2 qa[GRP][2], qb[GRP][2];
            auto qload = [&](d2 (&v2)[GRP][2], auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
                    constexpr int I = decltype(Ic)::value;
                    static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                        constexpr int J = decltype(Jc)::value;
                        if constexpr (w2_owner<NT, NWV>(J) == WV) {
                            constexpr int s = w2_slot<NT, NWV>(I, J);
                            if constexpr (s / GRP == g) {
                                constexpr long long t = wtile(NT, I, J);
                                v2[s % GRP][0] = pq[t * 128];
                                v2[s % GRP][1] = pq[t * 128 + 64];
                            }
                        }
                    });
                });
            };
            auto qadd = [&](d2 (&v2)[GRP][2], auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                static_for<0, GRP>([&](auto ec) __attribute__((always_inline)) {
                    constexpr int s = g * GRP + decltype(ec)::value;
                    if constexpr (s < NS) {
                        d4 x = acc[s];
                        x[0] += v2[s % GRP][0][0]; x[1] += v2[s % GRP][0][1];
                        x[2] += v2[s % GRP][1][0]; x[3] += v2[s % GRP][1][1];
                        acc[s] = x;
                        wave_pin1<s>(acc[s]);
                    }
                });
            };
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
    if constexpr (WV == OWN_KI) {
        constexpr int t = w2_slot<NT, NWV>(kI, kI);
        d4 x = acc[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = (fq + 4 * r >= kc) ? 0.0 : x[r];
        acc[t] = x;
        wave_pin1<t>(acc[t]);
    }
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (!conj) {
        // ---- phase E: J = T - t t'/N (ref:600-601); t stays in the border column (ref:606 rhs)
        if constexpr (WV == OWN_KI) {
            static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
                constexpr int I = decltype(Ic)::value;
                constexpr int t = w2_slot<NT, NWV>(I, kI);
                if (fr == kc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) lds[C::OFF_VEC + 16 * I + fq + 4 * r] = acc[t][r];
                }
            });
        }
        __syncthreads();
        const double invN = 1.0 / (double)A.N;
        double tj[NT];
#pragma unroll
        for (int J = 0; J < NT; ++J) tj[J] = lds[C::OFF_VEC + 16 * J + fr];
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            if constexpr (w2_owns_from<NT, NWV>(WV, I)) {
                double ti[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ti[r] = lds[C::OFF_VEC + 16 * I + fq + 4 * r];
                static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(Jc)::value;
                    if constexpr (w2_owner<NT, NWV>(J) == WV) {
                        constexpr int t = w2_slot<NT, NWV>(I, J);
                        d4 x = acc[t];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool on = (J < kI || colv) && (I < kI || fq + 4 * r < kc);
                            x[r] += on ? -invN * (ti[r] * tj[J]) : 0.0;
                        }
                        acc[t] = x;
                        wave_pin1<t>(acc[t]);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }

    TP_MARK(4);
    // ---- phase F: blocked upper Cholesky S1 = R'R with the border column riding along (y = R^-T b)
    double badacc = 0.0;
    if constexpr (C::DF) {
    // Data-flow form (round 3): NO workgroup barrier inside the factorisation.  The 16-pivot chains are a serial path
    // through the block steps - chain(j+1) needs only tile (j+1, j+1) updated through step j - and with a barrier pair per
    // step EVERY wave waited for EVERY chain (12 - 15 x ~5 k cycles: 19 % of a window's time at k = 191, 28 % at k = 223).
    // Here the owner of column j+1 brings that one tile up to date first, runs its chain and publishes M_{j+1} while the
    // other waves - and then itself - do the rest of step j's trailing update; waves meet only through two kinds of flags,
    // "M_j published" and "wave w's tiles of block row j published", and the block-row image and the diagonal buffer
    // alternate between two copies (a wave writes copy j % 2 only after its trailing update of step j-1, which needed every
    // other wave's flag of step j-1, which each raised after finishing step j-2's reads of that copy).
    int* mflag = (int*)(lds + C::OFF_FLAG);
    int* rbflag = mflag + NT;
    bool sync_ok = true;
    auto chain = [&](auto jc) __attribute__((always_inline)) {       // tile (j, j) -> M_j = R_jj^-T, published
        constexpr int j = decltype(jc)::value;
        constexpr int tjj = w2_slot<NT, NWV>(j, j);
        const int npiv = (k - 16 * j < 16) ? (k - 16 * j) : 16;
        double* dg = lds + ((j & 1) ? C::OFF_DG2 : C::OFF_DG);
#pragma unroll
        for (int r = 0; r < 4; ++r) dg[(fq + 4 * r) * 16 + fr] = acc[tjj][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int c16 = lane & 15;
        double a[16];
        const double* src = (lane < 16) ? dg : (lds + C::OFF_IDT);
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = src[i * 16 + c16];
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
        if (lane >= 16 && lane < 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) lds[C::OFF_M + (j * 16 + i) * C::MLD + c16] = (i < npiv) ? a[i] : 0.0;
        }
        w2_publish(mflag + j, lane);
    };
    if constexpr (w2_owner<NT, NWV>(0) == WV) chain(ic<0>{});
    static_for<0, NT>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if (j < NTB) {
            const int rbo = (j & 1) ? C::OFF_RB2 : C::OFF_RB;
            // (3) this wave's tiles of block row j: R_jJ = M_j A_jJ, stored to the image for the waves that need them
            if constexpr (w2_owns_from<NT, NWV>(WV, j)) {
                sync_ok = w2_wait(mflag + j) && sync_ok;
                double mop[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) mop[r] = lds[C::OFF_M + (j * 16 + fr) * C::MLD + 4 * r + fq];    // M[fr][4r + fq]
                static_for<j, NT>([&](auto Jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(Jc)::value;
                    if constexpr (w2_owner<NT, NWV>(J) == WV && (J > j || j == kI)) {   // R_jj itself is never used again
                        constexpr int t = w2_slot<NT, NWV>(j, J);
                        d4 rj = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int r = 0; r < 4; ++r) rj = __builtin_amdgcn_mfma_f64_16x16x4f64(mop[r], acc[t][r], rj, 0, 0, 0);
                        acc[t] = rj;
                        if constexpr (J > j && J < kI) {                 // (j, kI) is read by its owner only
#pragma unroll
                            for (int r = 0; r < 4; ++r) lds[rbo + (J - 1) * 256 + r * 64 + lane] = rj[r];
                        }
                    }
                });
            }
            w2_publish(rbflag + j * NWV + WV, lane);                   // (every wave, also one without tiles in this row)
            if constexpr (j + 1 < NT) {
                // the serial path first: the next diagonal tile and its chain (operands of this wave only)
                constexpr bool OWN_NEXT = w2_owner<NT, NWV>(j + 1) == WV;
                if constexpr (OWN_NEXT) {
                    constexpr int t = w2_slot<NT, NWV>(j + 1, j + 1), tJ = w2_slot<NT, NWV>(j, j + 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[tJ][r], acc[tJ][r], acc[t], 0, 0, 1);
                    if (j + 1 < NTB) chain(ic<j + 1>{});
                }
                // (4) the rest of the trailing update A_IJ -= R_jI' R_jJ: it needs the other waves' tiles of block row j
                static_for<0, NWV>([&](auto vc) __attribute__((always_inline)) {
                    constexpr int v = decltype(vc)::value;
                    if constexpr (v != WV) sync_ok = w2_wait(rbflag + j * NWV + v) && sync_ok;
                });
                static_for<j + 1, NT>([&](auto Ic) __attribute__((always_inline)) {
                    constexpr int I = decltype(Ic)::value;
                    if constexpr (w2_owns_from<NT, NWV>(WV, I)) {
                        double aI[4];
                        if constexpr (w2_owner<NT, NWV>(I) == WV) {
                            constexpr int tI = w2_slot<NT, NWV>(j, I);
#pragma unroll
                            for (int r = 0; r < 4; ++r) aI[r] = acc[tI][r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) aI[r] = lds[rbo + (I - 1) * 256 + r * 64 + lane];
                        }
                        static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                            constexpr int J = decltype(Jc)::value;
                            if constexpr (w2_owner<NT, NWV>(J) == WV && !(OWN_NEXT && I == j + 1 && J == j + 1)) {
                                constexpr int t = w2_slot<NT, NWV>(I, J), tJ = w2_slot<NT, NWV>(j, J);
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI[r], acc[tJ][r], acc[t], 0, 0, 1);
                            }
                        });
                    }
                });
            }
        }
    });
    if (!sync_ok) badacc = __builtin_nan("");                          // (never in a correct run: see w2_wait)
    __syncthreads();                                                   // every wave has left the factorisation
    } else {
    // barrier form: the owner of the diagonal tile runs the chain while every other wave waits (A/B builds)
    static_for<0, NT>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if (j < NTB) {
            const int npiv = (k - 16 * j < 16) ? (k - 16 * j) : 16;
            if constexpr (w2_owner<NT, NWV>(j) == WV) {
                constexpr int tjj = w2_slot<NT, NWV>(j, j);
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
                // (2) 16 pivots (posterior_wave_impl.h, phase F)
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
            }
            __syncthreads();                                           // M_j is published
            // (3) this wave's tiles of block row j: R_jJ = M A_jJ (A operand: M, B operand: the tile's own registers),
            // stored to the block-row image for the waves that need them as A operands
            if constexpr (w2_owns_from<NT, NWV>(WV, j)) {
                double mop[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) mop[r] = lds[C::OFF_M + (j * 16 + fr) * C::MLD + 4 * r + fq];    // M[fr][4r + fq]
                static_for<j, NT>([&](auto Jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(Jc)::value;
                    if constexpr (w2_owner<NT, NWV>(J) == WV && (J > j || j == kI)) {   // R_jj itself is never used again
                        constexpr int t = w2_slot<NT, NWV>(j, J);
                        d4 rj = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int r = 0; r < 4; ++r) rj = __builtin_amdgcn_mfma_f64_16x16x4f64(mop[r], acc[t][r], rj, 0, 0, 0);
                        acc[t] = rj;
                        if constexpr (J > j && J < kI) {                 // (j, kI) is read by its owner only
#pragma unroll
                            for (int r = 0; r < 4; ++r) lds[C::OFF_RB + (J - 1) * 256 + r * 64 + lane] = rj[r];
                        }
                    }
                });
            }
            if constexpr (j + 1 < NT) {
                __syncthreads();                                       // block row j is published
                // (4) trailing update A_IJ -= R_jI' R_jJ: B operand from this wave's registers, A operand from its
                // registers when it owns column I, else from the block-row image
                static_for<j + 1, NT>([&](auto Ic) __attribute__((always_inline)) {
                    constexpr int I = decltype(Ic)::value;
                    if constexpr (w2_owns_from<NT, NWV>(WV, I)) {
                        double aI[4];
                        if constexpr (w2_owner<NT, NWV>(I) == WV) {
                            constexpr int tI = w2_slot<NT, NWV>(j, I);
#pragma unroll
                            for (int r = 0; r < 4; ++r) aI[r] = acc[tI][r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) aI[r] = lds[C::OFF_RB + (I - 1) * 256 + r * 64 + lane];
                        }
                        static_for<I, NT>([&](auto Jc) __attribute__((always_inline)) {
                            constexpr int J = decltype(Jc)::value;
                            if constexpr (w2_owner<NT, NWV>(J) == WV) {
                                constexpr int t = w2_slot<NT, NWV>(I, J), tJ = w2_slot<NT, NWV>(j, J);
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI[r], acc[tJ][r], acc[t], 0, 0, 1);
                            }
                        });
                    }
                });
            }
        }
    });

    }

    TP_MARK(5);
    // ---- phase G: y, q1 = y'y (ref:574), back substitution R w = y along block rows
    {
        const bool notpd_w = __any((badacc != badacc) ? 1 : 0) != 0;
        if (lane == 0) lds[C::OFF_SCAL + 4 + WV] = notpd_w ? 1.0 : 0.0;
    }
    if constexpr (WV == OWN_KI) {
        double q1p = 0.0;
        static_for<0, NT>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value;
            constexpr int t = w2_slot<NT, NWV>(I, kI);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double y = (fr == kc && (I < kI || fq + 4 * r < kc)) ? acc[t][r] : 0.0;
                q1p = fma(y, y, q1p);
            }
        });
        const double q1w = wave_sum64(q1p);
        if (lane == 0) lds[C::OFF_SCAL + 1] = q1w;
    }
    double wcol[NT];          // wcol[J] = w[16 J + fr] in every lane group (every wave computes all of them)
#pragma unroll
    for (int J = 0; J < NT; ++J) wcol[J] = 0.0;
    static_for<0, NT>([&](auto ic_) __attribute__((always_inline)) {
        constexpr int Ib = NT - 1 - decltype(ic_)::value;
        if (Ib < NTB) {
            // this wave's share of z_r (rows 4r + fq of block Ib) = y - sum_{J > Ib} R_{Ib,J} w_J
            double part[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double u = 0.0;
                static_for<Ib + 1, NT>([&](auto Jc) __attribute__((always_inline)) {
                    constexpr int J = decltype(Jc)::value;
                    if constexpr (w2_owner<NT, NWV>(J) == WV) {
                        constexpr int t = w2_slot<NT, NWV>(Ib, J);
                        if (J < NTB) u = fma(acc[t][r], wcol[J], u);
                    }
                });
                double y = 0.0;
                if constexpr (WV == OWN_KI) {
                    constexpr int tb = w2_slot<NT, NWV>(Ib, kI);
                    y = (fr == kc && (Ib < kI || fq + 4 * r < kc)) ? acc[tb][r] : 0.0;
                }
                part[r] = rowgroup_sum16(y - u);
            }
            double* pbuf = lds + C::OFF_PART + (Ib & 1) * (NWV * 16);
            if (fr == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pbuf[WV * 16 + 4 * r + fq] = part[r];
            }
            __syncthreads();
            // w_Ib = M' z by MFMA: A[i][kk] = z[4r + kk] (any i), B[kk][n] = M[4r + kk][n]  =>  every row of the result is w'
            d4 wt = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double z = pbuf[4 * r + fq];
#pragma unroll
                for (int v = 1; v < NWV; ++v) z += pbuf[v * 16 + 4 * r + fq];
                const double mrow = lds[C::OFF_M + (Ib * 16 + 4 * r + fq) * C::MLD + fr];
                wt = __builtin_amdgcn_mfma_f64_16x16x4f64(z, mrow, wt, 0, 0, 0);
            }
            wcol[Ib] = wt[0];
            __builtin_amdgcn_sched_barrier(0);
        }
    });

    TP_MARK(6);
    // ---- phase H: weights, status, aux (ref:572-575, 836 / 849) - wave 0 writes them
    __syncthreads();          // (NTB == 0 cannot happen: k >= 16 (NT-1) >= 16; the flags and q1 were published by the barriers above)
    if constexpr (WV == 0) {
        const double q1 = lds[C::OFF_SCAL + 1];
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
        bool notpd = false;
#pragma unroll
        for (int v = 0; v < NWV; ++v) notpd = notpd || (lds[C::OFF_SCAL + 4 + v] != 0.0);
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

// waves per SIMD the register allocator is asked to keep (TP_WAVE2_OCC: A/B builds; 2 needs <= 14 tiles per wave)
#ifndef TP_WAVE2_OCC
#define TP_WAVE2_OCC 1
#endif
template <int NT, int NWV, bool LEAN, int MODE>
__global__ void __launch_bounds__(64 * NWV, TP_WAVE2_OCC) posterior_wave2_kernel(const tp_kargs_t A) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    wave_dispatch<NWV>(wv, [&](auto wc) __attribute__((always_inline)) {
        w2_body<NT, NWV, decltype(wc)::value, LEAN, MODE>(A, lds);
    });
}

template <int NT, int NWV, bool LEAN, int MODE>
inline int wave2_lds_bytes(const tp_kargs_t& a) {
    return W2Cfg<NT, NWV, w2_dataflow(NT, LEAN, MODE)>::LDS_BYTES + (LEAN ? 0 : wave_idx_bytes(a.n_r, a.m, a.strategy == 0));
}

template <int NT, int NWV, bool LEAN, int MODE>
hipError_t wave2_launch_mode(const tp_kargs_t& a, int grid8, hipStream_t stream) {
    const int lds_bytes = wave2_lds_bytes<NT, NWV, LEAN, MODE>(a);
    if (lds_bytes > WAVE2_LDS_LIMIT) return hipErrorNotSupported;     // nothing launched: launch_one falls back
    static std::atomic<unsigned long long> attr_done{0};      // one bit per device (tp_allow_dynamic_lds)
    { hipError_t e = tp_allow_dynamic_lds(attr_done, posterior_wave2_kernel<NT, NWV, LEAN, MODE>, WAVE2_LDS_LIMIT); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL((posterior_wave2_kernel<NT, NWV, LEAN, MODE>), dim3(grid8), dim3(64 * NWV), lds_bytes, stream, a);
    return hipGetLastError();
}

template <int NT, int NWV, bool LEAN>
hipError_t wave2_launch_variant(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info) {
    if (info) {
        info->grid = grid; info->block = 64 * NWV; info->ntile = NT;
        info->lds_bytes = wave_mode(a) == 1 ? wave2_lds_bytes<NT, NWV, LEAN, 1>(a) : wave2_lds_bytes<NT, NWV, LEAN, 0>(a);
    }
    const int grid8 = 8 * ((grid + 7) / 8);
    switch (wave_mode(a)) {
        case 0: return wave2_launch_mode<NT, NWV, LEAN, 0>(a, grid8, stream);
        case 1: return wave2_launch_mode<NT, NWV, LEAN, 1>(a, grid8, stream);
        default: return hipErrorNotSupported;        // launch_one keeps such batches on the multi-wave kernel
    }
}

}  // namespace
