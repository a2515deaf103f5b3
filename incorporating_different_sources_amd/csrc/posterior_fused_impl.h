#pragma once
// posterior_fused_impl.h - fused per-window posterior kernel for gfx950 (MI355X), register-tile path
// (k <= 239).  Instantiated once per tile count by posterior_fused_nt.hip.
//
// One workgroup (NW wavefronts of 64 lanes) owns one rolling window and keeps the whole upper triangle
// of the (k+1) x (k+1) bordered posterior matrix
//
//        [ S1   b ]      S1 = n0 m/(m-1) (Y-Ybar)'(Y-Ybar) + X'X        (ref:317-333, 180, 358)
//        [ b'   * ]      b  = c S0 w0 + t                               (ref:489 right-hand side)
//
// in v_mfma_f64_16x16x4_f64 accumulator tiles (16x16, NT tiles per side, tile (I,J), I<=J, lives in
// ONE wave's registers from the first Gram MFMA to the last back-substitution product).
// ref:LINE cites /root/reference/src/portfolio_calculations.py.
//
//   phase A  column means of the intraday rows (two-pass centring like DataFrame.cov, ref:317):
//            16-lane coalesced row segments, four iterations of loads in flight
//   phase B  Gram of the centred intraday rows, staged 16 rows at a time global -> registers -> LDS
//            (double-buffered; the next chunk's RAW loads are issued before the MFMA block and consumed
//            after it); column k of the staged rows carries z_r = (y_r - ybar).w0, so the same MFMAs
//            yield C w0 (border column) and w0'C w0
//   phase C  q0, c (ref:415-418); scale tiles by s = n0 m/(m-1), the border column by c s
//   phase D  Gram of the daily excess returns on top (ref:180); column k of the staged rows is 1, so
//            the border column accumulates t = X'1 (ref:222)
//   phase E  (Jeffreys) J = T - t t'/N (ref:600-601)
//   phase F  blocked upper Cholesky S1 = R'R, 16-row block steps: the diagonal tile goes to wave 0
//            through LDS; wave 0 eliminates it (column per lane, 16 pivots, multipliers by v_readlane,
//            rsqrt with look-ahead) together with 16 identity columns => M = R_jj^-T; every tile of the
//            block row becomes R_jJ = M A_jJ by MFMA with the tile's OWN accumulator registers as the B
//            operand; the trailing tiles are updated by MFMA from the LDS image of the block row.  The
//            border column comes out as y = R^-T b: the forward substitution is free.
//   phase G  q1 = y'y (= w1'S1 w1, ref:574), blocked back substitution R w1 = y with the R tiles
//            still in registers (DPP row reductions, partial sums added in fixed order)
//   phase H  weights = (n1+k+2) w1 / (n1-q1) / gamma (ref:572-575, 836) or w/gamma (ref:849)
//
// HBM traffic per window is the algorithmic minimum: each panel row of the window is read once
// (plus once more for the intraday means, mostly from L2 / Infinity Cache), k weights are written.
//
// Notes for whoever edits this file (each cost a measurable factor on MI355X):
//  * tile coordinates must be compile-time constants (for_tiles / static_for) and the whole window
//    body is instantiated per wave index (TP_WAVE_SPECIALISE): otherwise the accumulator array is
//    indexed through pointer-phis and lands in scratch;
//  * build with -mllvm -sink-common-insts=false for the same reason;
//  * nothing may consume a prefetched panel value before the MFMA block (load_chunk), and the
//    col_idx select must sit OUTSIDE the load loop: either mistake serialises the loads;
//  * lane constants are re-derived per phase from a laundered thread id (fresh()): CSE across phases
//    otherwise keeps ~100 registers alive for the whole kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "posterior_kernels.h"

// (included by posterior_fused_nt.hip, once per tile count NT)

typedef double d4 __attribute__((ext_vector_type(4)));

// how many 4-row MFMA k-steps of a staged chunk are unrolled together (operand reads in flight)
#ifndef TP_KSTEP_UNROLL
#define TP_KSTEP_UNROLL 1
#endif
constexpr int tp_kstep_unroll = TP_KSTEP_UNROLL;

// Diagnostic build only (make TP_STAMP=1): per-window s_memtime stamps at the phase boundaries, written
// to a buffer of their own (never into an output).  The product build compiles none of this.
#ifdef TP_STAMP
#define TP_LOOPSTAMP_PTR (A.stamps ? A.stamps + (w - A.w_first) * 40 + 8 : nullptr)
#define TP_MARK(slot) do { if (A.stamps && (tid0 == 0)) A.stamps[(w - A.w_first) * 40 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TP_LOOPSTAMP_PTR nullptr
#define TP_MARK(slot) do { } while (0)
#endif

namespace {

template <int NT_, int NW_>
struct Cfg {
    static constexpr int NT = NT_;                       // 16-column tiles per side (border column included)
    static constexpr int NW = NW_;                       // wavefronts per workgroup
    static constexpr int KP = 16 * NT;                   // padded column count
    static constexpr int LDX = (KP % 32 == 16) ? KP : KP + 16;  // LDS row stride (doubles); LDX % 32 == 16
                                                         // puts rows r, r+1 on opposite bank halves for ds_read_b64
    static constexpr int NTILES = NT * (NT + 1) / 2;
    static constexpr int SLOTS = (NTILES + NW - 1) / NW; // tiles per wave
    static constexpr int NTHREADS = 64 * NW;
    static constexpr int CH = (NW == 8) ? 32 : 16;       // staged rows per chunk
    static constexpr int ROWS_PER_PASS = NTHREADS / 16;  // 16 threads per staged row
    static constexpr int PASSES = CH / ROWS_PER_PASS;
    static_assert(PASSES >= 1, "bad staging geometry");
    // LDS carve (doubles)
    static constexpr int STAGE = CH * LDX;               // one staging buffer; buffer 0 doubles as the block-row
                                                         // image RB, buffer 1 as the R_jj^-T store (NT*256 <= 16*LDX)
    static constexpr int OFF_STAGE0 = 0;
    static constexpr int OFF_STAGE1 = STAGE;
    static constexpr int OFF_PART = 2 * STAGE;           // [NTILES][16] partial products of the back substitution
    static constexpr int OFF_YBAR = OFF_PART + NTILES * 16;  // [KP] intraday column means / Jeffreys t
    static constexpr int OFF_W0 = OFF_YBAR + KP;         // [KP] prior weights, zero padded
    static constexpr int OFF_YVEC = OFF_W0 + KP;         // [KP] y = R^-T b
    static constexpr int OFF_WVEC = OFF_YVEC + KP;       // [KP] solution
    static constexpr int OFF_DIAG = OFF_WVEC + KP;       // 2 x [16][16] diagonal tile handed to the eliminating wave
    static constexpr int OFF_SCAL = OFF_DIAG + 512;      // [8] scalars: 0 = z'z, 1 = not-positive-definite flag
    static constexpr int OFF_COFF = OFF_SCAL + 8;        // [KP] uint32: byte offsets 8*col_idx[c] of the gathered columns
    // [16][16] identity tile: the eliminating wave loads [A_jj | I] with ONE address select.  During the
    // factorisation the partial-product area of the back substitution is free and holds it when it is large
    // enough (NT >= 6; NT = 7 sits at the 40 KiB limit of 4 workgroups per CU), else it gets space of its own.
    static constexpr int IDT_DOUBLES = 256;              // identity tile only; lanes >= 32 re-read it (their values are never used)
    static constexpr bool IDT_IN_PART = NTILES * 16 >= IDT_DOUBLES;
    static constexpr int OFF_IDT = IDT_IN_PART ? OFF_PART : OFF_COFF + KP / 2;
    static constexpr int LDS_DOUBLES = OFF_COFF + KP / 2 + (IDT_IN_PART ? 0 : IDT_DOUBLES);
    static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
};

__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// rotate right by N lanes inside each row of 16 lanes (DPP row_ror:N) - no LDS crossbar involved
template <int N>
__device__ __forceinline__ double dpp_row_ror(double v) {
    // every lane of a row_ror has a source lane: no "old" value is needed (mov_dpp leaves it undefined)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x120 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x120 + N, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// sum over the 16 lanes that share lane>>4 (one MFMA row group); every lane gets the sum
__device__ __forceinline__ double rowgroup_sum16(double v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    return v;
}

// An identity the optimiser cannot see through.  Every phase re-derives its lane constants (row/column
// of the lane, LDS addresses, masks) from a laundered thread id, so that common-subexpression
// elimination and loop-invariant hoisting cannot stretch those values' live ranges over the whole
// kernel - that, not the accumulators, is what drove the register count to the 256 ceiling.
__device__ __forceinline__ int fresh(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// 1/d: v_rcp_f64 seed + two Newton steps (5 dependent DP operations)
__device__ __forceinline__ double rcp_nr(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// 1/sqrt(d): v_rsq_f64 seed (about 2^-24 relative) + two Newton steps; a non-positive or NaN d
// gives NaN/Inf, which the caller reports as "not positive definite"
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    double e = fma(-(d * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(d * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}

// 1/sqrt(d) from the v_rsq_f64 seed (23 good bits) by ONE third-order step, y (1 + e/2 + 3 e^2/8) with
// e = 1 - d y^2: the error term is O(e^3) ~ 2^-68.  Six instructions on a chain of five (rsqrt_nr: nine on
// seven) - the pivot chain of the factorisation pays for both.
__device__ __forceinline__ double rsqrt_cubic(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * y), y, 1.0);
    const double u = fma(e, 0.375, 0.5);
    return fma(y * e, u, y);
}

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- compile-time tile bookkeeping -----------------------------------------------------------
// Upper-triangle tiles (I <= J) are numbered row-major; wave WV of a workgroup owns tiles
// t = s*NW + WV (slot s).  Everything below is constexpr so that tile coordinates fold into
// immediates and no per-slot bookkeeping lives in registers.
template <int NT>
constexpr int tile_I(int t) { int i = 0, rem = t; while (rem >= NT - i) { rem -= NT - i; ++i; } return i; }
template <int NT>
constexpr int tile_J(int t) { int i = 0, rem = t; while (rem >= NT - i) { rem -= NT - i; ++i; } return i + rem; }

template <int V> using ic = std::integral_constant<int, V>;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(ic<B>{}); static_for<B + 1, E>(f); }
}

// run f(ic<wv>) with the wave index as a compile-time constant (wv is wave-uniform: scalar branches)
template <int NW, class F>
__device__ __forceinline__ void wave_dispatch(int wv, F&& f) {
    if constexpr (NW == 1) { f(ic<0>{}); }
    else if constexpr (NW == 2) { if (wv == 0) f(ic<0>{}); else f(ic<1>{}); }
    else if constexpr (NW == 4) {
        if (wv < 2) { if (wv == 0) f(ic<0>{}); else f(ic<1>{}); }
        else { if (wv == 2) f(ic<2>{}); else f(ic<3>{}); }
    } else {
        static_assert(NW == 8, "unsupported wave count");
        if (wv < 4) {
            if (wv < 2) { if (wv == 0) f(ic<0>{}); else f(ic<1>{}); }
            else { if (wv == 2) f(ic<2>{}); else f(ic<3>{}); }
        } else {
            if (wv < 6) { if (wv == 4) f(ic<4>{}); else f(ic<5>{}); }
            else { if (wv == 6) f(ic<6>{}); else f(ic<7>{}); }
        }
    }
}

// FIX >= 0: the caller is already specialised for wave FIX (no branch); FIX < 0: branch on wv
template <int NW, int FIX, class F>
__device__ __forceinline__ void wave_sel(int wv, F&& f) {
    if constexpr (FIX >= 0) f(ic<FIX>{});
    else wave_dispatch<NW>(wv, f);
}

// for every tile slot of wave WV: f(ic<s>, ic<I>, ic<J>)
template <class C, int WV, class F>
__device__ __forceinline__ void for_tiles(F&& f) {
    static_for<0, C::SLOTS>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        constexpr int t = s * C::NW + WV;
        if constexpr (t < C::NTILES) f(sc, ic<tile_I<C::NT>(t)>{}, ic<tile_J<C::NT>(t)>{});
    });
}

// Which rows / columns a phase stages.
struct RowSource {
    const double* base;     // panel base
    int ld;                 // leading dimension (doubles)
    const int* ridx;        // explicit rows of this window, or nullptr
    long long first;        // first row (contiguous mode)
    const double* sub_row;  // per-row subtrahend (rf_adj) or nullptr
    int count;              // rows of this window
    bool off32;             // every byte offset from `base` (explicit rows) / from the window's first row fits 32 bits
    int count0 = 0x7fffffff; // lean daily path: staged rows r >= count0 are panel rows first + r + jump (a second row range)
    int jump = 0;
};

// ---- staging.  VALU instructions are the scarce resource of this kernel: on gfx950 a wavefront streaming
// fp64 MFMAs lets the SIMD's other wavefronts issue about ONE vector instruction per MFMA (tools/
// coexec_probe.hip), so every v_* spent on addresses, masks or selects is paid in MFMA time.  Hence:
//  * addresses are ONE 32-bit offset per row (v_mad_u32_u24) on a uniform 64-bit base (scalar registers);
//    the 16-column groups are immediate offsets, gathered columns come from a byte-offset table in LDS;
//  * row masks exist only in the ragged last chunk (`full` is wave-uniform), column masks only in the
//    last 16-column group.

// Issue the global loads of one staged chunk: RAW values only.  Nothing here may consume a loaded
// value (no subtraction, no select): any use would make the compiler wait for the loads right here
// and the prefetch under the MFMA block would be lost (it was: 4.4 k cycles per chunk).  Masking,
// centring and the risk-free subtraction happen in store_chunk, after the MFMAs.
template <class C, bool HF>
__device__ __forceinline__ void load_chunk(const RowSource& src, const int* __restrict__ cols,
                                           const unsigned* __restrict__ coff, int k, int chunk, int tid, bool full,
                                           double (&v)[C::PASSES][C::NT], double (&sub)[C::PASSES]) {
    const int cb = tid & 15;
    constexpr int kI = C::NT - 1;
    if (src.off32) {
        // uniform base: the window's first row (contiguous rows) or the panel (explicit rows)
        const char* ub = (const char*)(src.ridx ? src.base : src.base + src.first * (long long)src.ld);
        const unsigned ld8 = (unsigned)src.ld * 8u;
        unsigned co[C::NT];
        if (cols) {
#pragma unroll
            for (int i = 0; i < C::NT; ++i) co[i] = coff[cb + 16 * i];       // padding columns repeat column k-1
        } else {
            const int cl = cb + 16 * kI;
            co[kI] = 8u * (unsigned)(cl < k ? cl : k - 1);                   // k >= 16 (NT-1): only the last group clamps
        }
#pragma unroll
        for (int ps = 0; ps < C::PASSES; ++ps) {
            const int r = chunk * C::CH + ps * C::ROWS_PER_PASS + (tid >> 4);
            const int rc = (full || r < src.count) ? r : src.count - 1;      // clamp (count >= 1 is validated on the host)
            const unsigned row = src.ridx ? (unsigned)src.ridx[rc] : (unsigned)rc;
            sub[ps] = 0.0;
            if (!HF && src.sub_row) sub[ps] = src.sub_row[rc];
            const unsigned ro = __umul24(row, ld8);          // rows, 8 ld < 2^24 (checked on the host)
            if (cols) {
#pragma unroll
                for (int i = 0; i < C::NT; ++i) v[ps][i] = *(const double*)(ub + (size_t)(ro + co[i]));
            } else {
                const unsigned o = ro + 8u * (unsigned)cb;
#pragma unroll
                for (int i = 0; i < kI; ++i) v[ps][i] = *(const double*)(ub + (size_t)o + 128 * i);
                v[ps][kI] = *(const double*)(ub + (size_t)(ro + co[kI]));
            }
        }
        return;
    }
    // explicit rows in a panel of 4 GiB or more: 64-bit addresses
    int ci[C::NT];
    if (cols) {
#pragma unroll
        for (int i = 0; i < C::NT; ++i) {
            const int c = cb + 16 * i;
            ci[i] = cols[((i < kI) || (c < k)) ? c : k - 1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < C::NT; ++i) {
            const int c = cb + 16 * i;
            ci[i] = ((i < kI) || (c < k)) ? c : k - 1;
        }
    }
#pragma unroll
    for (int ps = 0; ps < C::PASSES; ++ps) {
        const int r = chunk * C::CH + ps * C::ROWS_PER_PASS + (tid >> 4);
        const int rc = (r < src.count) ? r : src.count - 1;
        const long long row = src.ridx ? (long long)src.ridx[rc] : src.first + rc;
        sub[ps] = 0.0;
        if (!HF && src.sub_row) sub[ps] = src.sub_row[rc];
        const double* p = src.base + row * (long long)src.ld;
#pragma unroll
        for (int i = 0; i < C::NT; ++i) v[ps][i] = p[ci[i]];
    }
}

// Finish the staged values in registers (centre / z column for HF, ones column for daily) and write
// them to the LDS staging buffer.  `full`: every row of the chunk is a real row (wave-uniform).
template <class C, bool HF>
__device__ __forceinline__ void store_chunk(double* __restrict__ buf, const double* __restrict__ lds, int k,
                                            int chunk, int count, int tid, bool full, bool has_sub,
                                            double (&v)[C::PASSES][C::NT], const double (&sub)[C::PASSES]) {
    const int cb = tid & 15;
    constexpr int kI = C::NT - 1;            // the border column k always lies in the last 16-column group
    const int kc = k - 16 * kI;
    const bool cvl = cb < kc;                // column validity in the last group
    // intraday rows, one-pass scatter: column k+1 (when the last group has a spare column) carries ones, so that
    // the MFMAs also produce the column sums of the shifted rows (phase C)
    const double hfone = (HF && kc < 15 && cb == kc + 1) ? 1.0 : 0.0;
#pragma unroll
    for (int ps = 0; ps < C::PASSES; ++ps) {
        const int rl = ps * C::ROWS_PER_PASS + (tid >> 4);
        if (HF) {
            double z = 0.0;
#pragma unroll
            for (int i = 0; i < C::NT; ++i) {
                const int c = cb + 16 * i;
                v[ps][i] -= lds[C::OFF_YBAR + c];                            // ybar / shift row, w0 are zero-padded
                if (i == kI) v[ps][i] = cvl ? v[ps][i] : hfone;
            }
            if (!full) {
                const bool rv = chunk * C::CH + rl < count;
#pragma unroll
                for (int i = 0; i < C::NT; ++i) v[ps][i] = rv ? v[ps][i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < C::NT; ++i) z = fma(v[ps][i], lds[C::OFF_W0 + cb + 16 * i], z);
            z = rowgroup_sum16(z);
            if (cb == kc) v[ps][kI] = z;                 // border column: z_r = (y_r - ybar).w0
        } else {
            if (has_sub) {
#pragma unroll
                for (int i = 0; i < C::NT; ++i) v[ps][i] -= sub[ps];         // ref:57
            }
            v[ps][kI] = cvl ? v[ps][kI] : ((cb == kc) ? 1.0 : 0.0);          // border column: ones -> t = X'1
            if (!full) {
                const bool rv = chunk * C::CH + rl < count;
#pragma unroll
                for (int i = 0; i < C::NT; ++i) v[ps][i] = rv ? v[ps][i] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < C::NT; ++i) buf[rl * C::LDX + cb + 16 * i] = v[ps][i];
    }
}

// acc(I,J) (+/-)= rows[:, I]' rows[:, J] over NROWS staged rows (NROWS/4 MFMA k-steps), for the
// tiles of wave WV that satisfy `pick(I,J)`.  `lanebase` = buf + fq*LDX + fr, so every operand is
// one ds_read_b64 at an immediate offset.
template <class C, int WV, int NROWS, bool NEG, class Pick>
__device__ __forceinline__ void mfma_tiles(const double* __restrict__ lanebase, d4 (&acc)[C::SLOTS], Pick pick,
                                           int ksteps = NROWS / 4) {
#pragma clang loop unroll_count(tp_kstep_unroll)
    for (int s4 = 0; s4 < ksteps; ++s4) {
        for_tiles<C, WV>([&](auto sc, auto Ic, auto Jc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
            if (pick(I, J)) {
                const double a = lanebase[4 * s4 * C::LDX + 16 * I];
                const double b = lanebase[4 * s4 * C::LDX + 16 * J];
                // f64 MFMAs reuse the BLGP field as negate bits (bit 0: A): acc -= a'b without a vector negate
                acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[s], 0, 0, NEG ? 1 : 0);
            }
        });
    }
}

// The Gram phase for contiguous rows, ungathered columns and 32-bit offsets (the layout of a rolling
// window over one shared panel): the same loads, the same arithmetic in the same order as gram_phase below,
// with the per-chunk vector work cut to the bone - four loop-carried lane values (row offset, its clamp,
// the last column group's offset, the LDS write address) instead of re-deriving everything from the thread
// id, the row clamp as ONE v_min, masks only in the ragged last chunk.
struct NoChunkHook { __device__ __forceinline__ void operator()(int) const {} };

// `before_chunk(ch)` runs before the MFMAs of chunk ch (the prefix kernel stores the running sums there).
template <class C, bool HF, int FIX, class Hook = NoChunkHook>
__device__ __forceinline__ void gram_phase_lean(const RowSource& src, int k, double* lds, int tid0, int wv,
                                                d4 (&acc)[C::SLOTS], Hook before_chunk = Hook{}) {
    constexpr int kI = C::NT - 1;
    const int nchunks = (src.count + C::CH - 1) / C::CH;
    const bool has_sub = !HF && src.sub_row != nullptr;
    const char* ub = (const char*)(src.base + src.first * (long long)src.ld);      // uniform: scalar registers
    const unsigned ld8 = (unsigned)src.ld * 8u;
    const unsigned step = (unsigned)C::ROWS_PER_PASS * ld8;
    const int kc = k - 16 * kI;
    const int tid = fresh(tid0);
    const int cb = tid & 15;
    const int cl = cb + 16 * kI;
    unsigned voff = __umul24((unsigned)(tid >> 4), ld8) + 8u * (unsigned)cb;       // (row, column group 0) of this lane
    const unsigned vmax = __umul24((unsigned)(src.count - 1), ld8) + 8u * (unsigned)cb;
    const unsigned dlast = 8u * (unsigned)((cl < k ? cl : k - 1) - cb);            // k >= 16 (NT-1): only the last group clamps
    const unsigned wlds = (unsigned)((tid >> 4) * C::LDX + cb);
    const bool cvl = cb < kc;
    // daily rows: ones in the border column (t = X'1); intraday rows: ones in the spare column k+1 (column sums
    // of the shifted rows for the one-pass scatter, phase C) when the last column group has one
    const double border = (!HF && cb == kc) ? 1.0 : ((HF && kc < 15 && cb == kc + 1) ? 1.0 : 0.0);
    double v[C::PASSES][C::NT];
    double sub[C::PASSES];
    int row = tid >> 4;                       // this lane's row in the next chunk to be loaded (pass 0)

    auto load = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < C::PASSES; ++ps) {
            unsigned o = voff < vmax ? voff : vmax;                                // rows past the end re-read the last row
            if constexpr (!HF) {                                                   // two row ranges: from the row number
                const int r = row + ps * C::ROWS_PER_PASS;
                const int rc = r < src.count ? r : src.count - 1;
                o = __umul24((unsigned)(rc + (rc >= src.count0 ? src.jump : 0)), ld8) + 8u * (unsigned)cb;
            }
            sub[ps] = 0.0;
            if (has_sub) { const int r = row + ps * C::ROWS_PER_PASS; sub[ps] = src.sub_row[r < src.count ? r : src.count - 1]; }
#pragma unroll
            for (int i = 0; i < kI; ++i) v[ps][i] = *(const double*)(ub + (size_t)o + 128 * i);
            v[ps][kI] = *(const double*)(ub + (size_t)(o + dlast));
            voff += step;
        }
        row += C::CH;
    };
    auto store = [&](double* __restrict__ buf, int chunk) __attribute__((always_inline)) {
        const bool full = (chunk + 1) * C::CH <= src.count;
#pragma unroll
        for (int ps = 0; ps < C::PASSES; ++ps) {
            if (HF) {
#pragma unroll
                for (int i = 0; i < C::NT; ++i) v[ps][i] -= lds[C::OFF_YBAR + cb + 16 * i];    // ybar / shift row, w0: zero-padded
                v[ps][kI] = cvl ? v[ps][kI] : border;
            } else {
                if (has_sub) {
#pragma unroll
                    for (int i = 0; i < C::NT; ++i) v[ps][i] -= sub[ps];                       // ref:57
                }
                v[ps][kI] = cvl ? v[ps][kI] : border;                                        // ones column -> t = X'1
            }
            if (!full) {
                const bool rv = chunk * C::CH + ps * C::ROWS_PER_PASS + (tid >> 4) < src.count;
#pragma unroll
                for (int i = 0; i < C::NT; ++i) v[ps][i] = rv ? v[ps][i] : 0.0;
            }
            if (HF) {
                double z = 0.0;
#pragma unroll
                for (int i = 0; i < C::NT; ++i) z = fma(v[ps][i], lds[C::OFF_W0 + cb + 16 * i], z);
                z = rowgroup_sum16(z);
                if (cb == kc) v[ps][kI] = z;                                                 // z_r = (y_r - ybar).w0
            }
#pragma unroll
            for (int i = 0; i < C::NT; ++i) buf[wlds + ps * C::ROWS_PER_PASS * C::LDX + 16 * i] = v[ps][i];
        }
    };

    if (nchunks > 0) {
        load();
        store(lds + C::OFF_STAGE0, 0);
    }
    __syncthreads();
    // Accumulator tiles that were spilled across the previous phase come back by scratch loads; settle them
    // HERE, or the counter pass puts an s_waitcnt vmcnt(0) into the MFMA loop, where it would also wait for
    // the prefetch of the next chunk on every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    const int fr = tid & 15, fq = (tid & 63) >> 4;
#pragma nounroll
    for (int ch = 0; ch < nchunks; ++ch) {
        double* cur = lds + ((ch & 1) ? C::OFF_STAGE1 : C::OFF_STAGE0);
        double* nxt = lds + ((ch & 1) ? C::OFF_STAGE0 : C::OFF_STAGE1);
        const bool more = ch + 1 < nchunks;
        if (more) load();                             // global loads in flight under the MFMAs
        __builtin_amdgcn_sched_barrier(0);            // the scheduler must not sink these loads below the MFMA block
        const double* lanebase = cur + fq * C::LDX + fr;
        before_chunk(ch);
        // the ragged last chunk runs only the 4-row k-steps that hold rows (the rest of the staged chunk is zeros)
        const int rows_here = src.count - ch * C::CH;
        const int ksteps = rows_here >= C::CH ? C::CH / 4 : (rows_here + 3) >> 2;
        wave_sel<C::NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            mfma_tiles<C, decltype(wc)::value, C::CH, false>(lanebase, acc, [](int, int) { return true; }, ksteps);
        });
        if (more) store(nxt, ch + 1);
        __syncthreads();
    }
}

template <class C, bool HF, int FIX>
__device__ __forceinline__ void gram_phase(const RowSource& src, const int* __restrict__ cols, int k, double* lds,
                                           int tid0, int wv, d4 (&acc)[C::SLOTS], long long* loopstamps) {
#ifdef TP_STAMP
    // diagnostic: time spent in the four segments of a chunk iteration (issue loads | MFMA block |
    // finish + LDS write | barrier), summed over the phase, for the wave this body is specialised for
    long long seg[4] = {0, 0, 0, 0};
    long long tprev = 0;
#define TP_LOOPSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const long long tn = (long long)__builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); \
        if ((i) > 0) seg[((i) > 0) ? (i) - 1 : 0] += tn - tprev; tprev = tn; } while (0)
#else
#define TP_LOOPSTAMP(i) do { } while (0)
#endif
    const int nchunks = (src.count + C::CH - 1) / C::CH;
    double v[C::PASSES][C::NT];
    double sub[C::PASSES];
    const unsigned* coff = (const unsigned*)(lds + C::OFF_COFF);
    const bool has_sub = !HF && src.sub_row != nullptr;
    if (nchunks > 0) {
        const int tid = fresh(tid0);
        const bool full = C::CH <= src.count;
        load_chunk<C, HF>(src, cols, coff, k, 0, tid, full, v, sub);
        store_chunk<C, HF>(lds + C::OFF_STAGE0, lds, k, 0, src.count, tid, full, has_sub, v, sub);
    }
    __syncthreads();
#pragma nounroll
    for (int ch = 0; ch < nchunks; ++ch) {
        const int tid = fresh(tid0);                  // per-chunk lane constants: nothing hoisted out of the loop
        const int fr = tid & 15, fq = (tid & 63) >> 4;
        double* cur = lds + ((ch & 1) ? C::OFF_STAGE1 : C::OFF_STAGE0);
        double* nxt = lds + ((ch & 1) ? C::OFF_STAGE0 : C::OFF_STAGE1);
        const bool more = ch + 1 < nchunks;
        const bool full = (ch + 2) * C::CH <= src.count;                       // chunk ch+1 has no ragged rows
        TP_LOOPSTAMP(0);
        if (more) load_chunk<C, HF>(src, cols, coff, k, ch + 1, tid, full, v, sub);   // global loads in flight under the MFMAs
        __builtin_amdgcn_sched_barrier(0);   // the scheduler must not sink these loads below the MFMA block
        TP_LOOPSTAMP(1);
        const double* lanebase = cur + fq * C::LDX + fr;
        wave_sel<C::NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            mfma_tiles<C, decltype(wc)::value, C::CH, false>(lanebase, acc, [](int, int) { return true; });
        });
        TP_LOOPSTAMP(2);
        if (more) store_chunk<C, HF>(nxt, lds, k, ch + 1, src.count, tid, full, has_sub, v, sub);
        TP_LOOPSTAMP(3);
        __syncthreads();
        TP_LOOPSTAMP(4);
    }
#ifdef TP_STAMP
    if (loopstamps && (tid0 & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) loopstamps[4 * wv + i] = seg[i];
    }
#endif
#undef TP_LOOPSTAMP
}


// LEAN: the batch is in the contiguous layout (no row / column index arrays, 32-bit offsets) - a kernel of its
// own, chosen on the host, so that neither staging variant's registers weigh on the other's allocation (one
// kernel carrying both spilled the staged values INSIDE the generic chunk loop: 2.06 instead of 1.43 ms).
template <int NT, int NW, int FIX, bool LEAN>
__device__ __forceinline__ void window_body(const tp_kargs_t& A, double* lds, const int tid0, const int wv) {
    using C = Cfg<NT, NW>;
// lane constants of one phase: tid, lane, MFMA fragment column fr and row group fq, border-column masks
#define TP_LANE_CONSTANTS() \
    const int tid = fresh(tid0); const int lane = tid & 63; const int fr = lane & 15; const int fq = lane >> 4; \
    const bool colv = fr < kc; (void)lane; (void)fq; (void)colv
    const int k = A.k;
    // NT = ceil((k+1)/16): the border column k lies in the last tile column, always
    __builtin_assume(k >= 16 * (NT - 1));
    __builtin_assume(k <= 16 * NT - 1);
    constexpr int kI = NT - 1;                 // tile column of the border column
    const int kc = k - 16 * kI;                // its local column (0..15) = number of real rows/cols in block kI
    const int NTB = (kc == 0) ? NT - 1 : NT;   // block rows that hold pivots
    // One workgroup per window, XCD-aware: MI355X deals consecutive workgroup ids round-robin to its 8 XCDs (each
    // with an L2 of its own), so id -> (xcd = id % 8, slot = id / 8) -> window xcd * ceil(count / 8) + slot gives
    // every XCD ONE contiguous range of windows, in order.  Neighbouring rolling windows share their panel rows and -
    // 16 at a time - the very same slots of the shared Gram prefixes (57 KB each at k = 100): with the plain
    // id -> window map those 16 windows sat on 8 different XCDs and the prefix reads (2-3 slots per window, the
    // largest data stream of the kernel since round 2) came from the Infinity Cache instead of L2.
    const long long per_xcd = (A.w_count + 7) >> 3;
    const long long wl = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= per_xcd || wl >= A.w_count) return;
    const long long w = A.w_first + wl;

    const int* cols = A.col_idx ? A.col_idx + w * k : nullptr;
    d4 acc[C::SLOTS];
    unsigned* coff = (unsigned*)(lds + C::OFF_COFF);
    if (cols) {
        // byte offsets of the gathered columns, once per window; padding columns repeat column k-1 (their
        // values are masked in store_chunk) so that every load address is a real element
        for (int c = tid0; c < C::KP; c += C::NTHREADS) coff[c] = 8u * (unsigned)cols[c < k ? c : k - 1];
        __syncthreads();
    }

    double n0 = 0.0, cc = 0.0, q0 = 0.0;
    const bool conj = A.strategy == 0;
    if (tid0 == 0) lds[C::OFF_SCAL + 1] = 0.0;   // not-positive-definite flag (barriers of the Gram phase publish it)
    // debug read-back of one window's matrix (tp_batch_download_matrix): 1 = prior scatter S0 and
    // c S0 w0, 2 = canonical statistics T and t only, 3 = posterior S1 (or Jeffreys J) and rhs
    const int dbg = (A.dbg_S1 != nullptr && w == A.dbg_w) ? A.dbg_mode : 0;

    // dump the current bordered matrix: [k x k] symmetric part, then the border column
    auto dump_matrix = [&]() __attribute__((always_inline)) {
        TP_LANE_CONSTANTS();
        wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                const int gj = 16 * J + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = 16 * I + fq + 4 * r;
                    if (gi < k && gj < k) {
                        A.dbg_S1[(long long)gi * k + gj] = acc[s][r];
                        A.dbg_S1[(long long)gj * k + gi] = acc[s][r];
                    }
                    if (gi < k && gj == k) A.dbg_S1[(long long)k * k + gi] = acc[s][r];
                }
            });
        });
    };

    if (conj && dbg != 2) {
        n0 = A.n0[w];
        RowSource hs;
        hs.base = A.hf_panel; hs.ld = A.hf_ld;
        hs.ridx = A.hf_row_idx ? A.hf_row_idx + w * (long long)A.m : nullptr;
        hs.first = A.hf_start ? A.hf_start[w] : 0;
        hs.sub_row = nullptr;
        hs.count = A.hf_count ? A.hf_count[w] : A.m;
        hs.off32 = (A.hf_off32 & (hs.ridx ? 1 : 2)) != 0;
        TP_MARK(0);
        // ---- phase A.  The centred scatter of ref:317 (DataFrame.cov centres first) WITHOUT a pass of its own over
        // the intraday rows: with the rows shifted by the window's first row s (the shifted-data form of the
        // covariance: exact for constant columns, and |ybar - s| is of the order of one standard deviation, so the
        // correction below cancels about one bit),
        //     sum (y - ybar)(y - ybar)' = sum (y - s)(y - s)' - (1/m) t t',   t = sum (y - s),
        // and t comes out of the SAME MFMAs through a column of ones in the spare column k+1 (phase C applies the
        // rank-one term).  The means pass cost two HBM round trips per window (7 % of its lifetime) and made the
        // intraday panel - the one input that really streams from HBM - be read twice.  Universe sizes with
        // k+1 a multiple of 16 have no spare column and keep the two-pass form.
        const bool shifted = kc < 15;
        if (shifted) {
            TP_LANE_CONSTANTS();
            const long long row0 = hs.ridx ? (long long)hs.ridx[0] : hs.first;
            const double* p0 = hs.base + row0 * (long long)hs.ld;
            for (int c = tid; c < C::KP; c += C::NTHREADS) {
                const int cl = c < k ? c : k - 1;
                const double sv = p0[cols ? cols[cl] : cl];
                lds[C::OFF_YBAR + c] = (c < k) ? sv : 0.0;
                lds[C::OFF_W0 + c] = (c < k) ? A.w0[w * k + cl] : 0.0;
            }
            __syncthreads();
        } else {
            // column means in a pass of their own.  Same thread geometry as the staging (16 threads per row, 16-lane
            // coalesced segments): every thread sums its rows, the ROWS_PER_PASS partial rows meet in LDS and are
            // added in row order.
            TP_LANE_CONSTANTS();
            double cs[C::NT];
#pragma unroll
            for (int i = 0; i < C::NT; ++i) cs[i] = 0.0;
            const int cb = tid & 15;
            // same loads as the staging (load_chunk); full chunks need no mask at all (the padding columns'
            // sums are dropped when ybar is written), only the ragged last chunk masks its rows
            const int nfull = hs.count / C::CH;
#pragma unroll 4
            for (int ch = 0; ch < nfull; ++ch) {                            // loads of 4 chunks in flight
                double v[C::PASSES][C::NT];
                double sub[C::PASSES];
                load_chunk<C, true>(hs, cols, coff, k, ch, tid, true, v, sub);
#pragma unroll
                for (int ps = 0; ps < C::PASSES; ++ps)
#pragma unroll
                    for (int i = 0; i < C::NT; ++i) cs[i] += v[ps][i];
            }
            if (nfull * C::CH < hs.count) {
                double v[C::PASSES][C::NT];
                double sub[C::PASSES];
                load_chunk<C, true>(hs, cols, coff, k, nfull, tid, false, v, sub);
#pragma unroll
                for (int ps = 0; ps < C::PASSES; ++ps) {
                    const bool rv = nfull * C::CH + ps * C::ROWS_PER_PASS + (tid >> 4) < hs.count;
#pragma unroll
                    for (int i = 0; i < C::NT; ++i) cs[i] += rv ? v[ps][i] : 0.0;
                }
            }
            double* part = lds + C::OFF_STAGE0;                             // [ROWS_PER_PASS][LDX]
#pragma unroll
            for (int i = 0; i < C::NT; ++i) part[(tid >> 4) * C::LDX + cb + 16 * i] = cs[i];
            __syncthreads();
            for (int c = tid; c < C::KP; c += C::NTHREADS) {
                double sum = 0.0;
#pragma unroll 4
                for (int r = 0; r < C::ROWS_PER_PASS; ++r) sum += part[r * C::LDX + c];
                lds[C::OFF_YBAR + c] = (c < k) ? sum / (double)hs.count : 0.0;
                lds[C::OFF_W0 + c] = (c < k) ? A.w0[w * k + c] : 0.0;
            }
            __syncthreads();
        }
        TP_MARK(1);
        // the accumulators come to life only now: phase A had the whole register file for its loads
        static_for<0, C::SLOTS>([&](auto sc_) __attribute__((always_inline)) {
            acc[decltype(sc_)::value] = d4{0.0, 0.0, 0.0, 0.0};
        });
        // ---- phase B: centred intraday Gram
        if constexpr (LEAN) gram_phase_lean<C, true, FIX>(hs, k, lds, tid0, wv, acc);
        else gram_phase<C, true, FIX>(hs, cols, k, lds, tid0, wv, acc, nullptr);
        TP_MARK(2);
        // ---- phase C: (one-pass form) the rank-one term of the centring; q0, c, scaling (ref:333, 415-418)
        if (shifted) {
            TP_LANE_CONSTANTS();
            // column k+1 holds t_i = sum_r (y_r - s)_i for the asset columns and sum_r u_r in row k (u = (y - s).w0)
            wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
                for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                    constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                    if constexpr (J == kI) {
                        if (fr == kc + 1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                lds[C::OFF_YVEC + 16 * I + fq + 4 * r] = (I < kI || fq + 4 * r <= kc) ? acc[s][r] : 0.0;
                        }
                    }
                });
            });
            __syncthreads();
            const double invm = 1.0 / (double)hs.count;
            wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
                for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                    constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                    const double tj = lds[C::OFF_YVEC + 16 * J + fr];          // zero beyond column k
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double ti = lds[C::OFF_YVEC + 16 * I + fq + 4 * r];
                        acc[s][r] = fma(-(ti * invm), tj, acc[s][r]);
                    }
                });
            });
        }
        {
        TP_LANE_CONSTANTS();
        const double mm = (double)hs.count;
        const double sc = n0 * (mm / (mm - 1.0));
        wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                if constexpr (I == kI && J == kI) {
                    if (fr == kc && fq == (kc & 3)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r == (kc >> 2)) lds[C::OFF_SCAL + 0] = acc[s][r];   // z'z = w0'C w0
                    }
                }
            });
        });
        __syncthreads();
        q0 = sc * lds[C::OFF_SCAL + 0];
        const double a = n0 + k + 2;
        cc = (2 * n0) / (a + sqrt(a * a + 4 * n0 * q0));
        // S0 = sc * C on the real columns, c * sc * C w0 in the border column, zero beyond
        const double fcol = colv ? sc : ((fr == kc) ? cc * sc : 0.0);
        wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                if constexpr (J < kI) {
                    acc[s] *= sc;
                } else if constexpr (I < kI) {
                    acc[s] *= fcol;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[s][r] *= (fq + 4 * r < kc) ? fcol : 0.0;
                }
            });
        });
        }
        if (dbg == 1) { dump_matrix(); return; }
    }

    if (!(conj && dbg != 2)) {
        static_for<0, C::SLOTS>([&](auto sc_) __attribute__((always_inline)) {
            acc[decltype(sc_)::value] = d4{0.0, 0.0, 0.0, 0.0};
        });
    }
    TP_MARK(3);
    // ---- phase D: daily Gram (ref:180) + t in the border column (ref:222)
    {
        RowSource ds;
        ds.base = A.panel; ds.ld = A.panel_ld;
        ds.ridx = A.row_idx ? A.row_idx + w * (long long)A.n_r : nullptr;
        ds.first = A.start ? A.start[w] : 0;
        ds.sub_row = A.rf_adj ? A.rf_adj + w * (long long)A.n_r : nullptr;
        ds.count = A.n_rows ? A.n_rows[w] : A.n_r;
        ds.off32 = (A.panel_off32 & (ds.ridx ? 1 : 2)) != 0;
        if constexpr (LEAN) {
            // Shared Gram sums (block_gram_kernel below + tp_window_sums_kernel): the rows of this window that form whole
            // aligned C::CH-row blocks of the panel come as ONE precomputed block-window sum Q_L[b0] that every window
            // with the same blocks shares; only the < C::CH rows in front of the first whole block and behind the last
            // one go through the MFMAs here.  Rolling windows overlap almost entirely (stride 1: 248 of 249 rows), so
            // the daily Gram of a window costs ~3 k-steps instead of 63.  The decomposition depends on the window's
            // panel rows only - never on which other windows are in the batch.
            constexpr int BLK = C::CH;
            const long long b0 = (ds.first + BLK - 1) / BLK, b1 = (ds.first + ds.count) / BLK;
            // which of the batch's block-window tables holds this window's block count (the host planned them)
            const int Lw = (int)(b1 - b0);
            const int li = Lw == A.winsum_L[0] ? 0 : Lw == A.winsum_L[1] ? 1 : Lw == A.winsum_L[2] ? 2 : Lw == A.winsum_L[3] ? 3 : -1;
            const bool shared = A.winsum != nullptr && Lw > 0 && li >= 0;
            RowSource part = ds;
            if (shared) {              // the rows in front of the first whole block, then the rows behind the last one
                part.count0 = (int)(BLK * b0 - ds.first);
                part.jump = (int)(BLK * b1 - ds.first) - part.count0;
                part.count = part.count0 + (int)(ds.first + ds.count - BLK * b1);
            }
            if (part.count > 0) gram_phase_lean<C, false, FIX>(part, k, lds, tid0, wv, acc);
            if (shared) {
                // ONE table slot: Q_L[b0] = the Gram of the window's L whole blocks (window_sums kernel)
                TP_LANE_CONSTANTS();
                constexpr long long TILE = 4 * 64, SLOT = (long long)C::NTILES * TILE;
                const double* q = A.winsum + ((long long)li * A.prefix_nblk + b0) * SLOT;
                wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
                    constexpr int WV = decltype(wc)::value;
                    for_tiles<C, WV>([&](auto sc_, auto, auto) __attribute__((always_inline)) {
                        constexpr int s = decltype(sc_)::value;
                        constexpr long long t = s * NW + WV;
                        // [slot][tile][2][64 lanes][2]: registers (0,1) and (2,3) of a lane are 16 contiguous bytes
                        typedef double d2 __attribute__((ext_vector_type(2)));
                        const d2* pq = (const d2*)(q + t * TILE) + lane;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const d2 v2 = pq[64 * h];
                            acc[s][2 * h] += v2[0];
                            acc[s][2 * h + 1] += v2[1];
                        }
                    });
                });
            }
        } else {
            gram_phase<C, false, FIX>(ds, cols, k, lds, tid0, wv, acc, TP_LOOPSTAMP_PTR);
        }
    }

    {
    TP_LANE_CONSTANTS();
    // rows >= k of the bordered matrix are never pivots: clear them (they hold 1'X, n_r, ...);
    // Jeffreys: publish t (border column) for the rank-one correction
    wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
        for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
            if constexpr (J == kI) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (I == kI && fq + 4 * r >= kc) acc[s][r] = 0.0;
                    if (!conj && fr == kc) lds[C::OFF_YBAR + 16 * I + fq + 4 * r] = acc[s][r];
                }
            }
        });
    });
    if (dbg == 2) { dump_matrix(); return; }

    if (!conj) {
        // ---- phase E: J = T - t t'/N (ref:600-601); t stays in the border column (ref:606 rhs)
        __syncthreads();
        // center_rows 2 (TP_FLAG_NO_CENTER): the plain Gram matrix; shift (d, e): + d I + e 1 1' on the k x k
        // block (tp_batch_set_shift) - ridge / equicorrelated prior scale matrices such as ref:917, 924
        const double invN = A.center_rows == 2 ? 0.0
                          : 1.0 / (double)(A.center_rows ? (A.n_rows ? A.n_rows[w] : A.n_r) : A.N);
        const double sh_d = A.shift ? A.shift[2 * w] : 0.0;
        const double sh_e = A.shift ? A.shift[2 * w + 1] : 0.0;
        wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                const double tj = lds[C::OFF_YBAR + 16 * J + fr];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double ti = lds[C::OFF_YBAR + 16 * I + fq + 4 * r];
                    const bool on = (J < kI || colv) && (I < kI || fq + 4 * r < kc);
                    const double add = sh_e + ((I == J && fq + 4 * r == fr) ? sh_d : 0.0);
                    acc[s][r] += on ? add - invN * (ti * tj) : 0.0;
                }
            });
        });
    }
    if (A.rhs != nullptr) {
        // caller-supplied right-hand side in place of the border column (tp_batch_set_rhs)
        wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                if constexpr (J == kI) {
                    if (fr == kc) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int gi = 16 * I + fq + 4 * r;
                            const int gl = gi < k ? gi : k - 1;
                            const double x = A.rhs[w * k + gl];
                            acc[s][r] = (gi < k) ? x : 0.0;
                        }
                    }
                }
            });
        });
    }

    if (A.out_rhs != nullptr) {
        // the right-hand side this window is solved for (t = X'1 for Jeffreys: ref:222), for callers
        // that need it next to the solution (tp_batch_download_rhs)
        wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                if constexpr (J == kI) {
                    if (fr == kc) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int gi = 16 * I + fq + 4 * r;
                            if (gi < k) A.out_rhs[w * k + gi] = acc[s][r];
                        }
                    }
                }
            });
        });
    }
    }
    if (dbg == 3) dump_matrix();
#ifdef TP_STAMP
    if (A.phase_limit == 1) {      // diagnostic build only: time the Gram phases alone (keep the tiles alive)
        double sink = 0.0;
        static_for<0, C::SLOTS>([&](auto sc_) __attribute__((always_inline)) {
            constexpr int s = decltype(sc_)::value;
            sink += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3];
        });
        if (sink == 123.456) A.weights[w * k] = sink;
        return;
    }
#endif

    TP_MARK(4);
    // ---- phase F: blocked upper Cholesky S1 = R'R with the border column riding along.
    // Block step j: (1) the owner of diagonal tile (j,j) hands it to wave 0 through LDS;
    // (2) that wave eliminates it (column per lane, 16 pivots, multipliers by v_readlane) together
    // with 16 identity columns, which gives M = R_jj^-T; (3) every tile of block row j becomes
    // R_jJ = M A_jJ by MFMA - the accumulator registers ARE the B operand - and goes to LDS;
    // (4) the trailing tiles are updated from that LDS image by MFMA.  The border column of block row
    // j turns into y_j = (R^-T b)_j on the way: the forward substitution costs nothing extra.
    for (int i = tid0; i < C::IDT_DOUBLES; i += C::NTHREADS)      // identity tile of the elimination (published by the
        lds[C::OFF_IDT + i] = ((i >> 4) == (i & 15)) ? 1.0 : 0.0;  // barrier of block step 0)
    double* RB = lds + C::OFF_STAGE0;          // block row j of R, [16][LDX]
    double* MB = lds + C::OFF_STAGE1;          // M_j transposed, [NTB][16][16]: MB[j][c][i] = M_j[i][c]
#ifdef TP_STAMP
    long long fseg[4] = {0, 0, 0, 0};
    long long ft = 0;
#define TP_FSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const long long tn = (long long)__builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); \
        if ((i) > 0) fseg[((i) > 0) ? (i) - 1 : 0] += tn - ft; ft = tn; } while (0)
#else
#define TP_FSTAMP(i) do { } while (0)
#endif
#pragma nounroll
    for (int j = 0; j < NTB; ++j) {
        TP_LANE_CONSTANTS();
        TP_FSTAMP(0);
        const int npiv = (k - 16 * j < 16) ? (k - 16 * j) : 16;
        double* DG = lds + C::OFF_DIAG + (j & 1) * 256;
        // (1) diagonal tile -> LDS, row-major
        wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                if (I == J && I == j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) DG[(fq + 4 * r) * 16 + fr] = acc[s][r];
                }
            });
        });
        __syncthreads();
        TP_FSTAMP(1);
        // (2) elimination by one wave: lane c < 16 holds column c of the tile, lane 16 + c column c
        //     of the identity.  Pivots past npiv (last block row only) are made inert by selects, so
        //     the 16 steps are one straight-line block.
        // Only wave 0's instantiation carries this code (the window body is specialised per wave).
        if ((FIX == 0 || FIX < 0) && wv == 0) {
            // the pivot chain is the critical path of the window while three waves of this workgroup
            // wait at the barrier: let it win the SIMD's issue arbitration against other workgroups
            __builtin_amdgcn_s_setprio(3);
            const int c16 = lane & 15;
            double a[16];
            // lanes 0-15: the tile's columns; lanes 16-31: the identity's (lanes 32-63 repeat them, unused).  One
            // select on the base, 16 reads at immediate offsets (the obvious `lane < 16 ? DG[..] : (c16 == i)` was
            // compiled into 16 divergent branches with a serialised LDS read each).
            const double* src = (lane < 16) ? DG : (lds + C::OFF_IDT);
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = src[i * 16 + c16];
            bool bad = false;
            // The pivot loop is ISSUE-bound, not latency-bound, and what it is bound by is its count of vector
            // instructions: 2 v_readlane + 1 v_fma_f64 per multiplier, ~12 cycles each on a SIMD shared with three
            // other windows' MFMA streams (in-kernel stamps: 7.5 k cycles per 16 x 16 tile).  Measured in round 2,
            // all slower: multipliers by fp64 DPP (v_fmac_f64 row_newbcast, 440 instead of 615 instructions per
            // tile: +2 %, the fp64 DPP forms issue slowly); a square-root-free scalar pivot chain of 6 dependent steps
            // with the row updates interleaved between the steps (DPP: +25 % in this phase; v_readlane pairs:
            // +33 %, 840 instructions).  Look-ahead: as soon as pivot p has updated row p+1 the next pivot's rsqrt
            // (v_rsq_f64 + one cubic step) is started, so it overlaps the remaining row updates of pivot p.
            double d0 = readlane_d(a[0], 0);
            bad |= !(d0 > 0.0);
            double rinv = rsqrt_cubic(d0);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (p < npiv) {                 // wave-uniform: the last block row stops at its last real pivot
                    a[p] *= rinv;
                    double rinv_next = 1.0;
                    if (p + 1 < 16) {
                        const double s1 = readlane_d(a[p], p + 1);
                        a[p + 1] = fma(-s1, a[p], a[p + 1]);
                        double dn = readlane_d(a[p + 1], p + 1);
                        const bool live = p + 1 < npiv;
                        bad |= live && !(dn > 0.0);
                        dn = live ? dn : 1.0;
                        rinv_next = rsqrt_cubic(dn);
                    }
#pragma unroll
                    for (int i = p + 2; i < 16; ++i) {
                        const double sI = readlane_d(a[p], i);
                        a[i] = fma(-sI, a[p], a[i]);
                    }
                    rinv = rinv_next;
                    // keep the scheduler from hoisting later pivots' v_readlane results (SGPR pairs)
                    // across this point: one pivot's pairs fit the scalar file, all 136 do not
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (lane >= 16 && lane < 32) {
#pragma unroll
                for (int i = 0; i < 16; ++i) MB[j * 256 + c16 * 16 + i] = a[i];
            }
            if (bad && lane == 0) lds[C::OFF_SCAL + 1] = 1.0;
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        TP_FSTAMP(2);
        // (3) block row j: R_jJ = M A_jJ  (A operand M from LDS, B operand = the tile's own registers)
        wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
            for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                if (I == j) {
                    d4 rj = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double mop = MB[j * 256 + (4 * r + fq) * 16 + fr];   // M[fr][4r + fq]
                        rj = __builtin_amdgcn_mfma_f64_16x16x4f64(mop, acc[s][r], rj, 0, 0, 0);
                    }
                    acc[s] = rj;
                    if (I != J) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) RB[(fq + 4 * r) * C::LDX + 16 * J + fr] = rj[r];
                    }
                }
            });
        });
        __syncthreads();
        TP_FSTAMP(3);
        // (4) trailing update A_IJ -= R_jI' R_jJ (I > j) by MFMA from the LDS image of block row j
        {
            const double* lanebase = RB + fq * C::LDX + fr;
            wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
                mfma_tiles<C, decltype(wc)::value, 16, true>(lanebase, acc, [j](int I, int) { return I > j; });
            });
        }
        // no barrier here: RB is rewritten two barriers later, DG alternates between two buffers
        TP_FSTAMP(4);
    }
#ifdef TP_STAMP
    if (A.stamps && (tid0 & 63) == 0 && wv < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) A.stamps[(w - A.w_first) * 40 + 24 + 4 * wv + i] = fseg[i];
    }
#endif
#undef TP_FSTAMP

    TP_MARK(5);
    // ---- phase G: y, q1, back substitution R w = y with the R tiles still in registers
    bool notpd = false;
    double q1 = 0.0;
    {
    TP_LANE_CONSTANTS();
    __syncthreads();      // the last trailing update has read RB; part/yvec/wvec are about to be written
    wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
        for_tiles<C, decltype(wc)::value>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
            if constexpr (J == kI) {
                if (fr == kc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        lds[C::OFF_YVEC + 16 * I + fq + 4 * r] = (I < kI || fq + 4 * r < kc) ? acc[s][r] : 0.0;
                }
            }
        });
    });
    for (int c = tid; c < C::KP; c += C::NTHREADS) lds[C::OFF_WVEC + c] = 0.0;
    __syncthreads();
    notpd = lds[C::OFF_SCAL + 1] != 0.0;
    for (int i = lane; i < k; i += 64) { const double y = lds[C::OFF_YVEC + i]; q1 = fma(y, y, q1); }
    q1 = wave_sum64(q1);      // every wave computes the same value in the same order

    }
#pragma nounroll
    for (int Ib = NTB - 1; Ib >= 0; --Ib) {
        TP_LANE_CONSTANTS();
        const int npiv = (k - 16 * Ib < 16) ? (k - 16 * Ib) : 16;
        if ((FIX == 0 || FIX < 0) && wv == 0) {
            // z = y_Ib - sum_{J > Ib} R_{Ib,J} w_J   (fixed summation order)
            double z = 0.0;
            if (lane < 16) {
                z = lds[C::OFF_YVEC + 16 * Ib + lane];
                int t = 0;
                for (int i = 0; i < Ib; ++i) t += NT - i;   // tile index of (Ib, Ib)
                for (int J = Ib + 1; J < NTB; ++J) z -= lds[C::OFF_PART + (t + (J - Ib)) * 16 + lane];
            }
            // w_Ib = R_jj^-1 z = M' z : w[c] = sum_r M[r][c] z[r]
            double wacc = 0.0;
            const int c16 = lane & 15;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double zr = readlane_d(z, r);
                zr = (r < npiv) ? zr : 0.0;
                wacc = fma(MB[Ib * 256 + c16 * 16 + r], zr, wacc);
            }
            if (lane < npiv) lds[C::OFF_WVEC + 16 * Ib + lane] = wacc;
        }
        __syncthreads();
        if (Ib > 0) {
            // partial products of column block Ib: part[(I,Ib)][row] = sum_c R_{I,Ib}[row][c] w_Ib[c]
            const double wcol = lds[C::OFF_WVEC + 16 * Ib + fr];
            wave_sel<NW, FIX>(wv, [&](auto wc) __attribute__((always_inline)) {
                constexpr int WV = decltype(wc)::value;
                for_tiles<C, WV>([&](auto sc_, auto Ic, auto Jc) __attribute__((always_inline)) {
                    constexpr int s = decltype(sc_)::value, I = decltype(Ic)::value, J = decltype(Jc)::value;
                    if (J == Ib && I < J) {
                        constexpr int t = s * NW + WV;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double pr = rowgroup_sum16(acc[s][r] * wcol);
                            if (fr == 0) lds[C::OFF_PART + t * 16 + fq + 4 * r] = pr;
                        }
                    }
                });
            });
            __syncthreads();
        }
    }

    TP_MARK(6);
    // ---- phase H: weights, status, aux
    {
        TP_LANE_CONSTANTS();
        const double n1 = n0 + (double)A.N;
        const double denom = n1 - q1;
        bool bad = false;
        for (int i = tid; i < k; i += C::NTHREADS) {
            const double wi = lds[C::OFF_WVEC + i];
            double out;
            if (conj) out = 1.0 / A.gamma * ((n1 + k + 2) * wi / denom);     // ref:572-575, 836
            else out = 1.0 / A.gamma * wi;                                   // ref:849
            A.weights[w * k + i] = out;
            if (!isfinite(out)) bad = true;
        }
        const int anybad = __syncthreads_or(bad ? 1 : 0);
        if (tid == 0) {
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

// TP_WAVE_SPECIALISE = 1: the whole window body is instantiated once per wave index (tile
// coordinates are immediates everywhere, accumulators never cross a dispatch merge);
// 0: only the tile-touching snippets branch on the wave index.
// Occupancy the register allocator is asked to keep (wavefronts per SIMD; for NW = 4 that is
// workgroups per CU).  The serial pivot chain of the factorisation is hidden by windows in flight,
// not by ILP: on MI355X 4 resident windows per CU with ~270 spilled dwords beat 2 without spills
// (5.7 vs 4.1 M windows/s at k=100).  Tile counts whose accumulators alone exceed the budget get less.
#ifdef TP_MIN_WAVES_PER_SIMD
constexpr int tp_min_waves_for_tiles(int) { return TP_MIN_WAVES_PER_SIMD; }
#else
constexpr int tp_min_waves_for_tiles(int nt) {
    // measured per tile count (tools/sweep_k.py): nt 9: 3 -> +12 % over 2 (k=143); nt 11-12: 2 -> +35 % over 1 (k=175, 191)
    return nt <= 7 ? 4 : nt <= 9 ? 3 : 2;       // nt 6: 4 waves per window at 4 waves/SIMD, +6 % over 2 x 2 (k=95)
}
#endif

#ifndef TP_WAVE_SPECIALISE
#define TP_WAVE_SPECIALISE 1
#endif

template <int NT, int NW, bool LEAN>
__global__ void __launch_bounds__(64 * NW, tp_min_waves_for_tiles(NT)) posterior_fused_kernel(const tp_kargs_t A) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#if TP_WAVE_SPECIALISE
    wave_dispatch<NW>(wv, [&](auto wc) __attribute__((always_inline)) {
        window_body<NT, NW, decltype(wc)::value, LEAN>(A, lds, tid, decltype(wc)::value);
    });
#else
    window_body<NT, NW, -1, LEAN>(A, lds, tid, wv);
#endif
}

// Grams of the aligned C::CH-row blocks of the daily panel, one workgroup per block, with the staging and MFMA loop of
// the window kernel (same tile ownership, same accumulator layout).  Layout [block][tile][2][64 lanes][2] doubles: a
// window's wave loads the four registers of a tile with two coalesced 16-byte-per-lane reads.  Column k of every block
// is the ones column, so the border column of a block sum is t = X'1 and the corner the row count, as in phase D.
// tp_window_sums_launch then adds them up to the block-window sums Q_L the windows read.
template <int NT, int NW>
__global__ void __launch_bounds__(64 * NW) block_gram_kernel(const tp_kargs_t A, double* __restrict__ out) {
    using C = Cfg<NT, NW>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    d4 acc[C::SLOTS];
    static_for<0, C::SLOTS>([&](auto sc_) __attribute__((always_inline)) { acc[decltype(sc_)::value] = d4{0.0, 0.0, 0.0, 0.0}; });
    RowSource ds;
    ds.base = A.panel; ds.ld = A.panel_ld; ds.ridx = nullptr; ds.first = (long long)blockIdx.x * C::CH; ds.sub_row = nullptr;
    ds.count = C::CH; ds.off32 = true;
    gram_phase_lean<C, false, -1>(ds, A.k, lds, tid0, wv, acc);
    constexpr long long TILE = 4 * 64, SLOT = (long long)C::NTILES * TILE;
    double* slot = out + (long long)blockIdx.x * SLOT;
    const int lane = fresh(tid0) & 63;
    wave_dispatch<NW>(wv, [&](auto wc) __attribute__((always_inline)) {
        constexpr int WV = decltype(wc)::value;
        for_tiles<C, WV>([&](auto sc_, auto, auto) __attribute__((always_inline)) {
            constexpr int s = decltype(sc_)::value;
            constexpr long long t = s * NW + WV;
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2* p = (d2*)(slot + t * TILE) + lane;               // 16-byte stores
#pragma unroll
            for (int h = 0; h < 2; ++h) p[64 * h] = d2{acc[s][2 * h], acc[s][2 * h + 1]};
        });
    });
}

// the contiguous layout the LEAN kernel is built for: no index arrays, window-relative 32-bit offsets
inline bool tp_layout_is_lean(const tp_kargs_t& a) {
    if (a.col_idx || a.row_idx || !(a.panel_off32 & 2)) return false;
    if (a.strategy == 0 && (a.hf_row_idx || !(a.hf_off32 & 2))) return false;
    return true;
}

template <int NT, int NW, bool LEAN>
hipError_t launch_variant(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info) {
    using C = Cfg<NT, NW>;
    static std::atomic<unsigned long long> attr_done{0};      // one bit per device (tp_allow_dynamic_lds)
    { hipError_t e = tp_allow_dynamic_lds(attr_done, posterior_fused_kernel<NT, NW, LEAN>, C::LDS_BYTES); if (e != hipSuccess) return e; }
    if (info) { info->grid = grid; info->block = C::NTHREADS; info->lds_bytes = C::LDS_BYTES; info->ntile = NT; }
    const int grid8 = 8 * ((grid + 7) / 8);            // whole rounds of the 8 XCDs (window_body maps ids to windows)
    hipLaunchKernelGGL((posterior_fused_kernel<NT, NW, LEAN>), dim3(grid8), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    return hipGetLastError();
}

// launcher of the one-wave-per-window kernel of a tile count (posterior_wave_nt.hip), or nullptr
typedef hipError_t (*tp_wave_launch_fn)(const tp_kargs_t&, int, hipStream_t, tp_launch_info_t*, bool lean);

template <int NT, int NW>
hipError_t launch_one(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info, tp_wave_launch_fn wave) {
    using C = Cfg<NT, NW>;
    static_assert(C::CH == TP_PREFIX_BLOCK_ROWS(NT), "posterior_kernels.h: block rows of the shared Gram prefixes");
    if (wave_mode(a) == 2) wave = nullptr;         // read-backs, custom right-hand sides, shifts: the multi-wave kernel
    if (!tp_layout_is_lean(a)) {
        // general layout: the one- / two-wave kernels stage a pass's row indices in LDS (posterior_wave_impl.h,
        // WCfg::OFF_SUB); a batch whose passes are too long for that is answered with hipErrorNotSupported before anything
        // is launched and stays on the multi-wave kernel
        if (wave) {
            const hipError_t e = wave(a, grid, stream, info, false);
            if (e != hipErrorNotSupported) return e;
        }
        return launch_variant<NT, NW, false>(a, grid, stream, info);
    }
    if (a.winsum != nullptr) {
        // the shared sums first, on the same stream: part of every run, nothing is kept between runs
        static std::atomic<unsigned long long> attr_done{0};
        { hipError_t e = tp_allow_dynamic_lds(attr_done, block_gram_kernel<NT, NW>, C::LDS_BYTES); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL((block_gram_kernel<NT, NW>), dim3((unsigned)a.prefix_nblk), dim3(C::NTHREADS), C::LDS_BYTES, stream, a,
                           (double*)a.prefix);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        int n_L = 0;
        while (n_L < TP_WINSUM_MAX_L && a.winsum_L[n_L] > 0) ++n_L;
        e = tp_window_sums_launch(a.prefix, (double*)a.winsum, a.prefix_nblk, (size_t)C::NTILES * 256, a.winsum_L, n_L, stream);
        if (e != hipSuccess) return e;
    }
    if (wave) {
        const hipError_t e = wave(a, grid, stream, info, true);
        if (e != hipErrorNotSupported) return e;
    }
    return launch_variant<NT, NW, true>(a, grid, stream, info);
}

template <int NT, int NW>
int blocks_per_cu() {
    using C = Cfg<NT, NW>;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)posterior_fused_kernel<NT, NW, true>, C::NTHREADS,
                                                     C::LDS_BYTES) != hipSuccess || n < 1)
        n = 1;
    return n;
}

}  // namespace


// wavefronts per workgroup for a tile count: the tiles (NT (NT+1)/2 x 8 accumulator registers, spread over the
// waves) must fit the register file at the occupancy tp_min_waves_for_tiles asks for; measured per tile count
#ifdef TP_NW_OVERRIDE
constexpr int tp_waves_for_tiles(int) { return TP_NW_OVERRIDE; }
#else
constexpr int tp_waves_for_tiles(int nt) { return nt <= 3 ? 1 : (nt <= 5 ? 2 : (nt <= 12 ? 4 : 8)); }
#endif
