// posterior_tiled_wave.h - the Gram super-tile of the large-k path, ONE wavefront per 64 x 64 super-tile.
// Included by posterior_tiled.hip (inside its anonymous namespace, after its helpers).
//
// The 4-wave Gram kernels (tile64_kernel<MODE_GRAM>, tiled_gram_lean_kernel) stage 16 rows at a time through LDS
// and run 16 MFMAs per wave between two workgroup barriers: 64-67 % of the time the matrix pipe is busy
// (tools/gram_loop_probe.hip: the loop shape itself tops out at 75 %).  Here a wave owns all 16 tiles of the
// super-tile (128 accumulator registers, pinned to AGPRs) and loads the MFMA operands of a 4-row k-step straight
// from the panels - lane (fq, fr) reads row 4s + fq, column 16 i + fr: eight 8-byte loads are the A and B operands of
// all 16 MFMAs of the k-step.  No LDS, no barrier, three k-steps of loads in flight, two waves per SIMD.
// Same loads, same arithmetic per element, same summation order as the 4-wave kernels: results are bit-identical
// (the never-read tiles below the diagonal of a diagonal super-tile are left zero instead of being computed).
// The inline-assembly MFMAs follow the rules of posterior_wave_impl.h (s_nop 1 in front of every MFMA, a settle of
// 24 wait states before any other use of an accumulator; tools/check_mfma_hazards.py checks the generated ISA).

struct TRows {
    const double* base;     // panel
    long long ld;           // leading dimension (doubles)
    const int* ridx;        // explicit rows of this window, or nullptr
    long long first;        // first row (contiguous)
    const double* rowc;     // per-row constant: border entry c sqrt(s) z_r (intraday) / risk-free adjustment (daily), or nullptr
    int count;              // rows
    int count0;             // two row ranges (shared daily sums): staged rows r >= count0 are panel rows first + r + jump
    int jump;
};

__device__ __forceinline__ void tw_mfma_agpr(d4& c, double a, double b) {
    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void tw_pin1(d4& c0) { asm volatile("" : "+a"(c0)); }
__device__ __forceinline__ void tw_settle8(d4& c0, d4& c1, d4& c2, d4& c3, d4& c4, d4& c5, d4& c6, d4& c7) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7));
}

// One pass over rows: acc(a, b) += rows[:, A group a]' rows[:, B group b], a, b = 0..3, 4 rows per k-step.
//  HF:  intraday rows sqrt(s) (y - ybar), border column c sqrt(s) z_r (ref:317-333, 489)
//  !HF: daily rows minus the risk-free adjustment, border column 1 (ref:57, 180, 222)
//  NB: 16-column B groups per k-step: 4 = one 64 x 64 super-tile, 8 = two super-tiles side by side (64 x 128, the pair
//  kernel: the A operands are loaded once for both)
//  CS: also keep the column sums of the staged values (this lane's rows only) in cs[] - the Jeffreys rank-one term fused
//  into the Gram kernel needs t = X'1 for the rows AND the columns of the super-tile (gram64_wave_body, RANK1)
//  SUBR (!HF): subtract the reference row yb[] first (shared intraday sums: every row relative to ONE row of the panel, so
//  that a common offset of the returns does not meet the rank-one centring term as a difference of large numbers)
template <bool DIAG, bool EDGE, bool HF, int NB, bool CS = false, bool SUBR = false>
__device__ __forceinline__ void tw_gram_pass(const TRows& src, const int (&co)[4 + NB], const double (&yb)[4 + NB],
                                             const bool (&cval)[4 + NB], const bool (&cbord)[4 + NB], const double sqs, const int lane,
                                             d4 (&acc)[4 * NB], double (&cs)[4 + NB]) {
    constexpr int NO = DIAG ? NB : 4 + NB;      // operand registers per k-step (a diagonal super-tile: its B groups are the A groups)
    const int fq = lane >> 4;
    const int nks = (src.count + 3) >> 2;
    const bool has_c = src.rowc != nullptr;
    auto load = [&](double (&v)[NO], double& rc_, int ks) __attribute__((always_inline)) {
        int r = 4 * ks + fq;
        r = r < src.count ? r : src.count - 1;                      // rows past the end re-read the last row (masked below)
        const int rr = r + (r >= src.count0 ? src.jump : 0);
        const long long row = src.ridx ? (long long)src.ridx[rr] : src.first + rr;
        const double* p = src.base + row * src.ld;
#pragma unroll
        for (int i = 0; i < NO; ++i) v[i] = p[co[i]];
        rc_ = 0.0;
        if (has_c) rc_ = src.rowc[rr];
    };
    auto step = [&](double (&v)[NO], double rc_, int ks, auto maskc) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(maskc)::value != 0;
        if (HF) {
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                v[i] = sqs * (v[i] - yb[i]);                                       // sqrt(s) (y - ybar)
                if (EDGE) v[i] = cval[i] ? v[i] : (cbord[i] ? rc_ : 0.0);          // border: c sqrt(s) z_r
            }
        } else {
            if constexpr (SUBR) {
#pragma unroll
                for (int i = 0; i < NO; ++i) v[i] -= yb[i];
            }
            if (has_c) {
#pragma unroll
                for (int i = 0; i < NO; ++i) v[i] -= rc_;                          // x - rf (ref:57)
            }
            if (EDGE) {
#pragma unroll
                for (int i = 0; i < NO; ++i) v[i] = cval[i] ? v[i] : (cbord[i] ? 1.0 : 0.0);   // border: ones -> t
            }
        }
        if (MASK) {
            const bool rv = 4 * ks + fq < src.count;
#pragma unroll
            for (int i = 0; i < NO; ++i) v[i] = rv ? v[i] : 0.0;
        }
        if constexpr (CS) {
#pragma unroll
            for (int i = 0; i < NO; ++i) cs[i] += v[i];
        }
        static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
            static_for_t<0, NB>([&](auto bc) __attribute__((always_inline)) {
                constexpr int b = decltype(bc)::value;
                // a diagonal super-tile: the tiles below its diagonal are never read (the diagonal-block kernel and the
                // left-looking update use column >= row only) - 10 MFMAs per k-step instead of 16
                if constexpr (!DIAG || a <= b) tw_mfma_agpr(acc[NB * a + b], v[a], v[DIAG ? b : 4 + b]);
            });
        });
    };
    if (nks <= 0) return;
    double va[NO], vb[NO], vc[NO];
    double ra = 0.0, rb = 0.0, rc = 0.0;
    load(va, ra, 0);
    load(vb, rb, 1);
    static_for_t<0, 4 * NB>([&](auto tc) __attribute__((always_inline)) { tw_pin1(acc[decltype(tc)::value]); });
    int ks = 0;
#pragma nounroll
    for (; 4 * (ks + 3) <= src.count; ks += 3) {
        load(vc, rc, ks + 2);
        step(va, ra, ks, std::integral_constant<int, 0>{});
        load(va, ra, ks + 3);
        step(vb, rb, ks + 1, std::integral_constant<int, 0>{});
        load(vb, rb, ks + 4);
        step(vc, rc, ks + 2, std::integral_constant<int, 0>{});
    }
    if (ks < nks) {
        if (ks + 2 < nks) load(vc, rc, ks + 2);
        step(va, ra, ks, std::integral_constant<int, 1>{});
        if (ks + 1 < nks) step(vb, rb, ks + 1, std::integral_constant<int, 1>{});
        if (ks + 2 < nks) step(vc, rc, ks + 2, std::integral_constant<int, 1>{});
    }
    static_for_t<0, NB / 2>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = 8 * decltype(gc)::value;
        tw_settle8(acc[g], acc[g + 1], acc[g + 2], acc[g + 3], acc[g + 4], acc[g + 5], acc[g + 6], acc[g + 7]);
    });
}

// PAIR: the wave owns super-tiles (SI, SJ) AND (SI, SJ + 1): 32 tiles = all 256 AGPRs, one wave per SIMD, 12 operand loads
// and 32 MFMAs per k-step (a diagonal pair: 8 loads, 26 MFMAs) instead of 2 x (8 loads, 16 MFMAs)
// RANK1 (Jeffreys, 64 x 64 form): J = T - t t'/N (+ the optional shift d I + e 1 1') applied to the super-tile while it is
// still in registers, instead of by tiled_rank1_kernel's read-modify-write pass over the whole arena (34 % of a Jeffreys
// run at k = 500: 9.1 ms per 8,192 windows, its column of t read with a 4 KB stride).  t for the rows of SI and the columns
// of SJ: the border column of the shared table slots of super-tiles (SI, NS-1) and (SJ, NS-1) (four lanes hold it) plus
// the column sums of the rows this wave stages itself, put together in 1 KB of LDS.
template <bool DIAG, bool EDGE, bool PAIR = false, bool RANK1 = false>
__device__ __forceinline__ void gram64_wave_body(const tp_kargs_t& A, const tp_tiled_ws_t& ws, const long long wl, const int SI,
                                                 const int SJ, double* tv_lds = nullptr) {
    static_assert(!(PAIR && RANK1), "the fused rank-one term is built for the 64 x 64 form");
    constexpr int NB = PAIR ? 8 : 4;
    constexpr int NC = 4 + NB;
    const int lane = threadIdx.x;
    const int fr = lane & 15, fq = lane >> 4;
    const long long w = A.w_first + wl;
    const int k = A.k, KP = ws.KP;
    double* M = ws.arena + wl * (long long)KP * KP;
    const int* cols = A.col_idx ? A.col_idx + w * k : nullptr;
    const bool conj = A.strategy == 0;
    const int mm = conj ? (A.hf_count ? A.hf_count[w] : A.m) : 0;
    const int nr = A.n_rows ? A.n_rows[w] : A.n_r;
    const double sqs = conj ? ws.scal[wl * 8 + 1] : 0.0;
    const double* ybar = ws.ybar + wl * KP;

    int co[NC];
    double yb[NC];
    bool cval[NC], cbord[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        // operand groups 0..3: the A groups (super-tile column SI); 4..: the B groups of SJ (and SJ + 1).  A diagonal
        // super-tile uses the first NB entries only: SI's groups, then (pair) those of SI + 1
        const int st = DIAG ? SI + (i >> 2) : (i < 4 ? SI : SJ + ((i - 4) >> 2));
        const int gc = 64 * st + fr + 16 * (i & 3);
        cval[i] = !EDGE || gc < k;
        cbord[i] = EDGE && gc == k;
        const int gcl = cval[i] ? gc : k - 1;                      // padding columns re-read column k-1 (masked)
        co[i] = cols ? cols[gcl] : gcl;
        yb[i] = (conj && cval[i]) ? ybar[gcl] : 0.0;
    }
    d4 acc[4 * NB];
    static_for_t<0, 4 * NB>([&](auto tc) __attribute__((always_inline)) {
        acc[decltype(tc)::value] = d4{0.0, 0.0, 0.0, 0.0};
        tw_pin1(acc[decltype(tc)::value]);
    });

    double cs[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) cs[i] = 0.0;
    if constexpr (!RANK1) {
        if (conj) {
            TRows hs;
            hs.base = A.hf_panel; hs.ld = A.hf_ld;
            hs.ridx = A.hf_row_idx ? A.hf_row_idx + w * (long long)A.m : nullptr;
            hs.first = A.hf_start ? A.hf_start[w] : 0;
            hs.rowc = ws.zc + wl * (long long)A.m;
            hs.count = mm; hs.count0 = 0x7fffffff; hs.jump = 0;
            tw_gram_pass<DIAG, EDGE, true, NB>(hs, co, yb, cval, cbord, sqs, lane, acc, cs);
        }
    }
    // daily rows; with the shared block-window sums only the rows in front of the first whole aligned block and behind
    // the last one (see gram64_lean_body)
    TRows ds;
    ds.base = A.panel; ds.ld = A.panel_ld;
    ds.ridx = A.row_idx ? A.row_idx + w * (long long)A.n_r : nullptr;
    ds.first = A.start ? A.start[w] : 0;
    ds.rowc = A.rf_adj ? A.rf_adj + w * (long long)A.n_r : nullptr;
    ds.count = nr; ds.count0 = 0x7fffffff; ds.jump = 0;
    const long long pb0 = (ds.first + CH - 1) / CH, pb1 = (ds.first + nr) / CH;
    const int Lw = (int)(pb1 - pb0);
    const int li = Lw == A.winsum_L[0] ? 0 : Lw == A.winsum_L[1] ? 1 : Lw == A.winsum_L[2] ? 2 : Lw == A.winsum_L[3] ? 3 : -1;
    const long long tb0 = pb0 - A.prefix_blk0;                    // position in the tables of the sub-batch in flight
    const bool shared = A.winsum != nullptr && !ds.ridx && Lw > 0 && li >= 0 && tb0 >= 0 && tb0 + Lw <= A.prefix_nblk;
    if (shared) {
        ds.count0 = (int)(CH * pb0 - ds.first);
        ds.jump = (int)(CH * pb1 - ds.first) - ds.count0;
        ds.count = ds.count0 + (int)(ds.first + nr - CH * pb1);
    }
    tw_gram_pass<DIAG, EDGE, false, NB, RANK1>(ds, co, yb, cval, cbord, sqs, lane, acc, cs);

    // the table slot (tile row a of the super-tile at a time) and the store to the arena
    typedef double d2 __attribute__((ext_vector_type(2)));
    const long long ntile = (long long)ws.NS * (ws.NS + 1) / 2;
    // ---- RANK1: t for the 64 rows of SI (tv[0..63]) and the 64 columns of SJ (tv[64..127])
    double invN = 0.0, sh_d = 0.0, sh_e = 0.0;
    if constexpr (RANK1) {
        invN = A.center_rows == 2 ? 0.0 : 1.0 / (double)(A.center_rows ? nr : A.N);
        sh_d = A.shift ? A.shift[2 * w] : 0.0;
        sh_e = A.shift ? A.shift[2 * w + 1] : 0.0;
        // column sums of the rows staged here: add up the four row groups of a k-step (lanes 16 apart)
#pragma unroll
        for (int i = 0; i < (DIAG ? NB : NC); ++i) {
            double x = cs[i];
            x += __shfl_xor(x, 16);
            x += __shfl_xor(x, 32);
            cs[i] = x;
        }
        if (fq == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tv_lds[16 * i + fr] = cs[i];
                tv_lds[64 + 16 * i + fr] = cs[DIAG ? i : 4 + i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (shared) {
            // border column k of the table: super-tile column NS - 1, 16-column group bk, lane column kc; the lanes fr == kc
            // hold rows fq + 4 r of tile row a in the registers of the slot
            const int kl = k - 64 * (ws.NS - 1), bk = kl >> 4, kc = kl & 15;
            if (fr == kc) {
#pragma unroll
                for (int side = 0; side < (DIAG ? 1 : 2); ++side) {
                    const int S = side == 0 ? SI : SJ;
                    const d2* qb = (const d2*)(A.winsum + (((long long)li * A.prefix_nblk + tb0) * ntile + pair_index(S, ws.NS - 1, ws.NS)) * (SB * SB)) + lane;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const d2 lo = qb[a * 512 + (bk * 2 + 0) * 64], hi = qb[a * 512 + (bk * 2 + 1) * 64];
                        // rows of super-tile S that are assets only (the border row itself is never used as t)
                        tv_lds[64 * side + 16 * a + fq + 0] += lo[0];
                        tv_lds[64 * side + 16 * a + fq + 4] += lo[1];
                        tv_lds[64 * side + 16 * a + fq + 8] += hi[0];
                        tv_lds[64 * side + 16 * a + fq + 12] += hi[1];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (DIAG && lane < 64) tv_lds[64 + lane] = tv_lds[lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
    double tj[NB];
    if constexpr (RANK1) {
#pragma unroll
        for (int b = 0; b < NB; ++b) tj[b] = tv_lds[64 + 16 * b + fr];
    }
    // (the super-tiles of a pair are neighbours in the row-major numbering of the triangle: slot of (SI, SJ + 1) = slot + 1)
    const d2* q = shared ? (const d2*)(A.winsum + (((long long)li * A.prefix_nblk + tb0) * ntile + pair_index(SI, SJ, ws.NS)) * (SB * SB)) + lane
                         : nullptr;
    static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value;
        d2 v2[NB][2];
        if (shared) {
            // [..][tile row a][16-column group b][2][64 lanes][2]: registers (0,1) and (2,3) of a lane are 16 contiguous bytes
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) v2[b][h] = q[(b >> 2) * (SB * SB / 2) + a * 512 + ((b & 3) * 2 + h) * 64];
        }
        static_for_t<0, NB>([&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            d4 x = acc[NB * a + b];              // (below the diagonal of a diagonal super-tile: the zeros it started with)
            if (shared && (!DIAG || a <= b)) {
                x[0] += v2[b][0][0]; x[1] += v2[b][0][1];
                x[2] += v2[b][1][0]; x[3] += v2[b][1][1];
            }
            if constexpr (RANK1) {
                if (!DIAG || a <= b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gi = 64 * SI + 16 * a + fq + 4 * r, gj = 64 * SJ + 16 * b + fr;
                        const double ti = tv_lds[16 * a + fq + 4 * r];
                        const double add = sh_e + (gi == gj ? sh_d : 0.0) - invN * (ti * tj[b]);
                        x[r] += (!EDGE || (gi < k && gj < k)) ? add : 0.0;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(long long)(64 * SI + 16 * a + fq + 4 * r) * KP + 64 * SJ + 16 * b + fr] = x[r];
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

__global__ void __launch_bounds__(64, 2) tiled_gram_wave_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws) {
    long long wl;
    int tile, SI, SJ;
    if (!xcd_window_tile(ws.NS * (ws.NS + 1) / 2, A.w_count, wl, tile)) return;
    pair_decode(tile, ws.NS, SI, SJ);
    const bool edge = !(64 * SJ + 63 < A.k);                      // SI <= SJ: otherwise every column is a real asset
    if (SI == SJ) {
        if (edge) gram64_wave_body<true, true>(A, ws, wl, SI, SJ);
        else gram64_wave_body<true, false>(A, ws, wl, SI, SJ);
    } else {
        if (edge) gram64_wave_body<false, true>(A, ws, wl, SI, SJ);
        else gram64_wave_body<false, false>(A, ws, wl, SI, SJ);
    }
}

// ---- shared intraday sums (conjugate, large-k path) ------------------------------------------------------------------
// The intraday windows of consecutive dates overlap by all but one day: 388 of 389 rows at configs[2], 1,637 of 1,715 at
// configs[4], and pushing them through the MFMAs for every window was 52 % / 67 % of those runs.  Here the raw rows of each
// B-row block (one day of bars) are multiplied ONCE per sub-batch (tiled_hf_block_gram_kernel), tp_window_sums_kernel adds L
// consecutive block Grams up (additions only, as for the daily panel), and a window takes
//     S0 = s [ Q[b0] + (its rows outside whole blocks)' (same) - t t'/m ],      t = column sums of its rows
// with t from the border (ones) column of Q and the column sums of the edge rows - the centring the two-pass form gets from
// (y - ybar) becomes a rank-one term in registers.  What the two-pass form fed through the border column, c sqrt(s) z_r with
// z_r = (y_r - ybar).w0, is c S0 w0: every super-tile writes its 64-row pieces of S0 w0 (rows of I from the columns of J,
// and for I < J rows of J from the columns of I) to ws.part, and tiled_clear_kernel adds them up in a fixed order, forms
// q0 = w0' S0 w0 and c (ref:415-418) and puts c S0 w0 into the border column.  No tiled_prior_kernel, no pass over the
// intraday rows per window at all.
// the reference row of the shared intraday sums: row 0 of the intraday panel (the same for every sub-batch and every batch
// over this panel: results do not depend on how a run is cut); a non-finite entry counts as 0 (it must not poison windows
// that do not contain it)
__device__ __forceinline__ double tw_reference(const tp_kargs_t& A, const bool valid, const int col) {
    const double x = A.hf_panel[col];
    return (valid && isfinite(x)) ? x : 0.0;
}

template <bool DIAG, bool EDGE>
__device__ __forceinline__ void hfblock64_wave_body(const tp_kargs_t& A, const tp_tiled_ws_t& ws, double* out, const long long blk,
                                                    const long long tile, const int SI, const int SJ) {
    constexpr int NB = 4, NC = 8;
    const int lane = threadIdx.x;
    const int fr = lane & 15;
    const int k = A.k;
    int co[NC];
    double yb[NC], cs[NC];
    bool cval[NC], cbord[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int st = DIAG ? SI + (i >> 2) : (i < 4 ? SI : SJ + ((i - 4) >> 2));
        const int gc = 64 * st + fr + 16 * (i & 3);
        cval[i] = !EDGE || gc < k;
        cbord[i] = EDGE && gc == k;                                 // ones: the border column of a block Gram is its column sums
        co[i] = cval[i] ? gc : k - 1;
        yb[i] = tw_reference(A, cval[i], co[i]); cs[i] = 0.0;
    }
    d4 acc[4 * NB];
    static_for_t<0, 4 * NB>([&](auto tc) __attribute__((always_inline)) {
        acc[decltype(tc)::value] = d4{0.0, 0.0, 0.0, 0.0};
        tw_pin1(acc[decltype(tc)::value]);
    });
    TRows hs;
    hs.base = A.hf_panel; hs.ld = A.hf_ld; hs.ridx = nullptr; hs.rowc = nullptr;
    hs.first = A.hf_row0 + blk * A.hf_blk_rows;
    hs.count = A.hf_blk_rows; hs.count0 = 0x7fffffff; hs.jump = 0;
    tw_gram_pass<DIAG, EDGE, false, NB, false, true>(hs, co, yb, cval, cbord, 0.0, lane, acc, cs);
    typedef double d2 __attribute__((ext_vector_type(2)));
    const long long ntile = (long long)ws.NS * (ws.NS + 1) / 2;
    d2* p = (d2*)(out + (blk * ntile + tile) * (SB * SB)) + lane;
    static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value;
        static_for_t<0, NB>([&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            const d4 x = acc[NB * a + b];                           // (below the diagonal of a diagonal super-tile: zeros)
            p[a * 512 + (b * 2 + 0) * 64] = d2{x[0], x[1]};
            p[a * 512 + (b * 2 + 1) * 64] = d2{x[2], x[3]};
        });
    });
}

__global__ void __launch_bounds__(64, 2) tiled_hf_block_gram_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws, double* out) {
    // the super-tiles of one block on ONE XCD (xcd_window_tile with blocks in the place of windows): they all read the
    // block's 78 rows, and dealt round-robin over the 8 XCDs every L2 fetched them for itself (6.3 MB fetched per block
    // at k = 1000 for 0.6 MB of rows)
    long long blk;
    int tile, SI, SJ;
    if (!xcd_window_tile(ws.NS * (ws.NS + 1) / 2, A.hf_nblk, blk, tile)) return;
    pair_decode(tile, ws.NS, SI, SJ);
    const bool edge = !(64 * SJ + 63 < A.k);
    if (SI == SJ) {
        if (edge) hfblock64_wave_body<true, true>(A, ws, out, blk, tile, SI, SJ);
        else hfblock64_wave_body<true, false>(A, ws, out, blk, tile, SI, SJ);
    } else {
        if (edge) hfblock64_wave_body<false, true>(A, ws, out, blk, tile, SI, SJ);
        else hfblock64_wave_body<false, false>(A, ws, out, blk, tile, SI, SJ);
    }
}

// t (column sums over a window's rows) for the 64 rows of SI -> tv[0..63] and the 64 columns of SJ -> tv[64..127]: the column
// sums `cs` of the rows this wave staged (this lane's row group only, on entry) plus - when `tab` - the border column of the
// table slots whose row blocks are SI and SJ (qI / qJ: slot of super-tile (S, NS-1), already offset by the lane)
template <bool DIAG>
__device__ __forceinline__ void tw_build_t(double (&cs)[8], double* tv, const bool tab, const double* qI, const double* qJ,
                                           const int k, const int NS, const int lane) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < (DIAG ? 4 : 8); ++i) {
        double x = cs[i];
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        cs[i] = x;
    }
    if (fq == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tv[16 * i + fr] = cs[i];
            tv[64 + 16 * i + fr] = cs[DIAG ? i : 4 + i];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (tab) {
        // border column k of the table: super-tile column NS - 1, 16-column group bk, lane column kc; the lanes fr == kc
        // hold rows fq + 4 r of tile row a in the registers of the slot
        const int kl = k - 64 * (NS - 1), bk = kl >> 4, kc = kl & 15;
        if (fr == kc) {
#pragma unroll
            for (int side = 0; side < (DIAG ? 1 : 2); ++side) {
                const d2* qb = (const d2*)(side == 0 ? qI : qJ);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const d2 lo = qb[a * 512 + (bk * 2 + 0) * 64], hi = qb[a * 512 + (bk * 2 + 1) * 64];
                    tv[64 * side + 16 * a + fq + 0] += lo[0];
                    tv[64 * side + 16 * a + fq + 4] += lo[1];
                    tv[64 * side + 16 * a + fq + 8] += hi[0];
                    tv[64 * side + 16 * a + fq + 12] += hi[1];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (DIAG) tv[64 + lane] = tv[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// One super-tile of a conjugate window with the shared intraday sums.  lds: tv[128] | pI[64] | pJ[64]
template <bool DIAG, bool EDGE>
__device__ __forceinline__ void gram64_wave_hfs_body(const tp_kargs_t& A, const tp_tiled_ws_t& ws, const long long wl, const int SI,
                                                     const int SJ, double* lds) {
    constexpr int NB = 4, NC = 8;
    typedef double d2 __attribute__((ext_vector_type(2)));
    double* tv = lds;
    double* pI = lds + 128;
    double* pJ = lds + 192;
    double* w0J = lds + 256;        // prior weights of the columns of SJ (0 beyond the last asset)
    const int lane = threadIdx.x;
    const int fr = lane & 15, fq = lane >> 4;
    const long long w = A.w_first + wl;
    const int k = A.k, KP = ws.KP, NS = ws.NS;
    double* M = ws.arena + wl * (long long)KP * KP;
    const int mm = A.m;                                             // uniform intraday row count (host-checked)
    const int nr = A.n_rows ? A.n_rows[w] : A.n_r;
    const long long ntile = (long long)NS * (NS + 1) / 2;
    int co[NC];
    double yb[NC], cs[NC];
    bool cval[NC], cbord[NC], cnone[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int st = DIAG ? SI + (i >> 2) : (i < 4 ? SI : SJ + ((i - 4) >> 2));
        const int gc = 64 * st + fr + 16 * (i & 3);
        cval[i] = !EDGE || gc < k;
        cbord[i] = EDGE && gc == k;
        cnone[i] = false;
        co[i] = cval[i] ? gc : k - 1;
        yb[i] = tw_reference(A, cval[i], co[i]); cs[i] = 0.0;
    }
    pI[lane] = 0.0;
    pJ[lane] = 0.0;
    d4 acc[4 * NB];
    static_for_t<0, 4 * NB>([&](auto tc) __attribute__((always_inline)) {
        acc[decltype(tc)::value] = d4{0.0, 0.0, 0.0, 0.0};
        tw_pin1(acc[decltype(tc)::value]);
    });
    // ---- intraday: the rows outside the whole blocks, raw (no border column: the column sums are kept in cs)
    const long long hfirst = A.hf_start[w];
    const long long Bk = A.hf_blk_rows;
    const long long hb0 = (hfirst - A.hf_row0 + Bk - 1) / Bk, hb1 = (hfirst + mm - A.hf_row0) / Bk;
    {
        TRows hs;
        hs.base = A.hf_panel; hs.ld = A.hf_ld; hs.ridx = nullptr; hs.rowc = nullptr;
        hs.first = hfirst;
        hs.count0 = (int)(A.hf_row0 + Bk * hb0 - hfirst);
        hs.jump = (int)(Bk * (hb1 - hb0));
        hs.count = hs.count0 + (int)(hfirst + mm - (A.hf_row0 + Bk * hb1));
        tw_gram_pass<DIAG, EDGE, false, NB, true, true>(hs, co, yb, cval, cnone, 0.0, lane, acc, cs);
    }
    const double* hq = A.hf_winsum + (hb0 * ntile) * (SB * SB);
    tw_build_t<DIAG>(cs, tv, true, hq + pair_index(SI, NS - 1, NS) * (SB * SB) + 2 * lane, hq + pair_index(SJ, NS - 1, NS) * (SB * SB) + 2 * lane,
                     k, NS, lane);
    // ---- S0 = s (Q + edge - t t'/m) in registers, and this super-tile's pieces of S0 w0
    {
        const double n0 = A.n0[w];
        const double sc = n0 * ((double)mm / ((double)mm - 1.0));
        const double minv = 1.0 / (double)mm;
        const double* w0 = A.w0 + w * k;
        double colp[NB];
        {
            const int gj = 64 * SJ + lane;
            w0J[lane] = (!EDGE || gj < k) ? w0[gj < k ? gj : k - 1] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) colp[b] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const d2* q = (const d2*)(hq + pair_index(SI, SJ, NS) * (SB * SB)) + lane;
        static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
            double ti[4], w0i[4], rowp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = 64 * SI + 16 * a + fq + 4 * r;              // (SI <= SJ: rows of SI are assets unless SI is the last block)
                ti[r] = tv[16 * a + fq + 4 * r];
                w0i[r] = (gi < k) ? w0[gi] : 0.0;
                rowp[r] = 0.0;
            }
            d2 v2[NB][2];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) v2[b][h] = q[a * 512 + (b * 2 + h) * 64];
            static_for_t<0, NB>([&](auto bc) __attribute__((always_inline)) {
                constexpr int b = decltype(bc)::value;
                if constexpr (!DIAG || a <= b) {
                    d4 x = acc[NB * a + b];
                    x[0] += v2[b][0][0]; x[1] += v2[b][0][1];
                    x[2] += v2[b][1][0]; x[3] += v2[b][1][1];
                    const double tjb = tv[64 + 16 * b + fr], w0jb = w0J[16 * b + fr];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gi = 64 * SI + 16 * a + fq + 4 * r, gj = 64 * SJ + 16 * b + fr;
                        double y = sc * fma(-minv * ti[r], tjb, x[r]);
                        y = (gi < k && gj < k) ? y : 0.0;
                        x[r] = y;
                        rowp[r] = fma(y, w0jb, rowp[r]);
                        if (!DIAG || a < b) colp[b] = fma(y, w0i[r], colp[b]);
                    }
                    acc[NB * a + b] = x;
                    tw_pin1(acc[NB * a + b]);
                }
            });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double sum = rowgroup_sum16(rowp[r]);
                if (fr == 0) pI[16 * a + fq + 4 * r] = sum;
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double x = colp[b];
            x += __shfl_xor(x, 16);
            x += __shfl_xor(x, 32);
            if (fq == 0) {
                if (DIAG) pI[16 * b + fr] += x;                     // the mirrored halves of a diagonal super-tile: same rows
                else pJ[16 * b + fr] = x;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        double* part = ws.part + (wl * NS * NS) * 64;
        part[((long long)SI * NS + SJ) * 64 + lane] = pI[lane];
        if (!DIAG) part[((long long)SJ * NS + SI) * 64 + lane] = pJ[lane];
    }
    // ---- daily rows on top (as in gram64_wave_body)
    TRows ds;
    ds.base = A.panel; ds.ld = A.panel_ld;
    ds.ridx = nullptr;
    ds.first = A.start[w];
    ds.rowc = nullptr;
    ds.count = nr; ds.count0 = 0x7fffffff; ds.jump = 0;
    const long long pb0 = (ds.first + CH - 1) / CH, pb1 = (ds.first + nr) / CH;
    const int Lw = (int)(pb1 - pb0);
    const int li = Lw == A.winsum_L[0] ? 0 : Lw == A.winsum_L[1] ? 1 : Lw == A.winsum_L[2] ? 2 : Lw == A.winsum_L[3] ? 3 : -1;
    const long long tb0 = pb0 - A.prefix_blk0;
    const bool shared = A.winsum != nullptr && Lw > 0 && li >= 0 && tb0 >= 0 && tb0 + Lw <= A.prefix_nblk;
    if (shared) {
        ds.count0 = (int)(CH * pb0 - ds.first);
        ds.jump = (int)(CH * pb1 - ds.first) - ds.count0;
        ds.count = ds.count0 + (int)(ds.first + nr - CH * pb1);
    }
    tw_gram_pass<DIAG, EDGE, false, NB>(ds, co, yb, cval, cbord, 0.0, lane, acc, cs);
    const d2* q = shared ? (const d2*)(A.winsum + (((long long)li * A.prefix_nblk + tb0) * ntile + pair_index(SI, SJ, NS)) * (SB * SB)) + lane
                         : nullptr;
    static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value;
        d2 v2[NB][2];
        if (shared) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) v2[b][h] = q[a * 512 + (b * 2 + h) * 64];
        }
        static_for_t<0, NB>([&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            d4 x = acc[NB * a + b];
            if (shared && (!DIAG || a <= b)) {
                x[0] += v2[b][0][0]; x[1] += v2[b][0][1];
                x[2] += v2[b][1][0]; x[3] += v2[b][1][1];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) M[(long long)(64 * SI + 16 * a + fq + 4 * r) * KP + 64 * SJ + 16 * b + fr] = x[r];
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

__global__ void __launch_bounds__(64, 2) tiled_gram_wave_hfs_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws) {
    long long wl;
    int tile, SI, SJ;
    if (!xcd_window_tile(ws.NS * (ws.NS + 1) / 2, A.w_count, wl, tile)) return;
    pair_decode(tile, ws.NS, SI, SJ);
    const bool edge = !(64 * SJ + 63 < A.k);
    __shared__ double lds[320];
    if (SI == SJ) {
        if (edge) gram64_wave_hfs_body<true, true>(A, ws, wl, SI, SJ, lds);
        else gram64_wave_hfs_body<true, false>(A, ws, wl, SI, SJ, lds);
    } else {
        if (edge) gram64_wave_hfs_body<false, true>(A, ws, wl, SI, SJ, lds);
        else gram64_wave_hfs_body<false, false>(A, ws, wl, SI, SJ, lds);
    }
}

// Jeffreys with the rank-one term fused (no tiled_rank1_kernel behind it)
__global__ void __launch_bounds__(64, 2) tiled_gram_wave_rank1_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws) {
    long long wl;
    int tile, SI, SJ;
    if (!xcd_window_tile(ws.NS * (ws.NS + 1) / 2, A.w_count, wl, tile)) return;
    pair_decode(tile, ws.NS, SI, SJ);
    const bool edge = !(64 * SJ + 63 < A.k);
    __shared__ double tv[128];
    if (SI == SJ) {
        if (edge) gram64_wave_body<true, true, false, true>(A, ws, wl, SI, SJ, tv);
        else gram64_wave_body<true, false, false, true>(A, ws, wl, SI, SJ, tv);
    } else {
        if (edge) gram64_wave_body<false, true, false, true>(A, ws, wl, SI, SJ, tv);
        else gram64_wave_body<false, false, false, true>(A, ws, wl, SI, SJ, tv);
    }
}

// 64 x 128 per wavefront: super-tiles (I, J) and (I, J + 1) of a window, J = I, I + 2, ... (a single one at the end of an odd
// row).  Half the waves, each with all 256 AGPRs, ONE per SIMD: the 70 % matrix-pipe utilisation of the 64 x 64 form comes
// with two waves per SIMD whose vector / load phases do not hide under each other's MFMAs (MI355X: a wave streaming fp64
// MFMAs leaves its SIMD partner one vector instruction per MFMA); here a k-step is 32 MFMAs (2,048 cycles) for 12 loads.
__device__ __forceinline__ int tw_pair_count(int NS) { int n = 0; for (int i = 0; i < NS; ++i) n += (NS - i + 1) / 2; return n; }
__device__ __forceinline__ void tw_pair_decode(int p, int NS, int& SI, int& SJ, bool& two) {
    int i = 0, rem = p;
    while (rem >= (NS - i + 1) / 2) { rem -= (NS - i + 1) / 2; ++i; }
    SI = i; SJ = i + 2 * rem; two = SJ + 1 < NS;
}
__global__ void __launch_bounds__(64, 1) tiled_gram_wave_pair_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws, const int NP) {
    long long wl;
    int tile, SI, SJ;
    bool two;
    if (!xcd_window_tile(NP, A.w_count, wl, tile)) return;
    tw_pair_decode(tile, ws.NS, SI, SJ, two);
    const int last = SJ + (two ? 1 : 0);
    const bool edge = !(64 * last + 63 < A.k);
    if (two) {
        if (SI == SJ) {
            if (edge) gram64_wave_body<true, true, true>(A, ws, wl, SI, SJ);
            else gram64_wave_body<true, false, true>(A, ws, wl, SI, SJ);
        } else {
            if (edge) gram64_wave_body<false, true, true>(A, ws, wl, SI, SJ);
            else gram64_wave_body<false, false, true>(A, ws, wl, SI, SJ);
        }
    } else {
        if (SI == SJ) {
            if (edge) gram64_wave_body<true, true>(A, ws, wl, SI, SJ);
            else gram64_wave_body<true, false>(A, ws, wl, SI, SJ);
        } else {
            if (edge) gram64_wave_body<false, true>(A, ws, wl, SI, SJ);
            else gram64_wave_body<false, false>(A, ws, wl, SI, SJ);
        }
    }
}

// Measured and NOT kept (round 2): the factorisation's SYRK and TRSM super-tiles in the same one-wave form (eight loads
// and 16 MFMAs per k-step, no staging at all).  They read the ARENA - 22 GB per 4,096 windows at k = 500, from HBM, each
// tile a few times - and are bound by bytes in flight, not by the matrix pipe: two waves per SIMD with three k-steps of
// loads each keep fewer bytes in flight than four 4-wave workgroups per CU, and the whole run got slower (k = 500:
// 0.542 instead of 0.572 of the MFMA peak, k = 1000: 0.605 instead of 0.658; gpurun_out/r03j).

// ---- the 64 x 64 diagonal block by ONE wavefront ------------------------------------------------------------------
// tiled_diag_kernel factorises the block with 4 waves and a workgroup barrier per pivot (64 of them): 229 us per block
// step for 4,096 windows at k = 500, 10 % of the run.  Here the block is 4 x 4 MFMA tiles in ONE wave's registers and is
// factorised exactly like phase F of the one-wave register-tile kernel (posterior_wave_impl.h): per 16-row tile row the
// diagonal tile goes through LDS to one column per lane, 16 pivots by v_readlane eliminate it together with 16 identity
// columns (M_a = R_aa^-T), the tile row becomes R_aB = M_a A_aB and the trailing tiles A_IB -= R_aI' R_aB, all by MFMA with
// operands from the accumulators.  The 64 identity columns of the block ride along as 4 more tile columns: they come out
// as Y = R_jj^-T, whose transpose is the R_jj^-1 the TRSM and solve kernels read - no inverse pass.
__device__ __forceinline__ double tw_readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// 1/sqrt(d): v_rsq_f64 seed + one third-order step (see posterior_fused_impl.h, rsqrt_cubic)
__device__ __forceinline__ double tw_rsqrt_cubic(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * y), y, 1.0);
    const double u = fma(e, 0.375, 0.5);
    return fma(y * e, u, y);
}
constexpr int tw_ta(int a, int b) { return a * 4 - a * (a - 1) / 2 + (b - a); }       // A tile (a, b), a <= b: 0..9
constexpr int tw_ty(int b, int a) { return 10 + b * (b + 1) / 2 + a; }                 // Y tile (b, a), a <= b: 10..19

__device__ __forceinline__ void tw_settle2(d4& c0, d4& c1) { asm volatile("s_nop 15\n\ts_nop 7" : "+a"(c0), "+a"(c1)); }
__device__ __forceinline__ void tw_mfma_agpr_neg(d4& c, double a, double b) {
    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+a"(c) : "v"(a), "v"(b));
}

// UPDATE: the block first takes its left-looking update A_jj -= sum_{q<j} R_qj' R_qj here (rows 0 .. 64 j - 1 of the arena,
// columns of super-tile j: four operand loads and ten MFMAs per 4-row k-step, the tiles on and above the diagonal only)
// instead of in a launch of its own (tile64_kernel<MODE_SYRK_DIAG>: 201 us per block step at k = 500, 8 % of the run).
// waves per SIMD the register allocator is asked to keep.  Two (round 3, A/B build -DTP_DIAG_OCC=2: 128 + 128 registers, 51
// spilled) measured 16.36-16.53 ms against 16.28-16.30 ms per 4,096 windows at k = 500 and +-0 at k = 1000: not adopted
#ifndef TP_DIAG_OCC
#define TP_DIAG_OCC 1
#endif
#ifndef TP_DIAG_DEPTH
#define TP_DIAG_DEPTH 8
#endif
template <bool UPDATE>
__global__ void __launch_bounds__(64, TP_DIAG_OCC) tiled_diag_wave_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws, const int j) {
    constexpr int MLD = 17;
    __shared__ __attribute__((aligned(16))) double lds[256 + 256 + 16 * MLD];          // diagonal tile | identity | M_a
    double* DG = lds;
    double* IDT = lds + 256;
    double* MB = lds + 512;
    const int lane = threadIdx.x;
    const int fr = lane & 15, fq = lane >> 4;
    const long long wl = blockIdx.x;
    const int k = A.k, KP = ws.KP;
    double* M = ws.arena + wl * (long long)KP * KP;
    double* blk = M + (long long)(64 * j) * KP + 64 * j;
    const int npiv = (k - 64 * j < SB) ? (k - 64 * j) : SB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i;
        IDT[e] = ((e >> 4) == (e & 15)) ? 1.0 : 0.0;
    }
    d4 acc[20];
    static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value;
        static_for_t<a, 4>([&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            d4 x;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = blk[(long long)(16 * a + fq + 4 * r) * KP + 16 * b + fr];
            acc[tw_ta(a, b)] = x;
        });
        static_for_t<0, a + 1>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            d4 x = d4{0.0, 0.0, 0.0, 0.0};
            if (c == a) {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = (fq + 4 * r == fr) ? 1.0 : 0.0;
            }
            acc[tw_ty(a, c)] = x;
        });
    });
    if (UPDATE && j > 0) {
        const double* col = M + 64 * j + fr;                         // column 16 a + fr of super-tile j, row r: col[r KP + 16 a]
        const int nks = 16 * j;
        auto load = [&](double (&v)[4], int ks) __attribute__((always_inline)) {
            const double* p = col + (long long)(4 * ks + fq) * KP;
#pragma unroll
            for (int a = 0; a < 4; ++a) v[a] = p[16 * a];
        };
        auto step = [&](double (&v)[4]) __attribute__((always_inline)) {
            static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
                static_for_t<a, 4>([&](auto bc) __attribute__((always_inline)) {
                    constexpr int b = decltype(bc)::value;
                    tw_mfma_agpr_neg(acc[tw_ta(a, b)], v[a], v[b]);
                });
            });
        };
        // TP_DIAG_DEPTH k-steps of operand loads in flight (nks = 16 j is a multiple of it: no remainder code).  Eight
        // against the three of round 2 measured +-0 (k = 500, 8,192 windows: 31.7 / 18.3 ms conjugate / Jeffreys either
        // way): the kernel is bound by its serial work (ten MFMAs per k-step, then 64 pivots), not by the latency of
        // the arena rows.
        constexpr int D = TP_DIAG_DEPTH;
        static_assert(16 % D == 0, "the update loop runs whole groups of D k-steps");
        double v[D][4];
        static_for_t<0, D>([&](auto dc) __attribute__((always_inline)) { load(v[decltype(dc)::value], decltype(dc)::value); });
        static_for_t<0, 10>([&](auto tc) __attribute__((always_inline)) { tw_pin1(acc[decltype(tc)::value]); });
#pragma nounroll
        for (int ks = 0; ks < nks; ks += D) {
            static_for_t<0, D>([&](auto dc) __attribute__((always_inline)) {
                constexpr int d = decltype(dc)::value;
                step(v[d]);
                const int nx = ks + D + d;
                load(v[d], nx < nks ? nx : nks - 1);       // (past the end: a re-read of the last k-step, never used)
            });
        }
        tw_settle8(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
        tw_settle2(acc[8], acc[9]);
    }
    double badacc = 0.0;
    static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value;
        const int np = npiv - 16 * a < 16 ? npiv - 16 * a : 16;      // pivots of this tile row (uniform)
        if (np > 0) {
            // (1) diagonal tile -> LDS (row-major) -> one column per lane; lanes 16-31 take the identity's columns
#pragma unroll
            for (int r = 0; r < 4; ++r) DG[(fq + 4 * r) * 16 + fr] = acc[tw_ta(a, a)][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int c16 = lane & 15;
            double e[16];
            const double* src = (lane < 16) ? DG : IDT;
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = src[i * 16 + c16];
            // (2) the pivots (as in posterior_wave_impl.h, phase F)
            double rinv = tw_rsqrt_cubic(tw_readlane_d(e[0], 0));
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (p < np) {
                    e[p] *= rinv;
                    badacc = fma(0.0, rinv, badacc);              // a non-positive or NaN pivot: rinv is an infinity or a NaN
                    double rinv_next = 1.0;
                    if (p + 1 < 16) {
                        const double s1 = tw_readlane_d(e[p], p + 1);
                        e[p + 1] = fma(-s1, e[p], e[p + 1]);
                        double dn = tw_readlane_d(e[p + 1], p + 1);
                        dn = (p + 1 < np) ? dn : 1.0;
                        rinv_next = tw_rsqrt_cubic(dn);
                    }
#pragma unroll
                    for (int i = p + 2; i < 16; ++i) {
                        const double sI = tw_readlane_d(e[p], i);
                        e[i] = fma(-sI, e[p], e[i]);
                    }
                    rinv = rinv_next;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // M_a = R_aa^-T; rows past the last pivot (the border row, padding) stay identity rows
            if (lane >= 16 && lane < 32) {
#pragma unroll
                for (int i = 0; i < 16; ++i) MB[i * MLD + c16] = (i < np) ? e[i] : ((i == c16) ? 1.0 : 0.0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (3) tile row a: R_aB = M_a A_aB (B >= a) and Y_aC = M_a Y_aC (C <= a)
            double mop[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mop[r] = MB[fr * MLD + 4 * r + fq];          // M[fr][4r + fq]
            auto trsm = [&](d4& t) __attribute__((always_inline)) {
                d4 rj = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) rj = __builtin_amdgcn_mfma_f64_16x16x4f64(mop[r], t[r], rj, 0, 0, 0);
                t = rj;
            };
            static_for_t<a, 4>([&](auto bc) __attribute__((always_inline)) { trsm(acc[tw_ta(a, decltype(bc)::value)]); });
            static_for_t<0, a + 1>([&](auto cc) __attribute__((always_inline)) { trsm(acc[tw_ty(a, decltype(cc)::value)]); });
            // (4) trailing tiles: A_IB -= R_aI' R_aB, Y_IC -= R_aI' Y_aC
            static_for_t<a + 1, 4>([&](auto Ic) __attribute__((always_inline)) {
                constexpr int I = decltype(Ic)::value;
                static_for_t<I, 4>([&](auto bc) __attribute__((always_inline)) {
                    constexpr int b = decltype(bc)::value;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[tw_ta(I, b)] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[tw_ta(a, I)][r], acc[tw_ta(a, b)][r], acc[tw_ta(I, b)], 0, 0, 1);
                });
                static_for_t<0, a + 1>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[tw_ty(I, c)] = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[tw_ta(a, I)][r], acc[tw_ty(a, c)][r], acc[tw_ty(I, c)], 0, 0, 1);
                });
            });
            __builtin_amdgcn_wave_barrier();
        }
    });
    // factored rows back to the arena (upper part; the border column of the last block is y): rows past the last pivot untouched
    static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
        constexpr int a = decltype(ac)::value;
        static_for_t<a, 4>([&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * a + fq + 4 * r, c = 16 * b + fr;
                if (i < npiv && c >= i) blk[(long long)i * KP + c] = acc[tw_ta(a, b)][r];
            }
        });
    });
    // row-major R_jj^-1 = Y': (R^-1)[i][c] = Y[c][i]; identity in the rows / columns past the last pivot, zero below the diagonal
    double* rinvp = ws.rinv + (wl * ws.NSB + j) * (long long)(SB * SB);
    static_for_t<0, 4>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        static_for_t<0, 4>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
            if constexpr (a <= b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int yr = 16 * b + fq + 4 * r, yc = 16 * a + fr;           // Y[yr][yc] -> (R^-1)[yc][yr]
                    const double v = (yr < npiv) ? acc[tw_ty(b, a)][r] : ((yr == yc) ? 1.0 : 0.0);
                    rinvp[yc * SB + yr] = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) rinvp[(16 * a + fq + 4 * r) * SB + 16 * b + fr] = 0.0;   // (R^-1) tile (a, b), a > b
            }
        });
    });
    const bool bad = __any((badacc != badacc) ? 1 : 0) != 0;
    if (lane == 0 && bad) ws.flags[wl] = 1;
}
