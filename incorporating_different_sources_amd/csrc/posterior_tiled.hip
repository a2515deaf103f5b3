// posterior_tiled.hip - large-k path of the posterior (k beyond the register-tile kernel: S&P500-sized
// and larger universes, BASELINE configs k=500 and k=1000).  gfx950 only.
//
// The bordered posterior matrix [[S1, b], [b', .]] (b = c S0 w0 + t) of every in-flight window lives
// in an HBM arena as a row-major KP x KP square, KP = 64 * NS; work is cut into 64 x 64 super-tiles
// (4 x 4 f64 MFMA tiles, one workgroup of 4 wavefronts each).  ref:LINE cites
// /root/reference/src/portfolio_calculations.py.
//
//   tiled_prior_kernel   per window: intraday column means (ref:317), z = (Y - ybar) w0, q0 = s z'z,
//                        c (ref:415-418); leaves ybar and zc = c sqrt(s) z in the workspace
//   tile64_kernel<GRAM>  one super-tile of  s (Y-ybar)'(Y-ybar) + X'X  in ONE pass over the rows: the
//                        intraday rows are staged scaled by sqrt(s) with c sqrt(s) z_r in the border
//                        column, the daily rows with 1 in the border column (ref:180, 222, 333, 358, 489)
//   tiled_diag_kernel    block step j: Cholesky of the 64 x 64 diagonal block in LDS (upper, R'R) and
//                        its inverse R_jj^-1 (ref:485's inverse is never formed for the full matrix)
//   tile64_kernel<TRSM>  R_jJ = R_jj^-T A_jJ as an MFMA product with R_jj^-1 as the k-major A image
//   tile64_kernel<SYRK>  left-looking update of block row j: A_jJ -= sum_{q<j} R_qj' R_qJ, J >= j
//   tiled_solve_kernel   y = border column (forward substitution happened on the way), q1 = y'y,
//                        blocked back substitution, weights (ref:572-575, 836 / 849), status, aux
//
// Jeffreys (ref:600-606): J = T - t t'/N is applied as a rank-one correction by tile64_kernel<RANK1>
// after the Gram pass (t is the border column).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "posterior_kernels.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SB = 64;                 // super-block edge
constexpr int CH = 16;                 // staged rows per chunk
constexpr int LDX = 2 * SB + 16;       // LDS row stride of a staged chunk (A cols | B cols), 144 = 16 mod 32
constexpr int NTHREADS = 256;

enum { MODE_GRAM = 0, MODE_TRSM = 1, MODE_SYRK = 2, MODE_SYRK_DIAG = 3 };   // SYRK_DIAG: the update of tile (j, j) alone

template <int N>
__device__ __forceinline__ double dpp_row_ror(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x120 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x120 + N, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rowgroup_sum16(double v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    return v;
}
__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// upper-triangle pair p -> (a, b), a <= b < n, row-major
__device__ __forceinline__ void pair_decode(int p, int n, int& a, int& b) {
    int i = 0, rem = p;
    while (rem >= n - i) { rem -= n - i; ++i; }
    a = i; b = i + rem;
}

// inverse of pair_decode: row-major number of the upper-triangle pair (a, b), a <= b < n
__device__ __forceinline__ long long pair_index(int a, int b, int n) { return (long long)a * n - (long long)a * (a - 1) / 2 + (b - a); }

// XCD-aware mapping of a 1-D grid to (window, tile) for the kernels that run SEVERAL workgroups per window.
// MI355X deals consecutive workgroup ids round-robin to its 8 XCDs, each with its own 4 MB L2.  With
// (window, tile) = (blockIdx.x, blockIdx.y) the tiles of one window ran on one XCD but thousands of workgroups
// apart, so every tile re-read the window's rows (or arena tiles) from HBM / Infinity Cache: the Gram kernel
// moved 48 GB per launch at k = 500.  Here id -> xcd = id % 8, slot = id / 8, window = 8 (slot / NT) + xcd,
// tile = slot % NT: the NT tiles of a window are consecutive ON ONE XCD, whose L2 then serves the re-reads.
__device__ __forceinline__ bool xcd_window_tile(int NT, long long G, long long& wl, int& tile) {
    const long long id = blockIdx.x;
    const long long slot = id >> 3;
    wl = 8 * (slot / NT) + (id & 7);
    tile = (int)(slot % NT);
    return wl < G;
}
inline dim3 xcd_grid(int NT, long long G) { return dim3((unsigned)(((G + 7) / 8) * 8 * NT)); }

// ------------------------------------------------------------------------------------------------
// ONE pass over the intraday window (the kernel is HBM-bound: the window is 1.5 MB at k=500, 13.7 MB at k=1000,
// used by nothing else): a wavefront per row, lane l holds columns l, l+64, ... - the column sums and w0 stay in
// registers (NCI = ceil(k/64) of each, a template parameter so that the indices are static), and the same
// loaded values give u_r = y_r . w0.  Then z_r = u_r - ybar . w0  (= (y_r - ybar) . w0 of ref:317-333 with the
// subtraction taken out of the sum).
template <int NCI>
__global__ void __launch_bounds__(NTHREADS) tiled_prior_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws) {
    __shared__ double part[4][64 * NCI];
    __shared__ double red[4];
    __shared__ double sc[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long wl = blockIdx.x;                 // window inside the batch
    const long long w = A.w_first + wl;
    const int k = A.k;
    const int mm = A.hf_count ? A.hf_count[w] : A.m;
    const int* cols = A.col_idx ? A.col_idx + w * k : nullptr;
    const int* ridx = A.hf_row_idx ? A.hf_row_idx + w * (long long)A.m : nullptr;
    const long long first = A.hf_start ? A.hf_start[w] : 0;
    double* ybar = ws.ybar + wl * ws.KP;
    double* zc = ws.zc + wl * (long long)A.m;
    int gc[NCI];
    double w0r[NCI], cs[NCI];
#pragma unroll
    for (int i = 0; i < NCI; ++i) {
        const int c = lane + 64 * i;
        const int cl = c < k ? c : k - 1;                               // padding lanes re-read column k-1, weight 0
        gc[i] = cols ? cols[cl] : cl;
        w0r[i] = c < k ? A.w0[w * k + c] : 0.0;
        cs[i] = 0.0;
    }
    for (int r = wv; r < mm; r += 4) {                                  // rows r = wv (mod 4), in order
        const long long row = ridx ? (long long)ridx[r] : first + r;
        const double* p = A.hf_panel + row * (long long)A.hf_ld;
        double y[NCI];
#pragma unroll
        for (int i = 0; i < NCI; ++i) y[i] = p[gc[i]];
        double u = 0.0;
#pragma unroll
        for (int i = 0; i < NCI; ++i) { cs[i] += y[i]; u = fma(y[i], w0r[i], u); }
        u = wave_sum64(u);
        if (lane == 0) zc[r] = u;
    }
#pragma unroll
    for (int i = 0; i < NCI; ++i) part[wv][lane + 64 * i] = cs[i];
    __syncthreads();
    // column means (fixed order over the four row classes) and ybar . w0
    double yw = 0.0;
    for (int c = tid; c < ws.KP; c += NTHREADS) {
        double m = 0.0;
        if (c < k) {
            m = (((part[0][c] + part[1][c]) + part[2][c]) + part[3][c]) / (double)mm;
            yw = fma(m, A.w0[w * k + c], yw);
        }
        ybar[c] = m;
    }
    yw = wave_sum64(yw);
    if (lane == 0) red[wv] = yw;
    __syncthreads();
    const double ybw = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    double zz = 0.0;
    for (int r = tid; r < mm; r += NTHREADS) {
        const double z = zc[r] - ybw;
        zc[r] = z;
        zz = fma(z, z, zz);
    }
    zz = wave_sum64(zz);
    if (lane == 0) red[wv] = zz;
    __syncthreads();
    if (tid == 0) {
        const double n0 = A.n0[w];
        const double s = n0 * ((double)mm / ((double)mm - 1.0));
        const double q0 = s * (((red[0] + red[1]) + red[2]) + red[3]);
        const double a = n0 + k + 2;
        const double c = (2 * n0) / (a + sqrt(a * a + 4 * n0 * q0));
        double* o = ws.scal + wl * 8;
        o[0] = s; o[1] = sqrt(s); o[2] = c; o[3] = q0; o[4] = n0;
        sc[0] = c * sqrt(s);
    }
    __syncthreads();
    const double f = sc[0];
    for (int r = tid; r < mm; r += NTHREADS) zc[r] *= f;    // border-column entries of the staged intraday rows
}

// ------------------------------------------------------------------------------------------------
// One 64 x 64 super-tile: D[i][j] (+/-)= sum_r Aimg[r][i] Bimg[r][j] over the rows the mode provides.
template <int MODE>
__global__ void __launch_bounds__(NTHREADS) tile64_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws, const int j) {
    __shared__ __attribute__((aligned(16))) double lds[2 * CH * LDX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int k = A.k, KP = ws.KP, NS = ws.NS;
    const int T = NS - 1 - j;                                  // TRSM: block columns to the right of step j
    long long wl;
    int tile;
    if (!xcd_window_tile(MODE == MODE_GRAM ? NS * (NS + 1) / 2 : (MODE == MODE_TRSM ? T : (MODE == MODE_SYRK_DIAG ? 1 : T + 1)), A.w_count, wl, tile))
        return;
    const long long w = A.w_first + wl;
    double* M = ws.arena + wl * (long long)KP * KP;

    int SI, SJ;
    if (MODE == MODE_GRAM) pair_decode(tile, NS, SI, SJ);
    else if (MODE == MODE_TRSM) { SI = j; SJ = j + 1 + tile; }
    else { SI = j; SJ = j + tile; }             // SYRK (left-looking): tile (j, J), J >= j, gets ALL earlier block rows

    // row sources
    const int* cols = A.col_idx ? A.col_idx + w * k : nullptr;
    int mm = 0, nr = 0;
    const int* hridx = nullptr; const int* dridx = nullptr;
    long long hfirst = 0, dfirst = 0;
    const double* rf = nullptr;
    double sqs = 0.0;
    const double* ybar = ws.ybar + wl * KP;
    const double* zc = ws.zc + wl * (long long)A.m;
    if (MODE == MODE_GRAM) {
        if (A.strategy == 0) {
            mm = A.hf_count ? A.hf_count[w] : A.m;
            hridx = A.hf_row_idx ? A.hf_row_idx + w * (long long)A.m : nullptr;
            hfirst = A.hf_start ? A.hf_start[w] : 0;
            sqs = ws.scal[wl * 8 + 1];
        }
        nr = A.n_rows ? A.n_rows[w] : A.n_r;
        dridx = A.row_idx ? A.row_idx + w * (long long)A.n_r : nullptr;
        dfirst = A.start ? A.start[w] : 0;
        rf = A.rf_adj ? A.rf_adj + w * (long long)A.n_r : nullptr;
    }
    const double* rinv = ws.rinv + (wl * ws.NSB + j) * (long long)(SB * SB);

    d4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    if (MODE == MODE_SYRK || MODE == MODE_SYRK_DIAG) {           // C - R'R: start from C
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[b][r] = M[(long long)(64 * SI + 16 * wv + fq + 4 * r) * KP + 64 * SJ + 16 * b + fr];
    }

    // Staging geometry: 16 threads per row, thread (srow, cb) handles local columns cb + 16 i of the
    // A half (i < 4) and of the B half (i >= 4).  Per-thread column constants are fixed for the whole
    // kernel; the chunk loop issues RAW loads only (no select, no arithmetic on a loaded value before
    // the MFMA block) so that the next chunk's loads stay in flight under the MFMAs.
    const int srow = tid >> 4, cb = tid & 15;
    int pcol[8];            // GRAM: panel column (clamped) ; TRSM/SYRK: arena / rinv column
    double yb[8];           // GRAM: intraday column mean of that column
    bool cval[8], cbord[8]; // GRAM: real asset column / border column
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = cb + 16 * (i & 3);
        if (MODE == MODE_GRAM) {
            const int gc = 64 * ((i >> 2) ? SJ : SI) + c;
            cval[i] = gc < k;
            cbord[i] = gc == k;
            const int gcl = cval[i] ? gc : k - 1;
            pcol[i] = gcl;
            yb[i] = cval[i] ? ybar[gcl] : 0.0;
        } else {
            cval[i] = true; cbord[i] = false; yb[i] = 0.0;
            pcol[i] = (MODE == MODE_TRSM && (i >> 2) == 0) ? c : 64 * ((i >> 2) ? SJ : SI) + c;
        }
    }
    if (MODE == MODE_GRAM && cols) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pcol[i] = cols[pcol[i]];
    }
    const bool interior = MODE == MODE_GRAM && 64 * SJ + 63 < k;     // SI <= SJ: both column groups are real assets
    // GRAM rows: [intraday rows, padded to whole chunks][daily rows]; a chunk is purely one kind
    const int hchunks = (mm + CH - 1) / CH;
    const int nchunks = (MODE == MODE_GRAM) ? hchunks + (nr + CH - 1) / CH : ((MODE == MODE_SYRK || MODE == MODE_SYRK_DIAG) ? (SB / CH) * j : SB / CH);
    double v[8];
    double rowc = 0.0;      // per-row constant: border entry (intraday) / risk-free adjustment (daily)
    bool rowv = false;
    auto load = [&](int ch) {
        if (MODE == MODE_GRAM) {
            const bool hf = ch < hchunks;
            const int r = (hf ? ch : ch - hchunks) * CH + srow;
            const int cnt = hf ? mm : nr;
            rowv = r < cnt;
            const int rc = rowv ? r : cnt - 1;
            const int* ridx = hf ? hridx : dridx;
            const bool o32 = ((hf ? A.hf_off32 : A.panel_off32) & (ridx ? 1 : 2)) != 0;
            if (hf) rowc = zc[rc];
            else rowc = rf ? rf[rc] : 0.0;
            if (o32) {
                // one 32-bit offset per row on a uniform base (scalar registers) + the lane's column offsets:
                // vector instructions are what this kernel runs out of (see posterior_fused_impl.h, staging)
                const int ld = hf ? A.hf_ld : A.panel_ld;
                const double* pb = hf ? A.hf_panel : A.panel;
                const char* ub = (const char*)(ridx ? pb : pb + (hf ? hfirst : dfirst) * (long long)ld);
                const unsigned ro = __umul24(ridx ? (unsigned)ridx[rc] : (unsigned)rc, (unsigned)ld * 8u);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *(const double*)(ub + (size_t)(ro + 8u * (unsigned)pcol[i]));
            } else {
                const double* base;
                if (hf) {
                    const long long row = hridx ? (long long)hridx[rc] : hfirst + rc;
                    base = A.hf_panel + row * (long long)A.hf_ld;
                } else {
                    const long long row = dridx ? (long long)dridx[rc] : dfirst + rc;
                    base = A.panel + row * (long long)A.panel_ld;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = base[pcol[i]];
            }
        } else {
            // TRSM: the 64 rows of block row j; SYRK: the rows of block rows 0 .. j-1, one after the other
            const int r = (MODE == MODE_SYRK || MODE == MODE_SYRK_DIAG) ? ch * CH + srow : 64 * j + ch * CH + srow;
            const double* rowA = (MODE == MODE_TRSM) ? rinv + (ch * CH + srow) * SB : M + (long long)r * KP;
            const double* rowB = M + (long long)r * KP;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = rowA[pcol[i]];
#pragma unroll
            for (int i = 4; i < 8; ++i) v[i] = rowB[pcol[i]];
        }
    };
    auto store = [&](double* buf, int ch) {
        if (MODE == MODE_GRAM) {
            const bool hf = ch < hchunks;
            const bool ragged = ((hf ? ch : ch - hchunks) + 1) * CH > (hf ? mm : nr);   // uniform
            if (interior) {
                // every column of this super-tile pair is a real asset: no column selects at all
                if (hf) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = sqs * (v[i] - yb[i]);            // sqrt(s) (y - ybar)
                } else if (rf) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] -= rowc;                           // x - rf (ref:57)
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (hf) v[i] = cval[i] ? sqs * (v[i] - yb[i]) : (cbord[i] ? rowc : 0.0);   // ... | c sqrt(s) z_r
                    else v[i] = cval[i] ? v[i] - rowc : (cbord[i] ? 1.0 : 0.0);                // ... | 1
                }
            }
            if (ragged) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = rowv ? v[i] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) buf[srow * LDX + 64 * (i >> 2) + cb + 16 * (i & 3)] = v[i];
    };
    if (nchunks > 0) { load(0); store(lds, 0); }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        double* cur = lds + (ch & 1) * CH * LDX;
        double* nxt = lds + ((ch + 1) & 1) * CH * LDX;
        const bool more = ch + 1 < nchunks;
        if (more) load(ch + 1);
        __builtin_amdgcn_sched_barrier(0);
        const double* lb = cur + fq * LDX + fr;
#pragma unroll
        for (int s4 = 0; s4 < CH / 4; ++s4) {
            const double a = lb[4 * s4 * LDX + 16 * wv];
#pragma unroll
            for (int b = 0; b < 4; ++b)      // SYRK: acc -= a'b through the negate bit of the f64 MFMA (BLGP bit 0)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lb[4 * s4 * LDX + 64 + 16 * b], acc[b], 0, 0,
                                                              (MODE == MODE_SYRK || MODE == MODE_SYRK_DIAG) ? 1 : 0);
        }
        if (more) store(nxt, ch + 1);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            M[(long long)(64 * SI + 16 * wv + fq + 4 * r) * KP + 64 * SJ + 16 * b + fr] = acc[b][r];
}

// ------------------------------------------------------------------------------------------------
// The Gram super-tile again, for panels whose byte offsets fit 32 bits (every realistic one), written for the
// vector-instruction budget (see posterior_fused_impl.h, staging): the column selects of the padded / border
// columns exist only in EDGE super-tiles (the instantiation is picked per workgroup, uniformly), the intraday
// and the daily chunks are two straight-line code paths, addresses are one 32-bit row offset plus eight
// loop-invariant column offsets on a uniform base.  Same loads, same arithmetic, same order as
// tile64_kernel<MODE_GRAM>, which stays as the path for panels of 4 GiB and more.
template <bool EDGE>
__device__ __forceinline__ void gram64_lean_body(const tp_kargs_t& A, const tp_tiled_ws_t& ws, double* lds,
                                                 const long long wl, const int SI, const int SJ) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const long long w = A.w_first + wl;
    const int k = A.k, KP = ws.KP;
    double* M = ws.arena + wl * (long long)KP * KP;
    const int* cols = A.col_idx ? A.col_idx + w * k : nullptr;
    const bool conj = A.strategy == 0;
    const int mm = conj ? (A.hf_count ? A.hf_count[w] : A.m) : 0;
    const int nr = A.n_rows ? A.n_rows[w] : A.n_r;
    const int* hridx = (conj && A.hf_row_idx) ? A.hf_row_idx + w * (long long)A.m : nullptr;
    const int* dridx = A.row_idx ? A.row_idx + w * (long long)A.n_r : nullptr;
    const double* rf = A.rf_adj ? A.rf_adj + w * (long long)A.n_r : nullptr;
    const double sqs = conj ? ws.scal[wl * 8 + 1] : 0.0;
    const double* ybar = ws.ybar + wl * KP;
    const double* zc = ws.zc + wl * (long long)A.m;
    // uniform bases: the window's first row (contiguous rows) or the panel (explicit rows)
    const char* hub = conj ? (const char*)(hridx ? A.hf_panel : A.hf_panel + (A.hf_start ? A.hf_start[w] : 0) * (long long)A.hf_ld) : nullptr;
    const char* dub = (const char*)(dridx ? A.panel : A.panel + (A.start ? A.start[w] : 0) * (long long)A.panel_ld);
    const unsigned hld8 = (unsigned)A.hf_ld * 8u, dld8 = (unsigned)A.panel_ld * 8u;

    const int srow = tid >> 4, cb = tid & 15;
    unsigned co[8];          // byte offset of this lane's eight columns (A half: i < 4, B half: i >= 4)
    double yb[8];            // intraday column means of those columns
    bool cval[8], cbord[8];  // EDGE only: real asset column / border column
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gc = 64 * ((i >> 2) ? SJ : SI) + cb + 16 * (i & 3);
        cval[i] = !EDGE || gc < k;
        cbord[i] = EDGE && gc == k;
        const int gcl = cval[i] ? gc : k - 1;                      // padding columns re-read column k-1 (masked below)
        co[i] = 8u * (unsigned)(cols ? cols[gcl] : gcl);
        yb[i] = (conj && cval[i]) ? ybar[gcl] : 0.0;
    }
    d4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = d4{0.0, 0.0, 0.0, 0.0};

    // Shared running sums of the daily panel (tiled_prefix_kernel; posterior_fused_impl.h has the register-tile
    // form): the window's whole aligned CH-row blocks come as a difference of two sums every window shares, only the
    // rows in front of the first whole block (`lo` of them) and behind the last one are staged here.
    const long long dfirst = A.start ? A.start[w] : 0;
    const long long pb0 = (dfirst + CH - 1) / CH, pb1 = (dfirst + nr) / CH;
    const int Lw = (int)(pb1 - pb0);         // whole blocks of this window; the host planned a table for its count
    const int li = Lw == A.winsum_L[0] ? 0 : Lw == A.winsum_L[1] ? 1 : Lw == A.winsum_L[2] ? 2 : Lw == A.winsum_L[3] ? 3 : -1;
    const long long tb0 = pb0 - A.prefix_blk0;                    // position in the tables of the sub-batch in flight
    const bool shared = A.winsum != nullptr && !dridx && Lw > 0 && li >= 0 && tb0 >= 0 && tb0 + Lw <= A.prefix_nblk;
    const int lo = shared ? (int)(CH * pb0 - dfirst) : nr;            // staged daily rows r < lo: window rows r
    const int djump = shared ? (int)(CH * pb1 - dfirst) - lo : 0;     //                  r >= lo: window rows r + djump
    const int nrs = shared ? lo + (int)(dfirst + nr - CH * pb1) : nr; // staged daily rows
    const int hchunks = (mm + CH - 1) / CH;
    const int nchunks = hchunks + (nrs + CH - 1) / CH;
    double v[8];
    double rowc = 0.0;      // per-row constant: border entry (intraday) / risk-free adjustment (daily)
    auto load = [&](int ch) __attribute__((always_inline)) {
        if (ch < hchunks) {
            const int r = ch * CH + srow;
            const int rc = r < mm ? r : mm - 1;
            const unsigned ro = __umul24(hridx ? (unsigned)hridx[rc] : (unsigned)rc, hld8);
            rowc = zc[rc];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *(const double*)(hub + (size_t)(ro + co[i]));
        } else {
            const int r = (ch - hchunks) * CH + srow;
            int rc = r < nrs ? r : nrs - 1;
            rc += rc >= lo ? djump : 0;
            const unsigned ro = __umul24(dridx ? (unsigned)dridx[rc] : (unsigned)rc, dld8);
            rowc = rf ? rf[rc] : 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *(const double*)(dub + (size_t)(ro + co[i]));
        }
    };
    auto store = [&](double* buf, int ch) __attribute__((always_inline)) {
        if (ch < hchunks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = sqs * (v[i] - yb[i]);                                       // sqrt(s) (y - ybar)
                if (EDGE) v[i] = cval[i] ? v[i] : (cbord[i] ? rowc : 0.0);         // border: c sqrt(s) z_r
            }
            if ((ch + 1) * CH > mm) {                                              // ragged chunk (uniform)
                const bool rowv = ch * CH + srow < mm;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = rowv ? v[i] : 0.0;
            }
        } else {
            if (rf) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] -= rowc;                          // x - rf (ref:57)
            }
            if (EDGE) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = cval[i] ? v[i] : (cbord[i] ? 1.0 : 0.0);   // border: ones -> t
            }
            if ((ch - hchunks + 1) * CH > nrs) {
                const bool rowv = (ch - hchunks) * CH + srow < nrs;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = rowv ? v[i] : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) buf[srow * LDX + 64 * (i >> 2) + cb + 16 * (i & 3)] = v[i];
    };
    if (nchunks > 0) { load(0); store(lds, 0); }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        double* cur = lds + (ch & 1) * CH * LDX;
        double* nxt = lds + ((ch + 1) & 1) * CH * LDX;
        const bool more = ch + 1 < nchunks;
        if (more) load(ch + 1);
        __builtin_amdgcn_sched_barrier(0);
        const double* lb = cur + fq * LDX + fr;
#pragma unroll
        for (int s4 = 0; s4 < CH / 4; ++s4) {
            const double a = lb[4 * s4 * LDX + 16 * wv];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lb[4 * s4 * LDX + 64 + 16 * b], acc[b], 0, 0, 0);
        }
        if (more) store(nxt, ch + 1);
        __syncthreads();
    }
    if (shared) {
        // ONE table slot: Q_L[b0] = the Gram of the window's L whole blocks (tp_window_sums_kernel)
        const long long ntile = (long long)ws.NS * (ws.NS + 1) / 2;
        const long long tile = pair_index(SI, SJ, ws.NS);
        // [..][wave][16-column group b][2][64 lanes][2]: registers (0,1) and (2,3) of a lane are 16 contiguous bytes
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* q = (const d2*)(A.winsum + (((long long)li * A.prefix_nblk + tb0) * ntile + tile) * (SB * SB) + wv * 1024) + lane;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const d2 v2 = q[(b * 2 + h) * 64];
                acc[b][2 * h] += v2[0];
                acc[b][2 * h + 1] += v2[1];
            }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            M[(long long)(64 * SI + 16 * wv + fq + 4 * r) * KP + 64 * SJ + 16 * b + fr] = acc[b][r];
}

// Grams of the aligned CH-row blocks of the daily panel for the tiled path: one workgroup per (block, 64 x 64
// super-tile), staged like the daily half of the Gram kernel (ones in the border column, zeros beyond).  Layout
// [block][super-tile pair][wave][16-column group][2][64 lanes][2]; tp_window_sums_launch adds them up to the block-window
// sums Q_L the windows read (posterior_fused_impl.h has the register-tile form of the same scheme).
template <bool EDGE>
__device__ __forceinline__ void blockgram64_body(const tp_kargs_t& A, const tp_tiled_ws_t& ws, double* lds, double* out,
                                                 const long long blk, const long long tile, const int SI, const int SJ) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int k = A.k;
    const double* row = A.panel + ((blk + A.prefix_blk0) * CH + (tid >> 4)) * (long long)A.panel_ld;
    const int cb = tid & 15;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gc = 64 * ((i >> 2) ? SJ : SI) + cb + 16 * (i & 3);
        const bool cval = !EDGE || gc < k;
        const double x = row[cval ? gc : k - 1];
        v[i] = cval ? x : ((EDGE && gc == k) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) lds[(tid >> 4) * LDX + 64 * (i >> 2) + cb + 16 * (i & 3)] = v[i];
    __syncthreads();
    d4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    const double* lb = lds + fq * LDX + fr;
#pragma unroll
    for (int s4 = 0; s4 < CH / 4; ++s4) {
        const double a = lb[4 * s4 * LDX + 16 * wv];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lb[4 * s4 * LDX + 64 + 16 * b], acc[b], 0, 0, 0);
    }
    const long long ntile = (long long)ws.NS * (ws.NS + 1) / 2;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2* p = (d2*)(out + (blk * ntile + tile) * (SB * SB) + wv * 1024) + lane;     // 16-byte stores
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) p[(b * 2 + h) * 64] = d2{acc[b][2 * h], acc[b][2 * h + 1]};
}

__global__ void __launch_bounds__(NTHREADS) tiled_block_gram_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws, double* out) {
    __shared__ __attribute__((aligned(16))) double lds[CH * LDX];
    const long long ntile = (long long)ws.NS * (ws.NS + 1) / 2;
    const long long blk = blockIdx.x / ntile;
    const int tile = (int)(blockIdx.x % ntile);
    int SI, SJ;
    pair_decode(tile, ws.NS, SI, SJ);
    if (64 * SJ + 63 < A.k) blockgram64_body<false>(A, ws, lds, out, blk, tile, SI, SJ);
    else blockgram64_body<true>(A, ws, lds, out, blk, tile, SI, SJ);
}

__global__ void __launch_bounds__(NTHREADS) tiled_gram_lean_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws) {
    __shared__ __attribute__((aligned(16))) double lds[2 * CH * LDX];
    long long wl;
    int tile, SI, SJ;
    if (!xcd_window_tile(ws.NS * (ws.NS + 1) / 2, A.w_count, wl, tile)) return;
    pair_decode(tile, ws.NS, SI, SJ);
    if (64 * SJ + 63 < A.k) gram64_lean_body<false>(A, ws, lds, wl, SI, SJ);     // SI <= SJ: every column is a real asset
    else gram64_lean_body<true>(A, ws, lds, wl, SI, SJ);
}

// Jeffreys: J = T - t t'/N on one super-tile (t = border column k); rows/cols >= k untouched.
__global__ void __launch_bounds__(NTHREADS) tiled_rank1_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws) {
    const int tid = threadIdx.x;
    const int k = A.k, KP = ws.KP, NS = ws.NS;
    long long wl;
    int tile, SI, SJ;
    if (!xcd_window_tile(NS * (NS + 1) / 2, A.w_count, wl, tile)) return;
    double* M = ws.arena + wl * (long long)KP * KP;
    pair_decode(tile, NS, SI, SJ);
    const long long w = A.w_first + wl;
    const double invN = A.center_rows == 2 ? 0.0
                      : 1.0 / (double)(A.center_rows ? (A.n_rows ? A.n_rows[w] : A.n_r) : A.N);
    const double sh_d = A.shift ? A.shift[2 * w] : 0.0;         // tp_batch_set_shift: + d I + e 1 1'
    const double sh_e = A.shift ? A.shift[2 * w + 1] : 0.0;
    for (int e = tid; e < SB * SB; e += NTHREADS) {
        const int gi = 64 * SI + (e >> 6), gj = 64 * SJ + (e & 63);
        if (gi < k && gj < k) {
            const double ti = M[(long long)gi * KP + k], tj = M[(long long)gj * KP + k];
            M[(long long)gi * KP + gj] += sh_e + (gi == gj ? sh_d : 0.0) - invN * (ti * tj);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Block step j: upper Cholesky of the 64 x 64 diagonal block in LDS and R_jj^-1.  Rows >= npiv (the
// last block only: border row and padding) behave as identity rows.
// compile-time loop (the pivot loop below needs static register indices)
template <int I, int N, class F>
__device__ __forceinline__ void static_for_t(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_t<I + 1, N>(f);
    }
}

__global__ void __launch_bounds__(NTHREADS) tiled_diag_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws, const int j) {
    constexpr int LD = SB + 1;
    __shared__ double Tt[SB * LD];             // transpose buffer for R^-1 on the way out
    __shared__ double rowp[2][2 * SB];         // the pivot row (block | identity part), published by its owners
    __shared__ int bad;
    const int tid = threadIdx.x;
    const long long wl = blockIdx.x;
    const int k = A.k, KP = ws.KP;
    double* M = ws.arena + wl * (long long)KP * KP;
    const int npiv = (k - 64 * j < SB) ? (k - 64 * j) : SB;
    if (tid == 0) bad = 0;
    // Right-looking Cholesky of the 64 x 64 block with the block AND 64 identity columns in REGISTERS: thread
    // (g, c) holds column c of rows g, g+4, ... of both (2 x 16 values).  Per pivot the owners publish the
    // (unscaled) pivot row through LDS, ONE barrier, then every thread updates its own registers from LDS
    // broadcasts of the multipliers.  The identity columns come out as R^-T (as in the register-tile kernel's
    // phase F), i.e. R_jj^-1 needs no pass of its own; rows >= npiv (border row, padding) are never touched and
    // stay identity rows.  Fully unrolled so that register indices are static.
    const int c = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6);
    double a[16], m[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        a[r] = M[(long long)(64 * j + 4 * r + g) * KP + 64 * j + c];
        m[r] = (4 * r + g == c) ? 1.0 : 0.0;
    }
    static_for_t<0, 64>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value, r = p >> 2, g4 = p & 3;
        if (p < npiv) {                                                  // uniform
            if (g == g4) { rowp[p & 1][c] = a[r]; rowp[p & 1][SB + c] = m[r]; }
            __syncthreads();
            const double d = rowp[p & 1][p];
            if (!(d > 0.0) && tid == 0) bad = 1;
            double rinv = __builtin_amdgcn_rsq(d);                       // 1/sqrt(d): seed + two Newton steps
            double e = fma(-(d * rinv), rinv, 1.0);
            rinv = fma(0.5 * rinv, e, rinv);
            e = fma(-(d * rinv), rinv, 1.0);
            rinv = fma(0.5 * rinv, e, rinv);
            const double sc = rowp[p & 1][c] * rinv;                     // R[p][c]
            const double si = rowp[p & 1][SB + c] * rinv;                // (R^-T)[p][c]
            if (g == g4) { a[r] = (c >= p) ? sc : 0.0; m[r] = si; }      // row p is final (all 64 columns: the border rides along)
#pragma unroll
            for (int rr = r; rr < 16; ++rr) {
                const int i = 4 * rr + g;                                // wave-uniform row
                if (i > p && i < npiv) {
                    const double mu = rowp[p & 1][i] * rinv;             // LDS broadcast
                    if (c >= i) a[rr] = fma(-mu, sc, a[rr]);
                    m[rr] = fma(-mu, si, m[rr]);
                }
            }
        }
    });
    // factored rows back to the arena (upper part; the border column of the last block is y)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 4 * r + g;
        if (i < npiv && c >= i) M[(long long)(64 * j + i) * KP + 64 * j + c] = a[r];
        Tt[i * LD + c] = m[r];                                           // (R^-T)[i][c] = (R^-1)[c][i]
    }
    __syncthreads();
    double* rinvp = ws.rinv + (wl * ws.NSB + j) * (long long)(SB * SB);
    for (int e = tid; e < SB * SB; e += NTHREADS) {
        const int i = e >> 6, cc = e & 63;
        rinvp[e] = Tt[cc * LD + i];                                      // row-major R^-1
    }
    if (tid == 0 && bad) ws.flags[wl] = 1;
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NTHREADS) tiled_solve_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws) {
    extern __shared__ __attribute__((aligned(16))) double sm[];     // wvec[KP] | zv[64]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long wl = blockIdx.x;
    const long long w = A.w_first + wl;
    const int k = A.k, KP = ws.KP, NSB = ws.NSB;
    const double* M = ws.arena + wl * (long long)KP * KP;
    double* wvec = sm;
    double* zv = sm + KP;
    __shared__ double qs[4];
    __shared__ int anybad;
    if (tid == 0) anybad = 0;
    for (int c = tid; c < KP; c += NTHREADS) wvec[c] = 0.0;
    // q1 = y'y, y = border column
    double q = 0.0;
    for (int i = tid; i < k; i += NTHREADS) { const double y = M[(long long)i * KP + k]; q = fma(y, y, q); }
    q = wave_sum64(q);
    if (lane == 0) qs[wv] = q;
    __syncthreads();
    const double q1 = qs[0] + qs[1] + qs[2] + qs[3];

    const int srow = tid >> 4, cb = tid & 15;      // 16 lanes per row, 16 rows per pass
    for (int Jb = NSB - 1; Jb >= 0; --Jb) {
        const int npiv = (k - 64 * Jb < SB) ? (k - 64 * Jb) : SB;
        // z = y_Jb - sum_{c >= 64 (Jb+1)} R[row][c] w[c]
        for (int ps = 0; ps < 4; ++ps) {
            const int i = 16 * ps + srow;                 // local row
            const long long gi = 64 * Jb + i;
            double s = 0.0;
            if (i < npiv)
                for (int c = 64 * (Jb + 1) + cb; c < k; c += 16) s = fma(M[gi * KP + c], wvec[c], s);
            s = rowgroup_sum16(s);
            if (cb == 0) zv[i] = (i < npiv) ? M[gi * KP + k] - s : 0.0;
        }
        __syncthreads();
        // w_Jb = R_jj^-1 z
        const double* rinv = ws.rinv + (wl * ws.NSB + Jb) * (long long)(SB * SB);
        for (int ps = 0; ps < 4; ++ps) {
            const int i = 16 * ps + srow;
            double s = 0.0;
            for (int c = cb; c < SB; c += 16) s = fma(rinv[i * SB + c], zv[c], s);
            s = rowgroup_sum16(s);
            if (cb == 0 && i < npiv) wvec[64 * Jb + i] = s;
        }
        __syncthreads();
    }
    const bool conj = A.strategy == 0;
    const double n0 = conj ? ws.scal[wl * 8 + 4] : 0.0;
    const double n1 = n0 + (double)A.N;
    const double denom = n1 - q1;
    bool bad = false;
    for (int i = tid; i < k; i += NTHREADS) {
        const double wi = wvec[i];
        const double out = conj ? 1.0 / A.gamma * ((n1 + k + 2) * wi / denom) : 1.0 / A.gamma * wi;
        A.weights[w * k + i] = out;
        if (!isfinite(out)) bad = true;
    }
    if (bad) anybad = 1;
    __syncthreads();
    if (tid == 0) {
        int st = TP_KSTATUS_OK;
        if (ws.flags[wl]) st = TP_KSTATUS_NOT_PD;
        else if (anybad) st = TP_KSTATUS_NONFINITE;
        else if (conj && !(denom > 0.0)) st = TP_KSTATUS_BAD_DENOM;
        A.status[w] = st;
        if (A.aux) {
            double* ax = A.aux + w * 8;
            ax[0] = n0; ax[1] = conj ? n1 : 0.0; ax[2] = conj ? ws.scal[wl * 8 + 2] : 0.0;
            ax[3] = conj ? ws.scal[wl * 8 + 3] : 0.0; ax[4] = q1; ax[5] = conj ? denom : 0.0; ax[6] = 0.0; ax[7] = 0.0;
        }
    }
}

// rows >= k of the bordered matrix are never pivots: clear the border ROW (it holds 1'X etc.) so the
// last diagonal block sees zero rows there; also zero the not-positive-definite flags
__global__ void __launch_bounds__(NTHREADS) tiled_clear_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws) {
    const long long wl = blockIdx.x;
    const int k = A.k, KP = ws.KP;
    double* M = ws.arena + wl * (long long)KP * KP;
    if (A.strategy == 0 && A.hf_winsum != nullptr) {
        // shared intraday sums (posterior_tiled_wave.h): S0 w0 from the super-tiles' pieces in a fixed order, then
        // q0 = w0' S0 w0, c (ref:415-418) and c S0 w0 into the border column, where the two-pass form's Gram puts it
        __shared__ double red[4];
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const long long w = A.w_first + wl;
        const int NS = ws.NS;
        const double* part = ws.part + (wl * NS * NS) * 64;
        double* v0 = ws.ybar + wl * KP;
        double qq = 0.0;
        for (int i = tid; i < k; i += NTHREADS) {
            const double* pp = part + ((long long)(i >> 6) * NS) * 64 + (i & 63);
            double v = 0.0;
            for (int src = 0; src < NS; ++src) v += pp[src * 64];
            v0[i] = v;
            qq = fma(A.w0[w * k + i], v, qq);
        }
        qq = wave_sum64(qq);
        if (lane == 0) red[wv] = qq;
        __syncthreads();
        const double q0 = ((red[0] + red[1]) + red[2]) + red[3];
        const double n0 = A.n0[w];
        const double mm = (double)A.m;
        const double s = n0 * (mm / (mm - 1.0));
        const double a = n0 + k + 2;
        const double c = (2 * n0) / (a + sqrt(a * a + 4 * n0 * q0));
        for (int i = tid; i < k; i += NTHREADS) M[(long long)i * KP + k] += c * v0[i];
        if (tid == 0) {
            double* o = ws.scal + wl * 8;
            o[0] = s; o[1] = sqrt(s); o[2] = c; o[3] = q0; o[4] = n0;
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < (KP - k) * KP; e += NTHREADS) M[(long long)k * KP + e] = 0.0;
    if (A.rhs != nullptr)     // caller-supplied right-hand side in place of the border column
        for (int i = threadIdx.x; i < k; i += NTHREADS) M[(long long)i * KP + k] = A.rhs[(A.w_first + wl) * k + i];
    if (A.out_rhs != nullptr)
        for (int i = threadIdx.x; i < k; i += NTHREADS) A.out_rhs[(A.w_first + wl) * k + i] = M[(long long)i * KP + k];
    if (threadIdx.x == 0) ws.flags[wl] = 0;
}

}  // namespace

int tp_tiled_max_assets(void) { return 64 * 32 - 1; }

void tp_tiled_geometry(int k, int* KP, int* NS, int* NSB) {
    const int ns = (k + 1 + SB - 1) / SB;
    *NS = ns; *KP = ns * SB; *NSB = (k + SB - 1) / SB;
}

// ------------------------------------------------------------------------------------------------
// Block step j, the tiles (j, J), J > j, in ONE kernel: the left-looking update A_jJ -= sum_{q<j} R_qj' R_qJ and, on the tile
// still in registers, R_jJ = R_jj^-T A_jJ.  As two kernels (tile64_kernel<MODE_SYRK> then <MODE_TRSM>) every such tile was
// written to the arena and read back in between - the TRSM launch moved 1 GB per block step at k = 500 and ran at the HBM
// rate (9.5 % of the run).  Runs after tiled_diag*_kernel(j), which needs the updated tile (j, j) first (its own launch).
// Staging as in tile64_kernel; in the TRSM part the B half of chunk c (rows 16c..16c+15 of the tile) comes from the
// accumulators of wave c instead of from memory.
// chunks of operand loads in flight in the left-looking update.  2 (A/B build -DTP_SYRK_DEPTH=2, round 3: 136 instead of
// 112 registers, three instead of four workgroups per CU) measured 32.0 / 18.4 ms against 31.7 / 18.3 ms per 8,192 windows
// at k = 500 (conjugate / Jeffreys): the kernel is bound by HBM bytes (4.6 TB/s read + written), not by their latency
#ifndef TP_SYRK_DEPTH
#define TP_SYRK_DEPTH 1
#endif
__global__ void __launch_bounds__(NTHREADS) tile64_syrk_trsm_kernel(const tp_kargs_t A, const tp_tiled_ws_t ws, const int j) {
    __shared__ __attribute__((aligned(16))) double lds[2 * CH * LDX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int KP = ws.KP, NS = ws.NS;
    const int T = NS - 1 - j;
    long long wl;
    int tile;
    if (!xcd_window_tile(T, A.w_count, wl, tile)) return;
    double* M = ws.arena + wl * (long long)KP * KP;
    const int SI = j, SJ = j + 1 + tile;
    const double* rinv = ws.rinv + (wl * ws.NSB + j) * (long long)(SB * SB);
    d4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[b][r] = M[(long long)(64 * SI + 16 * wv + fq + 4 * r) * KP + 64 * SJ + 16 * b + fr];
    const int srow = tid >> 4, cb = tid & 15;
    double v[8];
    // ---- left-looking update: rows of block rows 0 .. j-1, A half = columns of super-tile j, B half = columns of super-tile J
    const int nchunks = (SB / CH) * j;
    auto load = [&](double (&vv)[8], int ch) __attribute__((always_inline)) {
        const double* row = M + (long long)(ch * CH + srow) * KP;
#pragma unroll
        for (int i = 0; i < 4; ++i) vv[i] = row[64 * SI + cb + 16 * i];
#pragma unroll
        for (int i = 0; i < 4; ++i) vv[4 + i] = row[64 * SJ + cb + 16 * i];
    };
    auto store = [&](double* buf, const double (&vv)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) buf[srow * LDX + 64 * (i >> 2) + cb + 16 * (i & 3)] = vv[i];
    };
    auto mma = [&](const double* cur) __attribute__((always_inline)) {
        const double* lb = cur + fq * LDX + fr;
#pragma unroll
        for (int s4 = 0; s4 < CH / 4; ++s4) {
            const double a = lb[4 * s4 * LDX + 16 * wv];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lb[4 * s4 * LDX + 64 + 16 * b], acc[b], 0, 0, 1);
        }
    };
    double* buf0 = lds;
#if TP_SYRK_DEPTH == 2
    double* buf1 = lds + CH * LDX;
    double v2[8];
    if (nchunks > 0) { load(v, 0); store(buf0, v); }
    if (nchunks > 1) load(v, 1);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {             // nchunks = 4 j: even
        if (ch + 2 < nchunks) load(v2, ch + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(buf0);
        store(buf1, v);                                    // chunk ch + 1
        __syncthreads();
        if (ch + 3 < nchunks) load(v, ch + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(buf1);
        if (ch + 2 < nchunks) store(buf0, v2);
        __syncthreads();
    }
#else
    if (nchunks > 0) { load(v, 0); store(buf0, v); }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        double* cur = lds + (ch & 1) * CH * LDX;
        double* nxt = lds + ((ch + 1) & 1) * CH * LDX;
        const bool more = ch + 1 < nchunks;
        if (more) load(v, ch + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(cur);
        if (more) store(nxt, v);
        __syncthreads();
    }
#endif
    // ---- R_jJ = R_jj^-T A_jJ: chunk c = rows 16c..16c+15 of R_jj^-1 (A half, from memory) and of the tile (B half, wave c's registers)
    d4 res[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) res[b] = d4{0.0, 0.0, 0.0, 0.0};
    auto load_t = [&](int c) __attribute__((always_inline)) {
        const double* row = rinv + (c * CH + srow) * SB;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = row[cb + 16 * i];
    };
    auto store_t = [&](double* buf, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[srow * LDX + cb + 16 * i] = v[i];
        if (wv == c) {             // this wave's 16 rows of the tile, in the layout of the staged rows
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) buf[(fq + 4 * r) * LDX + 64 + 16 * b + fr] = acc[b][r];
        }
    };
    load_t(0); store_t(lds, 0);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < SB / CH; ++c) {
        double* cur = lds + (c & 1) * CH * LDX;
        double* nxt = lds + ((c + 1) & 1) * CH * LDX;
        const bool more = c + 1 < SB / CH;
        if (more) load_t(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        const double* lb = cur + fq * LDX + fr;
#pragma unroll
        for (int s4 = 0; s4 < CH / 4; ++s4) {
            const double a = lb[4 * s4 * LDX + 16 * wv];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                res[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lb[4 * s4 * LDX + 64 + 16 * b], res[b], 0, 0, 0);
        }
        if (more) store_t(nxt, c + 1);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            M[(long long)(64 * SI + 16 * wv + fq + 4 * r) * KP + 64 * SJ + 16 * b + fr] = res[b][r];
}

#include "posterior_tiled_wave.h"

// Whole pipeline for windows [a.w_first, a.w_first + a.w_count) (a.w_count <= ws capacity), on `stream`.
size_t tp_tiled_slot_doubles(int k) {
    int KP, NS, NSB;
    tp_tiled_geometry(k, &KP, &NS, &NSB);
    return (size_t)(NS * (NS + 1) / 2) * SB * SB;
}

size_t tp_tiled_prefix_bytes(int k, long long panel_rows, int n_L, int* nblk_out) {
    const long long nblk = panel_rows / CH;
    if (nblk_out) *nblk_out = (int)nblk;
    return sizeof(double) * (size_t)nblk * (size_t)(1 + n_L) * tp_tiled_slot_doubles(k);
}

// the shared block Grams and block-window sums of a run, once per run
hipError_t tp_tiled_prefix_launch(const tp_kargs_t& a, const tp_tiled_ws_t& ws, hipStream_t stream) {
    const int NS = ws.NS;
    const bool conj = a.strategy == 0;
    const bool lean = (a.panel_off32 & (a.row_idx ? 1 : 2)) && (!conj || (a.hf_off32 & (a.hf_row_idx ? 1 : 2)));
    if (a.winsum == nullptr || !lean) return hipSuccess;
    hipLaunchKernelGGL(tiled_block_gram_kernel, dim3((unsigned)((long long)a.prefix_nblk * (NS * (NS + 1) / 2))), dim3(NTHREADS), 0,
                       stream, a, ws, (double*)a.prefix);
    int n_L = 0;
    while (n_L < TP_WINSUM_MAX_L && a.winsum_L[n_L] > 0) ++n_L;
    return tp_window_sums_launch(a.prefix, (double*)a.winsum, a.prefix_nblk, (size_t)(NS * (NS + 1) / 2) * SB * SB, a.winsum_L, n_L,
                                 stream, a.prefix_blk0);
}

hipError_t tp_tiled_launch(const tp_kargs_t& a_in, const tp_tiled_ws_t& ws, hipStream_t stream, bool build_prefix) {
    tp_kargs_t a = a_in;
    const int G = (int)a.w_count;
    if (G <= 0) return hipSuccess;
    const int NS = ws.NS, NSB = ws.NSB;
    const bool conj = a.strategy == 0;
    // 32-bit offsets for both panels in the layout they come in (explicit rows: bit 0, contiguous: bit 1)?
    const bool lean = (a.panel_off32 & (a.row_idx ? 1 : 2)) && (!conj || (a.hf_off32 & (a.hf_row_idx ? 1 : 2)));
    if (build_prefix) {
        hipError_t e = tp_tiled_prefix_launch(a, ws, stream);
        if (e != hipSuccess) return e;
    }
    // one wavefront per super-tile (posterior_tiled_wave.h) unless TP_TILED_WAVE=0 asks for the 4-wave kernels (A/B runs)
    const bool use_wave = a.opts.tiled_wave != 0;
    // shared intraday sums of this sub-batch (tangency_api.cpp plans them; the default one-wave kernels only)
    const bool hfs = conj && a_in.hf_winsum != nullptr && use_wave && a_in.opts.tiled_wave != 2;
    if (!hfs) a.hf_winsum = nullptr;
    if (hfs) {
        const long long ntile = (long long)NS * (NS + 1) / 2;
        hipLaunchKernelGGL(tiled_hf_block_gram_kernel, xcd_grid((int)ntile, a.hf_nblk), dim3(64), 0, stream, a, ws, (double*)a.hf_prefix);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = tp_window_sums_launch(a.hf_prefix, (double*)a.hf_winsum, a.hf_nblk, tp_tiled_slot_doubles(a.k), &a.hf_L, 1, stream);
        if (e != hipSuccess) return e;
    } else if (conj) {
        const int nci = (a.k + 63) / 64;
        if (nci <= 8) hipLaunchKernelGGL(tiled_prior_kernel<8>, dim3(G), dim3(NTHREADS), 0, stream, a, ws);
        else if (nci <= 16) hipLaunchKernelGGL(tiled_prior_kernel<16>, dim3(G), dim3(NTHREADS), 0, stream, a, ws);
        else hipLaunchKernelGGL(tiled_prior_kernel<32>, dim3(G), dim3(NTHREADS), 0, stream, a, ws);
    }
    if (hfs)
        hipLaunchKernelGGL(tiled_gram_wave_hfs_kernel, xcd_grid(NS * (NS + 1) / 2, G), dim3(64), 0, stream, a, ws);
    else if (a.opts.tiled_wave == 2) {        // 64 x 128 per wavefront (A/B: option tiled_wave = 2)
        int np = 0;
        for (int i = 0; i < NS; ++i) np += (NS - i + 1) / 2;
        hipLaunchKernelGGL(tiled_gram_wave_pair_kernel, xcd_grid(np, G), dim3(64), 0, stream, a, ws, np);
    } else if (use_wave && !conj)             // Jeffreys: the rank-one term J = T - t t'/N inside the Gram kernel
        hipLaunchKernelGGL(tiled_gram_wave_rank1_kernel, xcd_grid(NS * (NS + 1) / 2, G), dim3(64), 0, stream, a, ws);
    else if (use_wave)
        hipLaunchKernelGGL(tiled_gram_wave_kernel, xcd_grid(NS * (NS + 1) / 2, G), dim3(64), 0, stream, a, ws);
    else if (lean) hipLaunchKernelGGL(tiled_gram_lean_kernel, xcd_grid(NS * (NS + 1) / 2, G), dim3(NTHREADS), 0, stream, a, ws);
    else hipLaunchKernelGGL(tile64_kernel<MODE_GRAM>, xcd_grid(NS * (NS + 1) / 2, G), dim3(NTHREADS), 0, stream, a, ws, 0);
    if (!conj && !(use_wave && a.opts.tiled_wave != 2)) hipLaunchKernelGGL(tiled_rank1_kernel, xcd_grid(NS * (NS + 1) / 2, G), dim3(NTHREADS), 0, stream, a, ws);
    hipLaunchKernelGGL(tiled_clear_kernel, dim3(G), dim3(NTHREADS), 0, stream, a, ws);
    // Left-looking blocked Cholesky: block row j first receives the updates of ALL earlier block rows in one
    // pass (every arena tile is read and written once per factorisation, not once per block step), then its
    // diagonal block is factorised and the rest of the row is solved.
    for (int j = 0; j < NSB; ++j) {
        const int T = NS - 1 - j;
        // measured (TP_TILED_FUSE = 0 / 1 runs of round 2): fused +1.6..2 % at k = 500 (8 super-tiles per side), -1.5 % at
        // k = 1000 (16: the fused kernel's longer workgroups balance worse over the many tiles of a block row)
        const bool fused = a.opts.tiled_fuse >= 0 ? a.opts.tiled_fuse != 0 : NS <= 8;
        // the left-looking update: of the diagonal tile only (the rest of the row takes it together with its solve below), or of
        // the whole row (TP_TILED_FUSE=0: the three-kernel form, for A/B runs)
        const bool wave_diag = use_wave;
        // fused: the diagonal tile takes its update inside the one-wave diagonal-block kernel (or in a launch of its own in
        // front of the 4-wave one), the rest of the row together with its solve below
        if (j > 0 && fused && !wave_diag) hipLaunchKernelGGL(tile64_kernel<MODE_SYRK_DIAG>, xcd_grid(1, G), dim3(NTHREADS), 0, stream, a, ws, j);
        else if (j > 0 && !fused) hipLaunchKernelGGL(tile64_kernel<MODE_SYRK>, xcd_grid(T + 1, G), dim3(NTHREADS), 0, stream, a, ws, j);
        // the diagonal block by one wavefront per window (posterior_tiled_wave.h) unless TP_TILED_WAVE=0
        if (wave_diag && fused) hipLaunchKernelGGL(tiled_diag_wave_kernel<true>, dim3(G), dim3(64), 0, stream, a, ws, j);
        else if (wave_diag) hipLaunchKernelGGL(tiled_diag_wave_kernel<false>, dim3(G), dim3(64), 0, stream, a, ws, j);
        else hipLaunchKernelGGL(tiled_diag_kernel, dim3(G), dim3(NTHREADS), 0, stream, a, ws, j);
        if (T > 0) {
            if (fused) hipLaunchKernelGGL(tile64_syrk_trsm_kernel, xcd_grid(T, G), dim3(NTHREADS), 0, stream, a, ws, j);
            else hipLaunchKernelGGL(tile64_kernel<MODE_TRSM>, xcd_grid(T, G), dim3(NTHREADS), 0, stream, a, ws, j);
        }
    }
    const size_t smem = sizeof(double) * (size_t)(ws.KP + SB);
    hipLaunchKernelGGL(tiled_solve_kernel, dim3(G), dim3(NTHREADS), smem, stream, a, ws);
    return hipGetLastError();
}
