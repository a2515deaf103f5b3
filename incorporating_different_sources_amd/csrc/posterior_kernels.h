// posterior_kernels.h - internal interface between the C-ABI (tangency_api.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TP_KSTATUS_OK 0
#define TP_KSTATUS_NOT_PD 1
#define TP_KSTATUS_NONFINITE 2
#define TP_KSTATUS_BAD_DENOM 3

// Kernel-selection switches of a handle (A/B measurements and tests).  -1 = automatic.  Read from the environment once,
// in tp_create (TP_WAVE_KERNEL, TP_TILED_WAVE, TP_TILED_FUSE), changed through tp_set_option; the launchers read
// them from the batch's arguments, never from the environment.
struct tp_kopts_t {
    int wave_kernel = -1;   // register-tile path: 0 = multi-wave kernel, 1 = one-wave kernel (k <= 143), 2 = two-wave kernel
    int tiled_wave = -1;    // large-k path: 0 = the 4-wave Gram / diagonal-block kernels
    int tiled_fuse = -1;    // large-k path: 0 / 1 = three-kernel / fused left-looking update + solve
};

// Dynamic-LDS limit of a kernel: a per-DEVICE function attribute.  The one-process-all-GPUs mode launches the same
// kernel on several devices from several host threads, so "already set" is a bit per device ordinal (devices beyond
// 63 set it on every launch).
#include <atomic>
template <typename K>
inline hipError_t tp_allow_dynamic_lds(std::atomic<unsigned long long>& done_mask, K kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (bit && (done_mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    if (bit) done_mask.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

// Device-side view of one batch (all pointers are DEVICE pointers; optional ones may be null).
struct tp_kargs_t {
    const double* panel;
    const long long* start;
    const int* row_idx;
    const int* n_rows;
    const int* col_idx;
    const double* rf_adj;
    const double* hf_panel;
    const long long* hf_start;
    const int* hf_row_idx;
    const int* hf_count;
    const double* w0;
    const double* n0;
    const double* prefix; // optional (contiguous layout, no rf_adj): workspace of the shared Gram sums of the daily panel:
                          // the per-block Grams G[prefix_nblk][slot], filled by every launch before the window kernel
    int prefix_nblk;      // whole TP_PREFIX_BLOCK_ROWS-row blocks of the panel
    int prefix_blk0;      // large-k path: the tables cover blocks prefix_blk0 .. prefix_blk0 + prefix_nblk - 1 (those of the
                          // sub-batch in flight; a whole-panel table at k = 1000 and 125,000 windows would be 102 GB); else 0
    // behind the block Grams: one table of block-window sums Q_L[b0] = G[b0] + .. + G[b0 + L - 1] per whole-block count
    // L that occurs among the batch's windows (at most TP_WINSUM_MAX_L)
    const double* winsum; // Q tables: [n_L][prefix_nblk][slot]
    int winsum_L[4];      // the block counts L (0 = unused entry)
    // large-k path, conjugate, contiguous intraday windows that advance by a fixed stride B (one day of bars): the whole
    // B-row blocks of every window come from block Grams of the intraday panel that the windows of ONE sub-batch share
    // (DESIGN.md section 4b).  Tables of the sub-batch in flight, rebuilt by every launch; null = every row through the MFMAs
    const double* hf_prefix;   // block Grams [hf_nblk][slot] of the raw intraday rows (ones in the border column)
    const double* hf_winsum;   // block-window sums [hf_nblk][slot]: Q[b] = G[b] + .. + G[b + hf_L - 1]
    long long hf_row0;         // intraday panel row where block 0 of these tables starts
    int hf_blk_rows, hf_nblk, hf_L;
    const double* rhs;    // optional [W x k]: replaces the border column before the factorisation
    const double* shift;  // optional [W x 2], Jeffreys only: (d, e) adds d I + e 1 1' to the matrix that is factorised
    double* weights;
    int* status;
    double* aux;
    double* out_rhs;      // optional [W x k]: the right-hand side (border column) each window was solved for
    long long* stamps;    // diagnostic builds only: [w_count x 8] s_memtime stamps per window
    double* dbg_S1;       // optional [k*k + k]: S1 (or J) and the right-hand side of window dbg_w
    long long dbg_w;
    int dbg_mode;         // 1 prior (S0 | c S0 w0), 2 canonical statistics (T | t), 3 posterior (S1 or J | rhs)
    long long w_first, w_count;
    int panel_ld, hf_ld;
    int panel_off32, hf_off32;   // bit 0: explicit-row windows, bit 1: contiguous windows may use 32-bit byte offsets (u24 x u24)
    int k, N, n_r, m, strategy;
    int phase_limit;      // diagnostic (TP_PHASE_LIMIT): 1 = stop after the Gram phases (outputs are then invalid)
    int center_rows;      // Jeffreys: 0 = J = T - t t'/N, 1 = divide by the window's row count instead, 2 = plain T
    double gamma;
    tp_kopts_t opts;      // host-side only: which kernels run this batch
};

struct tp_launch_info_t { int grid, block, lds_bytes, ntile; };

// Shared Gram prefixes of the register-tile path: aligned blocks of the staged chunk's rows (16; 32 with eight waves,
// NT >= 13); 16 rows on the tiled path.
#define TP_WINSUM_MAX_L 4           /* register-tile path: distinct whole-block counts per batch that get a table */
#define TP_WINSUM_RUN 16            /* block positions one thread of tp_window_sums_kernel serves (a group: posterior_fused.hip) */
#define TP_PREFIX_BLOCK_ROWS(nt) ((nt) <= 12 ? 16 : 32)
// doubles per table slot (one Gram of one block or block window: every upper-triangle tile, 4 registers x 64 lanes)
inline size_t tp_fused_slot_doubles(int k) { const int nt = (k + 1 + 15) / 16; return (size_t)(nt * (nt + 1) / 2) * 256; }
inline size_t tp_fused_prefix_bytes(int k, long long panel_rows, int n_L, int* nblk_out) {
    const int nt = (k + 1 + 15) / 16;
    const long long nblk = panel_rows / TP_PREFIX_BLOCK_ROWS(nt);
    if (nblk_out) *nblk_out = (int)nblk;
    return sizeof(double) * (size_t)nblk * (size_t)(1 + n_L) * tp_fused_slot_doubles(k);
}
// Q_L tables from the block Grams (posterior_fused.hip): elementwise, additions only, groups of TP_WINSUM_RUN positions
// abs0: absolute block index of table position 0 (groups are cut in absolute positions: posterior_fused.hip)
hipError_t tp_window_sums_launch(const double* G, double* Q, int nblk, size_t slot_doubles, const int* L, int n_L,
                                 hipStream_t stream, long long abs0 = 0);

// which register-tile kernel runs a tile count: 0 = the multi-wave kernel (posterior_fused_impl.h), 1 = one wave per
// window (posterior_wave_impl.h), 2 = two / four waves per window (posterior_wave2_impl.h); `choice` =
// tp_kopts_t::wave_kernel >= 0 overrides the automatic pick (A/B measurements)
int tp_pick_wave_kernel(int nt, int choice);
// 0 / 1: a plain conjugate / Jeffreys batch (weights, statuses, aux only) - what the one-wave kernel is built for;
// 2: a batch with a matrix read-back, a custom right-hand side, tp_batch_keep_rhs, a shift or a non-default centring
inline int wave_mode(const tp_kargs_t& a) {
    const bool plain = a.dbg_S1 == nullptr && a.rhs == nullptr && a.out_rhs == nullptr && a.shift == nullptr && a.center_rows == 0;
    return plain ? (a.strategy == 0 ? 0 : 1) : 2;
}

// register-tile fused kernel (posterior_fused.hip): k <= tp_fused_max_assets()
int tp_fused_max_assets(void);
hipError_t tp_fused_launch(const tp_kargs_t& a, int grid, hipStream_t stream, tp_launch_info_t* info,
                           int* want_occupancy);

// large-k tiled path (posterior_tiled.hip): k <= tp_tiled_max_assets().  The workspace holds the
// in-flight windows of one batch: arena [G][KP][KP], rinv [G][NSB][64][64], ybar [G][KP], zc [G][m],
// scal [G][8] (s, sqrt s, c, q0, n0), flags [G].
struct tp_tiled_ws_t {
    double* arena;
    double* rinv;
    double* ybar;
    double* zc;
    double* scal;
    int* flags;
    double* part;         // shared intraday sums: partial products S w0 per (window, row block, column block) [G][NS][NS][64]
    int KP, NS, NSB;
};
int tp_tiled_max_assets(void);
void tp_tiled_geometry(int k, int* KP, int* NS, int* NSB);
hipError_t tp_tiled_launch(const tp_kargs_t& a, const tp_tiled_ws_t& ws, hipStream_t stream, bool build_prefix);
hipError_t tp_tiled_prefix_launch(const tp_kargs_t& a, const tp_tiled_ws_t& ws, hipStream_t stream);
// bytes of the shared block Grams + n_L block-window tables of the daily panel in the tiled layout (16-row blocks)
size_t tp_tiled_prefix_bytes(int k, long long panel_rows, int n_L, int* nblk_out);
size_t tp_tiled_slot_doubles(int k);

// price front-end (returns_frontend.hip): out[i][c] = log(prices[num[i]][c] / prices[den[i]][c]), NaN -> 0
hipError_t tp_log_return_rows_launch(const double* prices, int ld, const int* num, const int* den, long long n_out,
                                     double* out, hipStream_t stream);
