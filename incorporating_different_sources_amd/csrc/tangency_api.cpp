// tangency_api.cpp - C-ABI of libtangency.so (include/tangency_posterior.h): contexts, device
// buffers, launches, timing and the RCCL gather.  Compiled with hipcc; no kernels in this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <initializer_list>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/tangency_posterior.h"
#include "posterior_kernels.h"

static_assert(TP_STATUS_NOT_PD == TP_KSTATUS_NOT_PD && TP_STATUS_NONFINITE == TP_KSTATUS_NONFINITE &&
              TP_STATUS_BAD_DENOM == TP_KSTATUS_BAD_DENOM, "status codes out of sync");
static_assert(TP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {
thread_local std::string g_create_error;
}

#define TP_MAX_LANES 4
struct tp_handle_s {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, reg0 = nullptr, reg1 = nullptr;
    hipDeviceProp_t prop;
    std::string err;
    double kernel_ms = 0, h2d_ms = 0, d2h_ms = 0, gather_ms = 0;
    tp_launch_info_t last_launch{0, 0, 0, 0};
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool kernel_timed = false;   // ev0/ev1 bracket the last tp_batch_run and have not been read yet
    // overlapped gather (tp_batch_gather_async): its own high-priority stream next to the kernel stream
    hipStream_t comm_stream = nullptr;
    hipEvent_t cg0 = nullptr, cg1 = nullptr;
    // asynchronous uploads (tp_batch_upload_async): a copy stream of their own, so that the H2D copies of the next
    // batch run under the kernel of the current one
    hipStream_t copy_stream = nullptr;
    hipEvent_t cp0 = nullptr, cp1 = nullptr;
    bool copy_timed = false;
    bool gather_timed = false;
    tp_batch_t deferred = nullptr;  // batch whose tp_batch_gather_async is requested but not yet on the gather stream
    // Tuning switches (A/B measurements, tests): read from the environment ONCE, in tp_create, and changed only through
    // tp_set_option on this handle - no launch path reads the environment (several host threads launch at once in the
    // one-process-all-GPUs mode while a test may be changing it).
    tp_kopts_t opts{};
    int no_shared_gram = 0;         // TP_NO_SHARED_GRAM / "no_shared_gram"
    int hf_share_min_blocks = 6;    // "hf_share_min_blocks": whole intraday blocks per window from which the large-k path shares them
    int tiled_arena_gib = 0;        // TP_TILED_ARENA_GIB / "tiled_arena_gib" (0: default)
    int tiled_arena_mib = 0;        // TP_TILED_ARENA_MIB / "tiled_arena_mib": a sub-GiB arena per lane (depth-first sub-batches)
    int tiled_lanes = 0;            // TP_TILED_LANES / "tiled_lanes": sub-batches in flight on streams of their own (0: default)
    hipStream_t lane_stream[TP_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_done[TP_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_start = nullptr;
    int phase_limit = 0;            // TP_PHASE_LIMIT (diagnostic builds only)
    std::vector<tp_batch_t> batches;   // live batches of this handle (destroyed with it if the caller forgot them)
    // per-step kernel times inside a tp_region_begin / tp_region_end bracket: every timed launch of the region records
    // its own event pair (no host wait in between), tp_region_end reads them all (tp_region_steps returns them)
    std::vector<hipEvent_t> ring0, ring1;
    int ring_used = 0;
    bool in_region = false;
    std::vector<double> step_ms;
};
#define TP_REGION_MAX_STEPS 512

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct tp_batch_s {
    tp_handle_t h = nullptr;
    tp_params_t p{};
    int64_t W = 0;
    int panel_ld = 0, hf_ld = 0;
    DevBuf panel, start, row_idx, n_rows, col_idx, rf_adj, hf_panel, hf_start, hf_row_idx, hf_count, w0, n0;
    DevBuf weights, status, aux, dbg, gather_w, gather_s, weights2, status2, stamps, rhs, out_rhs, shift;
    DevBuf fe_prices, fe_num, fe_den, fe_hf_prices, fe_hf_num, fe_hf_den;   // price front-end staging (freed after a synchronous upload)
    DevBuf prefix;                                            // shared Gram prefixes of the daily panel (register-tile path)
    int prefix_nblk = 0;                                      // > 0: the layout qualifies (decided at upload)
    int winsum_L[4] = {0, 0, 0, 0};                           // register-tile path: the whole-block counts of the windows
    DevBuf t_arena[TP_MAX_LANES], t_rinv[TP_MAX_LANES], t_ybar[TP_MAX_LANES], t_zc[TP_MAX_LANES], t_scal[TP_MAX_LANES],
        t_flags[TP_MAX_LANES];                                // large-k path workspace, one per lane
    // large-k path, conjugate: shared intraday sums (posterior_tiled_wave.h).  Decided at upload (plan_shared_hf): the
    // windows' intraday rows are contiguous, of one length, and advance by hf_B rows; the tables are per sub-batch
    // large-k path: the daily tables cover the blocks of the sub-batch in flight (plan_daily_tables); host copies of
    // what the block ranges are computed from
    bool prefix_per_sub = false;
    std::vector<int64_t> h_start;
    std::vector<int32_t> h_n_rows;
    std::vector<int64_t> h_hf_start;                          // host copy of hf_start (the sub-batches' block ranges)
    int hf_B = 0, hf_L = 0;                                   // rows per block (0 = not shared), whole blocks per window
    long long hf_phase = 0;                                   // blocks start at rows = hf_phase (mod hf_B)
    DevBuf hf_prefix;                                         // block Grams + block-window sums of the sub-batch in flight
    DevBuf t_part[TP_MAX_LANES];                              // pieces of S0 w0 per (window, row block, column block)
    int64_t tiled_capacity = 0;                               // windows in flight per sub-batch (per lane)
    int tiled_lanes = 0;                                      // lanes the workspace was sized for
    bool uploaded = false;
    bool gathered = false;
    bool rhs_valid = false;                          // out_rhs was allocated before the last run (tp_batch_keep_rhs)
    hipEvent_t ran = nullptr;                        // end of this batch's last launch (recorded by every tp_batch_run)
    hipEvent_t upload_done = nullptr;                // tp_batch_upload_async: end of the copies on the copy stream
    bool upload_pending = false;
    // tp_batch_gather_async: results alternate between (weights, status) and (weights2, status2), so that the
    // gather of run i reads one pair while run i+1 writes the other; run i+2 waits for that gather's event
    bool pingpong = false;
    int parity = 0;                                  // pair written by the last run
    hipEvent_t gather_done[2] = {nullptr, nullptr};
    bool gather_pending[2] = {false, false};
    hipEvent_t snap = nullptr;                       // end of the run whose results the requested gather reads
    bool gather_req = false;                         // requested by tp_batch_gather_async, issued by flush_gather
    int gather_req_parity = 0, gather_root = 0;
    double* out_weights() const { return (double*)(parity ? weights2.p : weights.p); }
    int32_t* out_status() const { return (int32_t*)(parity ? status2.p : status.p); }
};

namespace {

// ---- lifetime ---------------------------------------------------------------------------------------------------
// Every live handle is registered here.  A C atexit handler - registered by the first tp_create, i.e. AFTER the HIP and
// RCCL libraries registered their own static teardown, so it runs BEFORE them - destroys what the caller left alive
// (batches first, then the RCCL communicator, streams, events) while the runtimes still work, and then marks the
// library as shut down: a tp_destroy / tp_batch_destroy that arrives later (an object finalised by a host language
// during its own exit) frees nothing on the device and makes no HIP / RCCL call.  Round 2's exit-time aborts
// (std::bad_variant_access, a core dump after a failed test) were exactly such calls and leaked communicators.
std::mutex g_registry_mutex;
std::vector<tp_handle_t> g_live_handles;
std::atomic<bool> g_shut_down{false};
std::atomic<bool> g_exiting{false};        // inside the exit handler: release resources, never start a collective
bool g_atexit_registered = false;

bool runtime_gone(hipError_t e) {
    return e == hipErrorDeinitialized || e == hipErrorNotInitialized || e == hipErrorContextIsDestroyed ||
           e == hipErrorInvalidContext;
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

int destroy_handle(tp_handle_t h, bool device_calls);
int destroy_batch(tp_batch_t b, bool device_calls);

void shutdown_at_exit() {
    g_exiting.store(true);
    std::vector<tp_handle_t> live;
    { std::lock_guard<std::mutex> lk(g_registry_mutex); live.swap(g_live_handles); }
    for (tp_handle_t h : live) (void)destroy_handle(h, true);
    g_shut_down.store(true);
}

int fail(tp_handle_t h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(h, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail((h), TP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define NCCL_TRY(h, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
    return fail((h), TP_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); } while (0)

int ensure(tp_handle_t h, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 8;
    if (b.bytes >= bytes) return TP_OK;
    if (b.p) { HIP_TRY(h, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    HIP_TRY(h, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return TP_OK;
}

int put(tp_handle_t h, DevBuf& b, const void* src, size_t bytes, hipStream_t st = nullptr) {
    if (!src) {   // optional input absent: drop any stale copy
        if (b.p) { HIP_TRY(h, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
        return TP_OK;
    }
    int rc = ensure(h, b, bytes);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st ? st : h->stream));
    return TP_OK;
}

void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

// a device buffer that lives for one call: freed on every return path
struct ScratchBuf : DevBuf {
    ScratchBuf() = default;
    ScratchBuf(const ScratchBuf&) = delete;
    ScratchBuf& operator=(const ScratchBuf&) = delete;
    ~ScratchBuf() { release(*this); }
};

int check_params(tp_handle_t h, const tp_params_t* p, int64_t W) {
    if (!p) return fail(h, TP_ERR_INVALID, "params is NULL");
    if (W < 0) return fail(h, TP_ERR_INVALID, "W=%lld < 0", (long long)W);
    if (p->k < 1) return fail(h, TP_ERR_INVALID, "k=%d < 1", p->k);
    if (p->N < 1 || p->n_r < 1) return fail(h, TP_ERR_INVALID, "N=%d n_r=%d must be >= 1", p->N, p->n_r);
    if (p->strategy != TP_STRATEGY_CONJUGATE && p->strategy != TP_STRATEGY_JEFFREYS)
        return fail(h, TP_ERR_INVALID, "unknown strategy %d", p->strategy);
    if (p->strategy == TP_STRATEGY_CONJUGATE && p->m < 2)
        return fail(h, TP_ERR_INVALID, "conjugate prior needs m >= 2 intraday returns (m=%d)", p->m);
    if (!(p->gamma > 0.0) && !(p->gamma < 0.0)) return fail(h, TP_ERR_INVALID, "gamma must be non-zero");
    if (p->k > tp_max_assets())
        return fail(h, TP_ERR_UNSUPPORTED, "k=%d exceeds the largest supported universe %d", p->k, tp_max_assets());
    return TP_OK;
}

tp_kargs_t make_kargs(tp_batch_t b) {
    tp_kargs_t a;
    memset(&a, 0, sizeof a);
    a.panel = (const double*)b->panel.p;
    a.start = (const long long*)b->start.p;
    a.row_idx = (const int*)b->row_idx.p;
    a.n_rows = (const int*)b->n_rows.p;
    a.col_idx = (const int*)b->col_idx.p;
    a.rf_adj = (const double*)b->rf_adj.p;
    a.hf_panel = (const double*)b->hf_panel.p;
    a.hf_start = (const long long*)b->hf_start.p;
    a.hf_row_idx = (const int*)b->hf_row_idx.p;
    a.hf_count = (const int*)b->hf_count.p;
    a.w0 = (const double*)b->w0.p;
    a.n0 = (const double*)b->n0.p;
    a.prefix = (b->prefix_nblk > 0 && !b->prefix_per_sub) ? (const double*)b->prefix.p : nullptr;
    a.prefix_nblk = b->prefix_per_sub ? 0 : b->prefix_nblk;
    a.prefix_blk0 = 0;
    a.winsum = nullptr;
    for (int i = 0; i < 4; ++i) a.winsum_L[i] = 0;
    if (b->prefix_nblk > 0 && !b->prefix_per_sub) {      // block Grams first, the block-window tables behind them
        const size_t slot = b->p.k <= tp_fused_max_assets() ? tp_fused_slot_doubles(b->p.k) : tp_tiled_slot_doubles(b->p.k);
        a.winsum = (const double*)b->prefix.p + (size_t)b->prefix_nblk * slot;
        for (int i = 0; i < 4; ++i) a.winsum_L[i] = b->winsum_L[i];
    }
    a.rhs = (const double*)b->rhs.p;
    a.shift = (const double*)b->shift.p;
    a.center_rows = (b->p.flags & TP_FLAG_NO_CENTER) ? 2 : (b->p.flags & TP_FLAG_CENTER_BY_ROWS) ? 1 : 0;
    a.phase_limit = 0;
#ifdef TP_STAMP
    a.phase_limit = b->h->phase_limit;                // diagnostic build only (TP_PHASE_LIMIT, read in tp_create)
#endif
    a.opts = b->h->opts;
    a.weights = b->out_weights();
    a.status = (int*)b->out_status();
    a.aux = (double*)b->aux.p;
    a.out_rhs = (double*)b->out_rhs.p;
    a.stamps = (long long*)b->stamps.p;
    a.dbg_S1 = nullptr;
    a.dbg_w = -1;
    a.w_first = 0;
    a.w_count = b->W;
    a.panel_ld = b->panel_ld;
    a.hf_ld = b->hf_ld;
    // 32-bit addressing of the staging loads: row x (8 ld) as a 24 x 24 bit product that fits 32 bits
    auto off32 = [](size_t bytes, int ld, int rows_per_window) {
        if (ld < 1 || (size_t)ld * 8 >= (1u << 24)) return 0;
        const size_t rows = bytes / ((size_t)ld * 8);
        int f = 0;
        if (bytes < (1ull << 32) && rows < (1u << 24)) f |= 1;
        // + 64: the lean loop advances its row offset one chunk past the window before it is clamped
        if (((size_t)rows_per_window + 64) * ld * 8 < (1ull << 32) && (size_t)rows_per_window < (1u << 24)) f |= 2;
        return f;
    };
    a.panel_off32 = off32(b->panel.bytes, b->panel_ld, b->p.n_r);
    a.hf_off32 = b->hf_panel.p ? off32(b->hf_panel.bytes, b->hf_ld, b->p.m) : 0;
    a.k = b->p.k; a.N = b->p.N; a.n_r = b->p.n_r; a.m = b->p.m; a.strategy = b->p.strategy;
    a.gamma = b->p.gamma;
    return a;
}

// Host-side validation of every index the kernel will dereference: a bad offset must never reach
// the device (an out-of-bounds access can take the whole node down).
// price front-end: every (numerator, denominator) row must lie inside the price panel
int validate_pairs(tp_handle_t h, const char* what, const int32_t* num, const int32_t* den, int64_t n, int64_t price_rows) {
    if (!num) return TP_OK;
    if (!den) return fail(h, TP_ERR_INVALID, "%s_num without %s_den", what, what);
    if (n < 1 || n > 0x7fffffffLL) return fail(h, TP_ERR_INVALID, "%s_rows=%lld out of range", what, (long long)n);
    for (int64_t i = 0; i < n; ++i)
        if (num[i] < 0 || num[i] >= price_rows || den[i] < 0 || den[i] >= price_rows)
            return fail(h, TP_ERR_INVALID, "%s pair %lld = (%d, %d) outside the price panel (%lld rows)", what,
                        (long long)i, num[i], den[i], (long long)price_rows);
    return TP_OK;
}

int validate_inputs(tp_handle_t h, const tp_params_t& p, int64_t W, const tp_inputs_t* in_raw) {
    if (!in_raw) return fail(h, TP_ERR_INVALID, "inputs is NULL");
    if (!in_raw->panel || in_raw->panel_rows < 1 || in_raw->panel_ld < 1) return fail(h, TP_ERR_INVALID, "panel missing");
    int rcp = validate_pairs(h, "ret", in_raw->ret_num, in_raw->ret_den, in_raw->ret_rows, in_raw->panel_rows);
    if (rcp != TP_OK) return rcp;
    if (p.strategy == TP_STRATEGY_CONJUGATE && in_raw->hf_panel) {
        rcp = validate_pairs(h, "hf_ret", in_raw->hf_ret_num, in_raw->hf_ret_den, in_raw->hf_ret_rows, in_raw->hf_rows);
        if (rcp != TP_OK) return rcp;
    }
    // with the price front-end the windows address rows of the RETURN panel
    tp_inputs_t eff = *in_raw;
    if (eff.ret_num) eff.panel_rows = eff.ret_rows;
    if (eff.hf_ret_num) eff.hf_rows = eff.hf_ret_rows;
    const tp_inputs_t* in = &eff;
    if (!in->start && !in->row_idx) return fail(h, TP_ERR_INVALID, "need start[] or row_idx[]");
    const bool conj = p.strategy == TP_STRATEGY_CONJUGATE;
    if (conj) {
        if (!in->hf_panel || in->hf_rows < 1 || in->hf_ld < 1) return fail(h, TP_ERR_INVALID, "hf_panel missing");
        if (!in->hf_start && !in->hf_row_idx) return fail(h, TP_ERR_INVALID, "need hf_start[] or hf_row_idx[]");
        if (!in->w0 || !in->n0) return fail(h, TP_ERR_INVALID, "conjugate prior needs w0[] and n0[]");
    }
    const int ncol_need = in->col_idx ? 0 : p.k;
    if (ncol_need > in->panel_ld) return fail(h, TP_ERR_INVALID, "panel_ld=%d < k=%d", in->panel_ld, p.k);
    if (conj && ncol_need > in->hf_ld) return fail(h, TP_ERR_INVALID, "hf_ld=%d < k=%d", in->hf_ld, p.k);
    for (int64_t w = 0; w < W; ++w) {
        const int nr = in->n_rows ? in->n_rows[w] : p.n_r;
        if (nr < 1 || nr > p.n_r) return fail(h, TP_ERR_INVALID, "n_rows[%lld]=%d outside [1,%d]", (long long)w, nr, p.n_r);
        if (in->row_idx) {
            for (int r = 0; r < nr; ++r) {
                const int64_t row = in->row_idx[w * (int64_t)p.n_r + r];
                if (row < 0 || row >= in->panel_rows)
                    return fail(h, TP_ERR_INVALID, "row_idx[%lld][%d]=%lld outside the panel", (long long)w, r, (long long)row);
            }
        } else if (nr > in->panel_rows || in->start[w] < 0 || in->start[w] > in->panel_rows - nr) {   // no start + nr: it may overflow
            return fail(h, TP_ERR_INVALID, "window %lld: %d rows from row %lld lie outside the panel (%lld rows)", (long long)w,
                        nr, (long long)in->start[w], (long long)in->panel_rows);
        }
        if (in->col_idx) {
            for (int j = 0; j < p.k; ++j) {
                const int c = in->col_idx[w * (int64_t)p.k + j];
                if (c < 0 || c >= in->panel_ld || (conj && c >= in->hf_ld))
                    return fail(h, TP_ERR_INVALID, "col_idx[%lld][%d]=%d outside the panel", (long long)w, j, c);
            }
        }
        if (conj) {
            const int mm = in->hf_count ? in->hf_count[w] : p.m;
            if (mm < 2 || mm > p.m) return fail(h, TP_ERR_INVALID, "hf_count[%lld]=%d outside [2,%d]", (long long)w, mm, p.m);
            if (in->hf_row_idx) {
                for (int r = 0; r < mm; ++r) {
                    const int64_t row = in->hf_row_idx[w * (int64_t)p.m + r];
                    if (row < 0 || row >= in->hf_rows)
                        return fail(h, TP_ERR_INVALID, "hf_row_idx[%lld][%d]=%lld outside the panel", (long long)w, r, (long long)row);
                }
            } else if (mm > in->hf_rows || in->hf_start[w] < 0 || in->hf_start[w] > in->hf_rows - mm) {
                return fail(h, TP_ERR_INVALID, "window %lld: %d intraday rows from row %lld lie outside the panel (%lld rows)",
                            (long long)w, mm, (long long)in->hf_start[w], (long long)in->hf_rows);
            }
        }
    }
    return TP_OK;
}

// Workspace of the large-k path.  Default: ONE lane whose arena holds as many in-flight windows as 32 GiB allow (fewer,
// larger launches).  Depth-first alternative (options tiled_lanes / tiled_arena_mib): several small sub-batches in flight,
// each on a stream and a workspace of its own, sized so that all arenas together stay inside the 256 MiB Infinity Cache
// - the left-looking update then re-reads a window's block rows from cache instead of streaming them from HBM.
int ensure_tiled_ws(tp_batch_t b, tp_tiled_ws_t* ws, int* lanes_out) {
    tp_handle_t h = b->h;
    int KP, NS, NSB;
    tp_tiled_geometry(b->p.k, &KP, &NS, &NSB);
    const size_t per_window = sizeof(double) * ((size_t)KP * KP + (size_t)NSB * 64 * 64 + KP + (size_t)b->p.m + 8) + 4;
    int lanes = h->tiled_lanes >= 1 ? (h->tiled_lanes > TP_MAX_LANES ? TP_MAX_LANES : h->tiled_lanes) : 1;
    // in-flight windows of one sub-batch: an arena budget of 32 GiB of the 288 (fewer, larger launches: measured
    // +2-4 % over 6 GiB at k = 500), never more than a third of what is free; tiled_arena_gib / _mib override it
    unsigned long long gib = 32;
    { size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (free_b >> 30) / 3 < gib) gib = (free_b >> 30) / 3 > 1 ? (free_b >> 30) / 3 : 1; }
    if (h->tiled_arena_gib >= 1 && h->tiled_arena_gib <= 200) gib = (unsigned long long)h->tiled_arena_gib;
    unsigned long long arena_bytes = (gib << 30) / (unsigned long long)lanes;
    if (h->tiled_arena_mib >= 1 && h->tiled_arena_mib <= 200 * 1024) arena_bytes = (unsigned long long)h->tiled_arena_mib << 20;
    int64_t G = (int64_t)(arena_bytes / per_window);
    if (G < 1) G = 1;
    if (G > b->W) G = b->W;
    if (G > 65535) G = 65535;
    if ((int64_t)lanes * G > b->W) lanes = (int)((b->W + G - 1) / G);
    if (b->tiled_capacity < G || b->tiled_lanes < lanes) {
        if (b->tiled_capacity > G) G = b->tiled_capacity;
        for (int l = 0; l < lanes; ++l) {
            int rc = ensure(h, b->t_arena[l], sizeof(double) * (size_t)G * KP * KP);
            if (rc == TP_OK) rc = ensure(h, b->t_rinv[l], sizeof(double) * (size_t)G * NSB * 64 * 64);
            if (rc == TP_OK) rc = ensure(h, b->t_ybar[l], sizeof(double) * (size_t)G * KP);
            if (rc == TP_OK) rc = ensure(h, b->t_zc[l], sizeof(double) * (size_t)G * (b->p.m > 0 ? b->p.m : 1));
            if (rc == TP_OK) rc = ensure(h, b->t_scal[l], sizeof(double) * (size_t)G * 8);
            if (rc == TP_OK) rc = ensure(h, b->t_flags[l], sizeof(int) * (size_t)G);
            if (rc != TP_OK) return rc;
        }
        b->tiled_capacity = G;
        b->tiled_lanes = lanes;
    }
    if (b->hf_B > 0)
        for (int l = 0; l < lanes; ++l) {
            int rc = ensure(h, b->t_part[l], sizeof(double) * (size_t)b->tiled_capacity * NS * NS * 64);
            if (rc != TP_OK) return rc;
        }
    for (int l = 0; l < lanes; ++l) {
        ws[l].arena = (double*)b->t_arena[l].p; ws[l].rinv = (double*)b->t_rinv[l].p; ws[l].ybar = (double*)b->t_ybar[l].p;
        ws[l].zc = (double*)b->t_zc[l].p; ws[l].scal = (double*)b->t_scal[l].p; ws[l].flags = (int*)b->t_flags[l].p;
        ws[l].part = (double*)b->t_part[l].p;
        ws[l].KP = KP; ws[l].NS = NS; ws[l].NSB = NSB;
    }
    *lanes_out = lanes;
    return TP_OK;
}

// Shared daily sums of the large-k path for ONE sub-batch (or, `whole`, for the whole panel: several lanes in flight): the
// 16-row blocks its windows cover, block Grams first, one table of block-window sums per whole-block count behind them.
int plan_daily_tables(tp_batch_t b, tp_kargs_t& sub, bool whole) {
    tp_handle_t h = b->h;
    sub.prefix = nullptr; sub.winsum = nullptr; sub.prefix_nblk = 0; sub.prefix_blk0 = 0;
    for (int i = 0; i < 4; ++i) sub.winsum_L[i] = 0;
    long long lo = 0, hi = b->prefix_nblk;
    if (!whole) {
        lo = 0x7fffffffffffffffLL; hi = -1;
        for (int64_t w = sub.w_first; w < sub.w_first + sub.w_count; ++w) {
            const long long f = b->h_start[(size_t)w], cnt = b->h_n_rows.empty() ? b->p.n_r : b->h_n_rows[(size_t)w];
            const long long b0 = (f + 15) / 16, b1 = (f + cnt) / 16;
            if (b1 <= b0) continue;
            if (b0 < lo) lo = b0;
            if (b1 > hi) hi = b1;
        }
        if (hi <= lo) return TP_OK;
        if (hi > b->prefix_nblk) hi = b->prefix_nblk;
        // sharing pays while the windows' rows outnumber the rows of the blocks a few times over
        if ((double)sub.w_count * b->p.n_r < 3.0 * 16.0 * (double)(hi - lo)) return TP_OK;
    }
    const long long nblk = hi - lo;
    int n_L = 0;
    while (n_L < TP_WINSUM_MAX_L && b->winsum_L[n_L] > 0) ++n_L;
    if (nblk < 1 || nblk > 0x3fffffff || n_L == 0) return TP_OK;
    const size_t slot = tp_tiled_slot_doubles(b->p.k);
    const size_t bytes = sizeof(double) * (size_t)nblk * (size_t)(1 + n_L) * slot;
    if (bytes > b->prefix.bytes) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > (free_b + b->prefix.bytes) / 3) return TP_OK;
        int rc = ensure(h, b->prefix, bytes);
        if (rc != TP_OK) return rc;
    }
    sub.prefix = (const double*)b->prefix.p;
    sub.winsum = (const double*)b->prefix.p + (size_t)nblk * slot;
    sub.prefix_nblk = (int)nblk;
    sub.prefix_blk0 = (int)lo;
    for (int i = 0; i < 4; ++i) sub.winsum_L[i] = b->winsum_L[i];
    return TP_OK;
}

// Shared intraday sums of ONE sub-batch (windows sub.w_first .. + sub.w_count): the block range its windows cover, the
// tables sized for it (block Grams, then the block-window sums).  A sub-batch whose windows do not all have hf_L whole blocks
// inside one affordable range keeps the two-pass form (sub.hf_winsum stays null).
int plan_hf_tables(tp_batch_t b, tp_kargs_t& sub) {
    tp_handle_t h = b->h;
    sub.hf_prefix = nullptr; sub.hf_winsum = nullptr;
    const long long B = b->hf_B, ph = b->hf_phase, m = b->p.m;
    long long lo = 0x7fffffffffffffffLL, hi = -1;
    for (int64_t w = sub.w_first; w < sub.w_first + sub.w_count; ++w) {
        const long long f = b->h_hf_start[(size_t)w];
        const long long b0 = (f - ph + B - 1) / B, b1 = (f + m - ph) / B;      // f - ph > -B
        if (b1 - b0 != b->hf_L) return TP_OK;
        if (b0 < lo) lo = b0;
        if (b1 > hi) hi = b1;
    }
    // the table starts on a multiple of the block-window sums' group length: a position's sum is then the same sequence of
    // additions whichever sub-batch asks for it (results do not depend on how a run is cut into sub-batches)
    const long long run = b->hf_L < TP_WINSUM_RUN ? b->hf_L : TP_WINSUM_RUN;
    lo = (lo / run) * run;
    const long long nblk = hi - lo;
    if (nblk < b->hf_L || nblk > 0x3fffffff) return TP_OK;
    // sharing pays while the windows outnumber the blocks they touch a few times over
    if ((double)sub.w_count * (double)b->hf_L < 2.0 * (double)nblk) return TP_OK;
    const size_t slot = tp_tiled_slot_doubles(b->p.k);
    const size_t bytes = sizeof(double) * 2 * (size_t)nblk * slot;
    if (bytes > b->hf_prefix.bytes) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > (free_b + b->hf_prefix.bytes) / 2) return TP_OK;
        int rc = ensure(h, b->hf_prefix, bytes);
        if (rc != TP_OK) return rc;
    }
    sub.hf_prefix = (const double*)b->hf_prefix.p;
    sub.hf_winsum = (const double*)b->hf_prefix.p + (size_t)nblk * slot;
    sub.hf_row0 = ph + B * lo;
    sub.hf_blk_rows = (int)B;
    sub.hf_nblk = (int)nblk;
    sub.hf_L = b->hf_L;
    return TP_OK;
}

int ensure_lane_streams(tp_handle_t h, int lanes) {
    if (!h->lane_start) HIP_TRY(h, hipEventCreateWithFlags(&h->lane_start, hipEventDisableTiming));
    for (int l = 0; l < lanes; ++l) {
        if (!h->lane_stream[l]) HIP_TRY(h, hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking));
        if (!h->lane_done[l]) HIP_TRY(h, hipEventCreateWithFlags(&h->lane_done[l], hipEventDisableTiming));
    }
    return TP_OK;
}

int launch(tp_batch_t b, const tp_kargs_t& a, int64_t count, bool timed) {
    tp_handle_t h = b->h;
    if (count <= 0) return TP_OK;
    if (count > 0x7fffffffLL) return fail(h, TP_ERR_INVALID, "too many windows in one launch");
    // inside a region the launch is bracketed by its own pair of the ring (read by tp_region_end), outside by ev0 / ev1
    hipEvent_t t0 = h->ev0, t1 = h->ev1;
    const bool ring = timed && h->in_region && h->ring_used < (int)h->ring0.size();
    if (ring) { t0 = h->ring0[(size_t)h->ring_used]; t1 = h->ring1[(size_t)h->ring_used]; }
    auto timed_done = [&]() { if (ring) ++h->ring_used; else h->kernel_timed = true; };
    if (a.k <= tp_fused_max_assets()) {
        if (timed) HIP_TRY(h, hipEventRecord(t0, h->stream));
        hipError_t e = tp_fused_launch(a, (int)count, h->stream, &h->last_launch, nullptr);
        if (e != hipSuccess) return fail(h, TP_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
        if (timed) { HIP_TRY(h, hipEventRecord(t1, h->stream)); timed_done(); }
        return TP_OK;
    }
    // large-k path: sub-batches of in-flight windows through the tiled pipeline
    if (a.dbg_S1 != nullptr) return fail(h, TP_ERR_UNSUPPORTED, "matrix read-back is not available on the large-k path");
    tp_tiled_ws_t ws[TP_MAX_LANES];
    int lanes = 1;
    int rc = ensure_tiled_ws(b, ws, &lanes);
    if (rc != TP_OK) return rc;
    if (lanes > 1) { rc = ensure_lane_streams(h, lanes); if (rc != TP_OK) return rc; }
    if (timed) HIP_TRY(h, hipEventRecord(t0, h->stream));
    tp_kargs_t whole = a;
    if (lanes > 1 && b->prefix_per_sub) { rc = plan_daily_tables(b, whole, true); if (rc != TP_OK) return rc; }
    if (lanes > 1) {
        // the shared sums once, on the kernel stream; then every lane's stream starts behind them
        hipError_t e = tp_tiled_prefix_launch(whole, ws[0], h->stream);
        if (e != hipSuccess) return fail(h, TP_ERR_HIP, "tiled pipeline launch failed: %s", hipGetErrorString(e));
        HIP_TRY(h, hipEventRecord(h->lane_start, h->stream));
        for (int l = 0; l < lanes; ++l) HIP_TRY(h, hipStreamWaitEvent(h->lane_stream[l], h->lane_start, 0));
    }
    int64_t sb = 0;
    for (int64_t w0 = 0; w0 < count; w0 += b->tiled_capacity, ++sb) {
        tp_kargs_t sub = lanes > 1 ? whole : a;
        sub.w_first = a.w_first + w0;
        sub.w_count = (count - w0 < b->tiled_capacity) ? (count - w0) : b->tiled_capacity;
        const int l = (int)(sb % lanes);
        if (b->prefix_per_sub && lanes == 1) {
            rc = plan_daily_tables(b, sub, false);
            if (rc != TP_OK) return rc;
        }
        if (b->hf_B > 0 && lanes == 1 && a.strategy == TP_STRATEGY_CONJUGATE) {
            rc = plan_hf_tables(b, sub);
            if (rc != TP_OK) return rc;
        }
        // (per-sub-batch tables: every sub-batch builds its own; a whole-panel table: the first one builds it)
        hipError_t e = tp_tiled_launch(sub, ws[l], lanes > 1 ? h->lane_stream[l] : h->stream,
                                       lanes == 1 && (b->prefix_per_sub || w0 == 0));
        if (e != hipSuccess) return fail(h, TP_ERR_HIP, "tiled pipeline launch failed: %s", hipGetErrorString(e));
    }
    if (lanes > 1) {
        for (int l = 0; l < lanes; ++l) {
            HIP_TRY(h, hipEventRecord(h->lane_done[l], h->lane_stream[l]));
            HIP_TRY(h, hipStreamWaitEvent(h->stream, h->lane_done[l], 0));
        }
    }
    h->last_launch = tp_launch_info_t{(int)(count < b->tiled_capacity ? count : b->tiled_capacity), 256, 36864, ws[0].NS * 4};
    if (timed) { HIP_TRY(h, hipEventRecord(t1, h->stream)); timed_done(); }
    return TP_OK;
}

}  // namespace

static int harvest_kernel_time(tp_handle_t h);
static int flush_gather(tp_handle_t h);

namespace {

// device_calls = false, or a runtime that answers "deinitialised": only the host structures go.
int destroy_batch(tp_batch_t b, bool device_calls) {
    tp_handle_t h = b->h;
    if (device_calls && runtime_gone(hipSetDevice(h->device))) device_calls = false;
    if (device_calls) {
        // A gather that was requested (tp_batch_gather_async) but not yet put on its stream is a collective the peer
        // ranks may already be waiting in: issue it before the buffers go away - dropping it would hang them.
        // (not at process exit: the peers may be gone, and a collective nobody answers would hang the exit)
        if (h->deferred == b && !g_exiting.load()) (void)flush_gather(h);
        (void)hipStreamSynchronize(h->stream);      // (the lanes of the large-k path have joined the kernel stream)
        if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);   // a gather may still read the results
        for (hipEvent_t e : b->gather_done)
            if (e) (void)hipEventDestroy(e);
        if (b->snap) (void)hipEventDestroy(b->snap);
        if (b->upload_done) { (void)hipEventSynchronize(b->upload_done); (void)hipEventDestroy(b->upload_done); }
        if (b->ran) (void)hipEventDestroy(b->ran);
        DevBuf* all[] = {&b->fe_prices, &b->fe_num, &b->fe_den, &b->fe_hf_prices, &b->fe_hf_num, &b->fe_hf_den,
                         &b->panel, &b->start, &b->row_idx, &b->n_rows, &b->col_idx, &b->rf_adj, &b->hf_panel, &b->hf_start,
                         &b->hf_row_idx, &b->hf_count, &b->w0, &b->n0, &b->weights, &b->status, &b->aux, &b->dbg,
                         &b->gather_w, &b->gather_s, &b->weights2, &b->status2, &b->stamps, &b->rhs, &b->out_rhs, &b->shift,
                         &b->prefix, &b->hf_prefix};
        for (DevBuf* d : all) release(*d);
        for (int l = 0; l < TP_MAX_LANES; ++l)
            for (DevBuf* d : {&b->t_arena[l], &b->t_rinv[l], &b->t_ybar[l], &b->t_zc[l], &b->t_scal[l], &b->t_flags[l], &b->t_part[l]}) release(*d);
    }
    if (h->deferred == b) h->deferred = nullptr;
    delete b;
    return TP_OK;
}

int destroy_handle(tp_handle_t h, bool device_calls) {
    if (device_calls && runtime_gone(hipSetDevice(h->device))) device_calls = false;
    // batches the caller left alive go first: they hold device memory, events and possibly a requested gather
    std::vector<tp_batch_t> left;
    left.swap(h->batches);
    for (tp_batch_t b : left) (void)destroy_batch(b, device_calls);
    if (device_calls) {
        if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        if (h->comm) { (void)ncclCommDestroy(h->comm); h->comm = nullptr; }
        for (hipEvent_t e : {h->cg0, h->cg1})
            if (e) (void)hipEventDestroy(e);
        if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
        if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
        for (hipEvent_t e : {h->cp0, h->cp1, h->ev0, h->ev1, h->reg0, h->reg1})
            if (e) (void)hipEventDestroy(e);
        for (int l = 0; l < TP_MAX_LANES; ++l) {
            if (h->lane_stream[l]) { (void)hipStreamSynchronize(h->lane_stream[l]); (void)hipStreamDestroy(h->lane_stream[l]); }
            if (h->lane_done[l]) (void)hipEventDestroy(h->lane_done[l]);
        }
        if (h->lane_start) (void)hipEventDestroy(h->lane_start);
        for (hipEvent_t e : h->ring0) (void)hipEventDestroy(e);
        for (hipEvent_t e : h->ring1) (void)hipEventDestroy(e);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return TP_OK;
}

}  // namespace

extern "C" {

const char* tp_version(void) { return "tangency-posterior 0.6.0 (gfx950, fp64 MFMA: one wavefront per window k<=143, two or four per window k<=239, tiled pipeline k<=2047 with shared daily and intraday block Grams)"; }

int tp_max_assets(void) { return tp_tiled_max_assets(); }

int tp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* tp_last_error(tp_handle_t h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tp_create(int device_id, tp_handle_t* out) {
    if (!out) return fail(nullptr, TP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1)
        return fail(nullptr, TP_ERR_NO_DEVICE, "no HIP device available (%s); there is no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, TP_ERR_INVALID, "device %d out of range (0..%d)", device_id, n - 1);
    tp_handle_t h = new (std::nothrow) tp_handle_s();
    if (!h) return fail(nullptr, TP_ERR_INVALID, "out of host memory");
    h->device = device_id;
#define CREATE_TRY(expr) do { hipError_t e2_ = (expr); if (e2_ != hipSuccess) { \
        int rc_ = fail(nullptr, TP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e2_)); delete h; return rc_; } } while (0)
    CREATE_TRY(hipSetDevice(device_id));
    CREATE_TRY(hipGetDeviceProperties(&h->prop, device_id));
    if (strncmp(h->prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = fail(nullptr, TP_ERR_NO_DEVICE, "device %d is %s; libtangency is built for gfx950 only", device_id,
                      h->prop.gcnArchName);
        delete h;
        return rc;
    }
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));
    CREATE_TRY(hipEventCreate(&h->reg0));
    CREATE_TRY(hipEventCreate(&h->reg1));
#undef CREATE_TRY
    // the environment is read here and nowhere else (see tp_handle_s::opts)
    h->opts.wave_kernel = env_int("TP_WAVE_KERNEL", -1);
    h->opts.tiled_wave = env_int("TP_TILED_WAVE", -1);
    h->opts.tiled_fuse = env_int("TP_TILED_FUSE", -1);
    h->no_shared_gram = getenv("TP_NO_SHARED_GRAM") != nullptr ? 1 : 0;
    h->tiled_arena_gib = env_int("TP_TILED_ARENA_GIB", 0);
    h->tiled_arena_mib = env_int("TP_TILED_ARENA_MIB", 0);
    h->tiled_lanes = env_int("TP_TILED_LANES", 0);
    h->phase_limit = env_int("TP_PHASE_LIMIT", 0);
    {
        std::lock_guard<std::mutex> lk(g_registry_mutex);
        if (!g_atexit_registered) { atexit(shutdown_at_exit); g_atexit_registered = true; }
        g_live_handles.push_back(h);
    }
    *out = h;
    return TP_OK;
}

int tp_set_option(tp_handle_t h, const char* name, int value) {
    if (!h || !name) return TP_ERR_INVALID;
    const std::string n(name);
    if (n == "wave_kernel") h->opts.wave_kernel = value;
    else if (n == "tiled_wave") h->opts.tiled_wave = value;
    else if (n == "tiled_fuse") h->opts.tiled_fuse = value;
    else if (n == "no_shared_gram") h->no_shared_gram = value != 0;
    else if (n == "tiled_arena_gib") h->tiled_arena_gib = value;
    else if (n == "tiled_arena_mib") h->tiled_arena_mib = value;
    else if (n == "tiled_lanes") h->tiled_lanes = value;
    else if (n == "hf_share_min_blocks") h->hf_share_min_blocks = value;
    else return fail(h, TP_ERR_INVALID, "tp_set_option: unknown option '%s'", name);
    return TP_OK;
}

int tp_destroy(tp_handle_t h) {
    if (!h) return TP_OK;
    if (g_shut_down.load()) return TP_OK;           // the exit handler already took every live handle down (h is gone)
    {
        std::lock_guard<std::mutex> lk(g_registry_mutex);
        auto it = std::find(g_live_handles.begin(), g_live_handles.end(), h);
        if (it == g_live_handles.end()) return TP_OK;   // not (or no longer) a live handle: never touch it
        g_live_handles.erase(it);
    }
    return destroy_handle(h, true);
}

int tp_device_info(tp_handle_t h, char* name, int name_len, int* compute_units, int* clock_mhz, int64_t* hbm_bytes) {
    if (!h) return TP_ERR_INVALID;
    if (name && name_len > 0) { snprintf(name, (size_t)name_len, "%s (%s)", h->prop.name[0] ? h->prop.name : "AMD Instinct MI355X", h->prop.gcnArchName); }
    if (compute_units) *compute_units = h->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = h->prop.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = (int64_t)h->prop.totalGlobalMem;
    return TP_OK;
}

int tp_log_returns(tp_handle_t h, const double* prices, int64_t price_rows, int32_t ld, const int32_t* num,
                   const int32_t* den, int64_t n_out, double* out) {
    if (!h) return TP_ERR_INVALID;
    if (!prices || !out || price_rows < 1 || ld < 1) return fail(h, TP_ERR_INVALID, "tp_log_returns: prices / out missing");
    if (!num) return fail(h, TP_ERR_INVALID, "tp_log_returns: num missing");
    int rc = validate_pairs(h, "ret", num, den, n_out, price_rows);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    ScratchBuf dp, dn, dd, dout;
    rc = put(h, dp, prices, sizeof(double) * (size_t)price_rows * ld);
    if (rc == TP_OK) rc = put(h, dn, num, sizeof(int32_t) * (size_t)n_out);
    if (rc == TP_OK) rc = put(h, dd, den, sizeof(int32_t) * (size_t)n_out);
    if (rc == TP_OK) rc = ensure(h, dout, sizeof(double) * (size_t)n_out * ld);
    if (rc == TP_OK) {
        HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
        hipError_t e = tp_log_return_rows_launch((const double*)dp.p, ld, (const int*)dn.p, (const int*)dd.p,
                                                 (long long)n_out, (double*)dout.p, h->stream);
        if (e != hipSuccess) rc = fail(h, TP_ERR_HIP, "log-return kernel launch failed: %s", hipGetErrorString(e));
    }
    if (rc == TP_OK) {
        HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
        h->kernel_timed = true;                          // tp_last_timing().kernel_ms = this kernel
        HIP_TRY(h, hipMemcpyAsync(out, dout.p, sizeof(double) * (size_t)n_out * ld, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        rc = harvest_kernel_time(h);
    }
    return rc;
}

int tp_batch_create(tp_handle_t h, const tp_params_t* p, int64_t W, tp_batch_t* out) {
    if (!h || !out) return TP_ERR_INVALID;
    *out = nullptr;
    int rc = check_params(h, p, W);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    tp_batch_t b = new (std::nothrow) tp_batch_s();
    if (!b) return fail(h, TP_ERR_INVALID, "out of host memory");
    b->h = h; b->p = *p; b->W = W;
    rc = ensure(h, b->weights, sizeof(double) * (size_t)W * p->k);
    if (rc == TP_OK) rc = ensure(h, b->status, sizeof(int32_t) * (size_t)W);
    if (rc == TP_OK) rc = ensure(h, b->aux, sizeof(double) * (size_t)W * TP_AUX_STRIDE);
    if (rc != TP_OK) { destroy_batch(b, true); return rc; }
    h->batches.push_back(b);
    *out = b;
    return TP_OK;
}

int tp_batch_destroy(tp_batch_t b) {
    if (!b) return TP_OK;
    if (g_shut_down.load()) return TP_OK;           // destroyed with its handle by the exit handler
    tp_handle_t h = b->h;
    auto it = std::find(h->batches.begin(), h->batches.end(), b);
    if (it != h->batches.end()) h->batches.erase(it);
    return destroy_batch(b, true);
}

// Rolling windows over one shared panel overlap almost entirely; the register-tile path then takes the whole aligned
// row blocks of every window from running Gram sums of the panel that all windows share (posterior_fused_impl.h,
// block_gram_kernel + tp_window_sums_kernel; the tiled path: tiled_prefix_kernel) instead of pushing every row of every window through the MFMAs.  Qualifies: contiguous windows
// (start[]), no column gather, no per-row risk-free adjustment, k in the register-tile range, and windows that
// together cover the panel at least three times.  The sums are recomputed by EVERY tp_batch_run (nothing is kept
// between runs); TP_FLAG_NO_SHARED_GRAM switches the scheme off.
static int plan_shared_gram(tp_batch_t b, const tp_inputs_t* in) {
    tp_handle_t h = b->h;
    b->prefix_nblk = 0;
    b->prefix_per_sub = false;
    b->h_start.clear(); b->h_n_rows.clear();
    const tp_params_t& p = b->p;
    if ((p.flags & TP_FLAG_NO_SHARED_GRAM) || h->no_shared_gram) return TP_OK;
    if (in->row_idx || in->col_idx || in->rf_adj || !in->start) return TP_OK;
    const long long rows = in->ret_num ? in->ret_rows : in->panel_rows;
    int nblk = 0;
    size_t bytes = 0;
    for (int i = 0; i < 4; ++i) b->winsum_L[i] = 0;
    {
        // one table of block-window sums per whole-block count that occurs among the windows (rolling windows of one
        // length have two: 249 rows over 16-row blocks cover 14 or 15 whole blocks)
        const bool fused = p.k <= tp_fused_max_assets();
        const int blk = fused ? TP_PREFIX_BLOCK_ROWS((p.k + 1 + 15) / 16) : 16;
        int n_L = 0;
        for (int64_t w = 0; w < b->W; ++w) {
            const long long first = in->start[w], cnt = in->n_rows ? in->n_rows[w] : p.n_r;
            const long long L = (first + cnt) / blk - (first + blk - 1) / blk;
            if (L < 1) continue;
            int i = 0;
            while (i < n_L && b->winsum_L[i] != (int)L) ++i;
            if (i == n_L) {
                if (n_L == TP_WINSUM_MAX_L) { for (int q = 0; q < 4; ++q) b->winsum_L[q] = 0; return TP_OK; }   // irregular windows: no sharing
                b->winsum_L[n_L++] = (int)L;
            }
        }
        if (n_L == 0) return TP_OK;
        bytes = fused ? tp_fused_prefix_bytes(p.k, rows, n_L, &nblk) : tp_tiled_prefix_bytes(p.k, rows, n_L, &nblk);
    }
    if (nblk < 2 || (double)b->W * p.n_r < 3.0 * (double)rows) return TP_OK;
    // a handful of tiny windows: the two extra launches cost more than the rows they save (configs[0], k = 10, 100 windows:
    // 28.6 us with the shared sums, 21.5 us without)
    if (p.k <= 31 && b->W < 256) return TP_OK;
    if (p.k > tp_fused_max_assets()) {
        // large-k path: a slot is megabytes (4.35 MB at k = 1000), a table over the whole panel of a long run does not fit
        // (102 GB at 125,000 windows) - the tables are built per sub-batch, for the blocks its windows cover
        if ((size_t)in->panel_ld * 8 * 4096 >= (1ull << 32)) return TP_OK;
        b->prefix_per_sub = true;
        b->prefix_nblk = nblk;
        b->h_start.assign(in->start, in->start + b->W);
        if (in->n_rows) b->h_n_rows.assign(in->n_rows, in->n_rows + b->W);
        return TP_OK;
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 3) return TP_OK;   // never crowd out the batch itself
    if ((size_t)in->panel_ld * 8 * 4096 >= (1ull << 32)) return TP_OK;      // 32-bit offsets inside a segment
    int rc = ensure(h, b->prefix, bytes);
    if (rc != TP_OK) return rc;
    b->prefix_nblk = nblk;
    return TP_OK;
}

// Large-k path, conjugate: do the intraday windows qualify for shared sums (posterior_tiled_wave.h)?  Contiguous windows of
// ONE length over ungathered columns, all starts a multiple of a stride B apart (rolling dates: one day of bars),
// a contiguous daily layout without a risk-free adjustment (the kernel variant is built for that), and at least two whole
// B-row blocks per window.  Blocks are aligned to the windows' ENDS: a window that starts behind a day's first bar
// (its return is undefined, ref:311-312) ends on a day boundary.
static void plan_shared_hf(tp_batch_t b, const tp_inputs_t* in) {
    tp_handle_t h = b->h;
    const tp_params_t& p = b->p;
    b->hf_B = 0; b->hf_L = 0; b->hf_phase = 0;
    b->h_hf_start.clear();
    if (p.strategy != TP_STRATEGY_CONJUGATE || p.k <= tp_fused_max_assets()) return;
    if ((p.flags & TP_FLAG_NO_SHARED_GRAM) || h->no_shared_gram) return;
    if (!in->hf_start || in->hf_row_idx || in->hf_count || in->col_idx || in->row_idx || in->rf_adj || !in->start) return;
    if (b->W < 4) return;
    // B: the largest stride all starts are multiples of (apart from a common offset) - the order of the windows in the
    // batch does not matter, a reversed or shuffled batch takes the same decision and the same tables
    long long B = 0;
    for (int64_t w = 1; w < b->W; ++w) {
        long long d = in->hf_start[w] - in->hf_start[0];
        if (d < 0) d = -d;
        while (d != 0) { const long long t = B % d; B = d; d = t; }       // B = gcd(B, d)
    }
    if (B < 16 || B > 4096) return;
    const long long m = p.m;
    const long long ph = (in->hf_start[0] + m) % B;      // (start - ph > -B: the ceilings below stay exact)
    const long long f = in->hf_start[0];
    const long long L = (f + m - ph) / B - (f - ph + B - 1) / B;
    // Measured at k = 500 (8,192 windows): with L = 4 whole days per window the tables cost what they save (block Grams 0.52 us +
    // block-window sums 0.43 us + a second slot read per window against 312 rows at 4.6 ns); at k = 1,000 with L = 21:
    // 21.1 k -> 49.3 k windows/s.  Break-even L = 3.5; shared from 6 (option hf_share_min_blocks, tests use 2).
    if (L < 2 || L < h->hf_share_min_blocks) return;
    b->hf_B = (int)B; b->hf_L = (int)L; b->hf_phase = ph;
    b->h_hf_start.assign(in->hf_start, in->hf_start + b->W);
}

// H2D of one batch on stream `st`.  wait = true: the synchronous form (tp_batch_upload) on the kernel stream;
// wait = false: copies are only queued (pinned host memory makes them truly asynchronous), upload_done marks their end.
static int upload_common(tp_batch_t b, const tp_inputs_t* in, hipStream_t st, bool wait) {
    tp_handle_t h = b->h;
    const tp_params_t& p = b->p;
    const int64_t W = b->W;
    int rc = validate_inputs(h, p, W, in);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const bool conj = p.strategy == TP_STRATEGY_CONJUGATE;
    // a launch of THIS batch that is still running reads the buffers about to be overwritten: the copy stream waits
    // for it (other batches' launches on the kernel stream are what the copies are meant to run under)
    if (!wait && b->ran) HIP_TRY(h, hipStreamWaitEvent(st, b->ran, 0));
    // a synchronous upload while an asynchronous one is still copying into the same buffers: let that one finish first
    if (wait && b->upload_pending && b->upload_done) HIP_TRY(h, hipEventSynchronize(b->upload_done));
    hipEvent_t e0 = wait ? h->ev0 : h->cp0, e1 = wait ? h->ev1 : h->cp1;
    if (!wait && h->copy_timed) {     // read the previous asynchronous upload's span before its events are reused
        HIP_TRY(h, hipEventSynchronize(h->cp1));
        float ms0 = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms0, h->cp0, h->cp1));
        h->h2d_ms = ms0;
        h->copy_timed = false;
    }
    HIP_TRY(h, hipEventRecord(e0, st));
    if (wait) h->kernel_timed = false;
#define PUT(buf, ptr, bytes) do { rc = put(h, b->buf, (ptr), (bytes), st); if (rc != TP_OK) return rc; } while (0)
    // panels: log-returns as given, or formed on the device from prices (returns_frontend.hip)
    auto panel_in = [&](DevBuf& dst, DevBuf& prices, DevBuf& pnum, DevBuf& pden, const double* src, int64_t rows, int ld,
                        const int32_t* num, const int32_t* den, int64_t n_out) -> int {
        if (!num) return put(h, dst, src, sizeof(double) * (size_t)rows * ld, st);
        int r = put(h, prices, src, sizeof(double) * (size_t)rows * ld, st);
        if (r == TP_OK) r = put(h, pnum, num, sizeof(int32_t) * (size_t)n_out, st);
        if (r == TP_OK) r = put(h, pden, den, sizeof(int32_t) * (size_t)n_out, st);
        if (r == TP_OK) r = ensure(h, dst, sizeof(double) * (size_t)n_out * ld);
        if (r != TP_OK) return r;
        hipError_t e = tp_log_return_rows_launch((const double*)prices.p, ld, (const int*)pnum.p, (const int*)pden.p,
                                                 (long long)n_out, (double*)dst.p, st);
        if (e != hipSuccess) return fail(h, TP_ERR_HIP, "log-return kernel launch failed: %s", hipGetErrorString(e));
        return TP_OK;
    };
    rc = panel_in(b->panel, b->fe_prices, b->fe_num, b->fe_den, in->panel, in->panel_rows, in->panel_ld, in->ret_num,
                  in->ret_den, in->ret_rows);
    if (rc != TP_OK) return rc;
    PUT(start, in->start, sizeof(int64_t) * (size_t)W);
    PUT(row_idx, in->row_idx, sizeof(int32_t) * (size_t)W * p.n_r);
    PUT(n_rows, in->n_rows, sizeof(int32_t) * (size_t)W);
    PUT(col_idx, in->col_idx, sizeof(int32_t) * (size_t)W * p.k);
    PUT(rf_adj, in->rf_adj, sizeof(double) * (size_t)W * p.n_r);
    if (conj) {
        rc = panel_in(b->hf_panel, b->fe_hf_prices, b->fe_hf_num, b->fe_hf_den, in->hf_panel, in->hf_rows, in->hf_ld,
                      in->hf_ret_num, in->hf_ret_den, in->hf_ret_rows);
        if (rc != TP_OK) return rc;
        PUT(hf_start, in->hf_start, sizeof(int64_t) * (size_t)W);
        PUT(hf_row_idx, in->hf_row_idx, sizeof(int32_t) * (size_t)W * p.m);
        PUT(hf_count, in->hf_count, sizeof(int32_t) * (size_t)W);
        PUT(w0, in->w0, sizeof(double) * (size_t)W * p.k);
        PUT(n0, in->n0, sizeof(double) * (size_t)W);
    }
#undef PUT
    b->panel_ld = in->panel_ld;
    b->hf_ld = conj ? in->hf_ld : 0;
    rc = plan_shared_gram(b, in);
    if (rc != TP_OK) return rc;
    plan_shared_hf(b, in);
    HIP_TRY(h, hipEventRecord(e1, st));
    if (wait) {
        HIP_TRY(h, hipEventSynchronize(e1));
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
        h->h2d_ms = ms;
        // the price staging is needed only until the return panels exist
        for (DevBuf* d : {&b->fe_prices, &b->fe_num, &b->fe_den, &b->fe_hf_prices, &b->fe_hf_num, &b->fe_hf_den}) release(*d);
        b->upload_pending = false;
    } else {
        if (!b->upload_done) HIP_TRY(h, hipEventCreateWithFlags(&b->upload_done, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(b->upload_done, st));
        b->upload_pending = true;
        h->copy_timed = true;
    }
    b->uploaded = true;
    return TP_OK;
}

int tp_batch_upload(tp_batch_t b, const tp_inputs_t* in) {
    if (!b) return TP_ERR_INVALID;
    return upload_common(b, in, b->h->stream, true);
}

int tp_batch_upload_async(tp_batch_t b, const tp_inputs_t* in) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->copy_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreate(&h->cp0));
        HIP_TRY(h, hipEventCreate(&h->cp1));
    }
    return upload_common(b, in, h->copy_stream, false);
}

int tp_batch_upload_wait(tp_batch_t b) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    if (!b->upload_done) return TP_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventSynchronize(b->upload_done));
    if (h->copy_timed && hipEventQuery(h->cp1) == hipSuccess) {
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->cp0, h->cp1));
        h->h2d_ms = ms;
        h->copy_timed = false;
    }
    return TP_OK;
}

// Page-locked host memory for panels and results: hipMemcpyAsync from / to it is a real DMA at PCIe rate and does
// not block the calling thread (pageable buffers are staged through the runtime's own bounce buffers at about half
// the rate).
int tp_host_alloc(void** out, int64_t bytes) {
    if (!out || bytes < 0) return TP_ERR_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault) != hipSuccess) { *out = nullptr; return TP_ERR_HIP; }
    return TP_OK;
}

int tp_host_free(void* p) {
    if (!p) return TP_OK;
    return hipHostFree(p) == hipSuccess ? TP_OK : TP_ERR_HIP;
}

int tp_batch_shared_gram_blocks(tp_batch_t b) { return b ? b->prefix_nblk : 0; }
int tp_batch_shared_intraday_blocks(tp_batch_t b) { return (b && b->hf_B > 0) ? b->hf_L : 0; }

int tp_batch_set_rhs(tp_batch_t b, const double* rhs) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));       // a running launch may still read the old one
    int rc = put(h, b->rhs, rhs, sizeof(double) * (size_t)b->W * b->p.k);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TP_OK;
}

int tp_batch_set_shift(tp_batch_t b, const double* shift) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    if (shift && b->p.strategy != TP_STRATEGY_JEFFREYS)
        return fail(h, TP_ERR_INVALID, "tp_batch_set_shift applies to the Jeffreys strategy only");
    if (shift)
        for (int64_t i = 0; i < 2 * b->W; ++i)
            if (!(shift[i] >= 0.0) || !std::isfinite(shift[i]))
                return fail(h, TP_ERR_INVALID, "tp_batch_set_shift: shift[%lld] must be finite and >= 0", (long long)i);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));       // a running launch may still read the old one
    int rc = put(h, b->shift, shift, sizeof(double) * 2 * (size_t)b->W);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TP_OK;
}

int tp_batch_download_rhs(tp_batch_t b, double* rhs_out) {
    if (!b || !rhs_out) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    if (!b->uploaded) return fail(h, TP_ERR_INVALID, "tp_batch_download_rhs before tp_batch_upload");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t bytes = sizeof(double) * (size_t)b->W * b->p.k;
    if (!b->out_rhs.p || !b->rhs_valid)
        return fail(h, TP_ERR_INVALID, "tp_batch_download_rhs: call tp_batch_keep_rhs before the tp_batch_run whose "
                                       "right-hand sides are wanted");
    HIP_TRY(h, hipMemcpyAsync(rhs_out, b->out_rhs.p, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return harvest_kernel_time(h);
}

int tp_batch_keep_rhs(tp_batch_t b, int on) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));       // a running launch may still write the old buffer
    b->rhs_valid = false;
    if (!on) { release(b->out_rhs); return TP_OK; }
    return ensure(h, b->out_rhs, sizeof(double) * (size_t)b->W * b->p.k);
}

int tp_batch_run(tp_batch_t b) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    if (!b->uploaded) return fail(h, TP_ERR_INVALID, "tp_batch_run before tp_batch_upload");
    HIP_TRY(h, hipSetDevice(h->device));
    if (b->upload_pending) {          // tp_batch_upload_async: the kernel stream waits for the copy stream's event
        HIP_TRY(h, hipStreamWaitEvent(h->stream, b->upload_done, 0));
        b->upload_pending = false;
    }
    if (b->out_rhs.p) b->rhs_valid = true;
    if (b->pingpong) {
        b->parity ^= 1;
        if (b->gather_pending[b->parity]) {
            // The gather issued one run ago read this pair.  The HOST waits for it (the kernel of the previous run
            // is still executing, so the GPU does not idle): no stream of this library ever waits for another
            // stream's event on the device - such cross-queue waits cost ~0.1 ms per step (measured).
            HIP_TRY(h, hipEventSynchronize(b->gather_done[b->parity]));
            b->gather_pending[b->parity] = false;
        }
    }
    tp_kargs_t a = make_kargs(b);
    int rc = launch(b, a, b->W, true);
    if (rc != TP_OK) return rc;
    // the end of this launch, recorded on EVERY run: a later tp_batch_upload_async of this batch - also the first one,
    // after synchronous uploads - makes the copy stream wait for it before it overwrites what the launch reads
    if (!b->ran) HIP_TRY(h, hipEventCreateWithFlags(&b->ran, hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(b->ran, h->stream));
    return flush_gather(h);      // with the next kernel queued, put the requested gather of the previous run on its stream
}

static int harvest_kernel_time(tp_handle_t h) {
    if (h->kernel_timed) {
        HIP_TRY(h, hipEventSynchronize(h->ev1));
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        h->kernel_ms = ms;
        h->kernel_timed = false;
    }
    return TP_OK;
}

static int harvest_gather_time(tp_handle_t h) {
    if (h->gather_timed) {
        HIP_TRY(h, hipEventSynchronize(h->cg1));
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->cg0, h->cg1));
        h->gather_ms = ms;
        h->gather_timed = false;
    }
    return TP_OK;
}

int tp_synchronize(tp_handle_t h) {
    if (!h) return TP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    int rcf = flush_gather(h);
    if (rcf != TP_OK) return rcf;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) {
        HIP_TRY(h, hipStreamSynchronize(h->comm_stream));
        int rc = harvest_gather_time(h);
        if (rc != TP_OK) return rc;
    }
    return harvest_kernel_time(h);
}

int tp_batch_download(tp_batch_t b, double* weights, int32_t* status, double* aux) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int rc = harvest_kernel_time(h);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    if (weights) HIP_TRY(h, hipMemcpyAsync(weights, b->out_weights(), sizeof(double) * (size_t)b->W * b->p.k, hipMemcpyDeviceToHost, h->stream));
    if (status) HIP_TRY(h, hipMemcpyAsync(status, b->out_status(), sizeof(int32_t) * (size_t)b->W, hipMemcpyDeviceToHost, h->stream));
    if (aux) HIP_TRY(h, hipMemcpyAsync(aux, b->aux.p, sizeof(double) * (size_t)b->W * TP_AUX_STRIDE, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->d2h_ms = ms;
    return TP_OK;
}

int tp_batch_download_matrix(tp_batch_t b, int64_t w, int what, double* M, double* rhs) {
    if (!b || !M) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    if (!b->uploaded) return fail(h, TP_ERR_INVALID, "tp_batch_download_matrix before tp_batch_upload");
    if (w < 0 || w >= b->W) return fail(h, TP_ERR_INVALID, "window %lld out of range", (long long)w);
    if (what < TP_MATRIX_PRIOR || what > TP_MATRIX_POSTERIOR) return fail(h, TP_ERR_INVALID, "unknown matrix id %d", what);
    if (what == TP_MATRIX_PRIOR && b->p.strategy != TP_STRATEGY_CONJUGATE)
        return fail(h, TP_ERR_INVALID, "the prior scatter exists for the conjugate strategy only");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t kk = (size_t)b->p.k * b->p.k;
    int rc = ensure(h, b->dbg, sizeof(double) * (kk + b->p.k));
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    rc = harvest_kernel_time(h);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipMemsetAsync(b->dbg.p, 0, sizeof(double) * (kk + b->p.k), h->stream));
    tp_kargs_t a = make_kargs(b);
    a.dbg_S1 = (double*)b->dbg.p;
    a.dbg_w = w;
    a.dbg_mode = what;
    a.w_first = w;
    a.w_count = 1;
    rc = launch(b, a, 1, false);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(M, b->dbg.p, sizeof(double) * kk, hipMemcpyDeviceToHost, h->stream));
    if (rhs) HIP_TRY(h, hipMemcpyAsync(rhs, (double*)b->dbg.p + kk, sizeof(double) * b->p.k, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TP_OK;
}

int tp_batch_download_S1(tp_batch_t b, int64_t w, double* S1) {
    return tp_batch_download_matrix(b, w, TP_MATRIX_POSTERIOR, S1, nullptr);
}

int tp_batch_debug_stamps(tp_batch_t b, int64_t* stamps) {
    if (!b || !stamps) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
#ifdef TP_STAMP
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure(h, b->stamps, sizeof(int64_t) * (size_t)b->W * 40);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipMemsetAsync(b->stamps.p, 0, sizeof(int64_t) * (size_t)b->W * 40, h->stream));
    rc = tp_batch_run(b);
    if (rc != TP_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(stamps, b->stamps.p, sizeof(int64_t) * (size_t)b->W * 40, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return harvest_kernel_time(h);
#else
    return fail(h, TP_ERR_UNSUPPORTED, "libtangency was built without TP_STAMP (diagnostic phase stamps)");
#endif
}

int tp_posterior_batch(tp_handle_t h, const tp_params_t* p, int64_t W, const tp_inputs_t* in, double* weights,
                       int32_t* status, double* aux) {
    tp_batch_t b = nullptr;
    int rc = tp_batch_create(h, p, W, &b);
    if (rc != TP_OK) return rc;
    rc = tp_batch_upload(b, in);
    if (rc == TP_OK) rc = tp_batch_run(b);
    if (rc == TP_OK) rc = tp_batch_download(b, weights, status, aux);
    tp_batch_destroy(b);
    return rc;
}

int tp_last_timing(tp_handle_t h, double* kernel_ms, double* h2d_ms, double* d2h_ms, double* gather_ms) {
    if (!h) return TP_ERR_INVALID;
    if (kernel_ms) *kernel_ms = h->kernel_ms;
    if (h2d_ms) *h2d_ms = h->h2d_ms;
    if (d2h_ms) *d2h_ms = h->d2h_ms;
    if (gather_ms) *gather_ms = h->gather_ms;
    return TP_OK;
}

int tp_region_begin(tp_handle_t h) {
    if (!h) return TP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->ring0.empty()) {
        for (int i = 0; i < TP_REGION_MAX_STEPS; ++i) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_TRY(h, hipEventCreate(&a));
            h->ring0.push_back(a);
            HIP_TRY(h, hipEventCreate(&b));
            h->ring1.push_back(b);
        }
    }
    h->ring_used = 0;
    h->step_ms.clear();
    h->in_region = true;
    HIP_TRY(h, hipEventRecord(h->reg0, h->stream));
    return TP_OK;
}

int tp_region_end(tp_handle_t h, double* ms) {
    if (!h) return TP_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    h->in_region = false;
    HIP_TRY(h, hipEventRecord(h->reg1, h->stream));
    HIP_TRY(h, hipEventSynchronize(h->reg1));
    float f = 0;
    HIP_TRY(h, hipEventElapsedTime(&f, h->reg0, h->reg1));
    if (ms) *ms = f;
    for (int i = 0; i < h->ring_used; ++i) {
        float s = 0;
        HIP_TRY(h, hipEventElapsedTime(&s, h->ring0[(size_t)i], h->ring1[(size_t)i]));
        h->step_ms.push_back((double)s);
    }
    if (!h->step_ms.empty()) h->kernel_ms = h->step_ms.back();
    return harvest_kernel_time(h);
}

int tp_region_steps(tp_handle_t h, double* step_ms, int capacity, int* n_steps) {
    if (!h || !n_steps || capacity < 0) return TP_ERR_INVALID;
    const int n = (int)h->step_ms.size();
    *n_steps = n;
    if (step_ms)
        for (int i = 0; i < n && i < capacity; ++i) step_ms[i] = h->step_ms[(size_t)i];
    return TP_OK;
}

int tp_last_launch(tp_handle_t h, int* grid, int* block, int* lds_bytes, int* ntile) {
    if (!h) return TP_ERR_INVALID;
    if (grid) *grid = h->last_launch.grid;
    if (block) *block = h->last_launch.block;
    if (lds_bytes) *lds_bytes = h->last_launch.lds_bytes;
    if (ntile) *ntile = h->last_launch.ntile;
    return TP_OK;
}

int tp_comm_unique_id(void* id) {
    if (!id) return TP_ERR_INVALID;
    ncclUniqueId uid;
    if (ncclGetUniqueId(&uid) != ncclSuccess) return TP_ERR_RCCL;
    memcpy(id, &uid, sizeof uid);
    return TP_OK;
}

int tp_comm_init(tp_handle_t h, const void* id, int rank, int world) {
    if (!h || !id) return TP_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(h, TP_ERR_INVALID, "bad rank %d / world %d", rank, world);
    if (h->comm) return fail(h, TP_ERR_INVALID, "communicator already initialised");
    HIP_TRY(h, hipSetDevice(h->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    NCCL_TRY(h, ncclCommInitRank(&h->comm, world, uid, rank));
    h->rank = rank;
    h->world = world;
    return TP_OK;
}

int tp_comm_count(tp_handle_t h, int* ranks) {
    if (!h || !ranks) return TP_ERR_INVALID;
    *ranks = 0;
    if (!h->comm) return fail(h, TP_ERR_INVALID, "tp_comm_count without a communicator");
    NCCL_TRY(h, ncclCommCount(h->comm, ranks));
    return TP_OK;
}

// Single-process form: one communicator over the n handles of this process (rank i = handles[i]), no id exchange
// and no launcher - what main.py (one process, src/main.py:26) can use on an 8-GPU node.
int tp_comm_init_all(tp_handle_t* handles, int n) {
    if (!handles || n < 1) return TP_ERR_INVALID;
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return TP_ERR_INVALID;
        if (handles[i]->comm) return fail(handles[i], TP_ERR_INVALID, "communicator already initialised");
        devs[(size_t)i] = handles[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[(size_t)j] == devs[(size_t)i])
                return fail(handles[0], TP_ERR_INVALID, "tp_comm_init_all: handles %d and %d share device %d (one rank per GPU)", j, i, devs[(size_t)i]);
    }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    NCCL_TRY(handles[0], ncclCommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) { handles[i]->comm = comms[(size_t)i]; handles[i]->rank = i; handles[i]->world = n; }
    return TP_OK;
}

// The gather of tp_batch_gather for the single-process communicator: batches[i] lives on rank i, all with the same
// W; every rank's ncclGather pair is issued inside ONE group (a single thread drives all the devices), each on its
// handle's kernel stream.  Waits for root's stream; the result stays in root's HBM and, with host buffers given,
// is copied out [n x W x k] / [n x W].
int tp_group_gather(tp_batch_t* batches, int n, int root, double* weights_all, int32_t* status_all) {
    if (!batches || n < 1 || root < 0 || root >= n) return TP_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (!batches[i]) return TP_ERR_INVALID;
        tp_handle_t h = batches[i]->h;
        if (!h->comm || h->world != n || h->rank != i)
            return fail(h, TP_ERR_INVALID, "tp_group_gather: batches[%d] is not on rank %d of an %d-rank communicator", i, i, n);
        if (batches[i]->W != batches[0]->W || batches[i]->p.k != batches[0]->p.k)
            return fail(h, TP_ERR_INVALID, "tp_group_gather: every rank must hold the same W and k");
    }
    tp_batch_t rb = batches[root];
    tp_handle_t rh = rb->h;
    const size_t nw = (size_t)rb->W * rb->p.k, ns = (size_t)rb->W;
    HIP_TRY(rh, hipSetDevice(rh->device));
    int rc = ensure(rh, rb->gather_w, sizeof(double) * nw * n);
    if (rc == TP_OK) rc = ensure(rh, rb->gather_s, sizeof(int32_t) * ns * n);
    if (rc != TP_OK) return rc;
    NCCL_TRY(rh, ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        tp_batch_t b = batches[i];
        tp_handle_t h = b->h;
        const bool is_root = i == root;
        ncclResult_t r1 = ncclGather(b->out_weights(), is_root ? b->gather_w.p : nullptr, nw, ncclDouble, root, h->comm, h->stream);
        ncclResult_t r2 = r1 == ncclSuccess
            ? ncclGather(b->out_status(), is_root ? b->gather_s.p : nullptr, ns, ncclInt32, root, h->comm, h->stream) : r1;
        if (r2 != ncclSuccess) { (void)ncclGroupEnd(); return fail(rh, TP_ERR_RCCL, "ncclGather (rank %d) failed: %s", i, ncclGetErrorString(r2)); }
    }
    NCCL_TRY(rh, ncclGroupEnd());
    for (int i = 0; i < n; ++i) {
        tp_handle_t h = batches[i]->h;
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        int rck = harvest_kernel_time(h);
        if (rck != TP_OK) return rck;
    }
    rb->gathered = true;
    if (weights_all || status_all) return tp_batch_download_gathered(rb, weights_all, status_all);
    return TP_OK;
}

int tp_comm_destroy(tp_handle_t h) {
    if (!h) return TP_ERR_INVALID;
    if (h->deferred) { int rcf = flush_gather(h); if (rcf != TP_OK) return rcf; }
    if (h->comm_stream) { HIP_TRY(h, hipSetDevice(h->device)); HIP_TRY(h, hipStreamSynchronize(h->comm_stream)); }
    if (h->comm) { NCCL_TRY(h, ncclCommDestroy(h->comm)); h->comm = nullptr; }
    h->world = 1; h->rank = 0;
    return TP_OK;
}

int tp_batch_gather(tp_batch_t b, int root, double* weights_all, int32_t* status_all) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    if (!h->comm) return fail(h, TP_ERR_INVALID, "tp_batch_gather without tp_comm_init");
    if (root < 0 || root >= h->world) return fail(h, TP_ERR_INVALID, "bad root %d", root);
    HIP_TRY(h, hipSetDevice(h->device));
    // collectives of one communicator must be issued in the same order on every rank: a gather still waiting to
    // go onto the gather stream comes first, and this one only after it has finished there
    if (h->deferred) { int rcf = flush_gather(h); if (rcf != TP_OK) return rcf; }
    if (h->comm_stream) HIP_TRY(h, hipStreamSynchronize(h->comm_stream));
    const size_t nw = (size_t)b->W * b->p.k, ns = (size_t)b->W;
    const bool is_root = h->rank == root;
    if (is_root) {
        int rc = ensure(h, b->gather_w, sizeof(double) * nw * h->world);
        if (rc == TP_OK) rc = ensure(h, b->gather_s, sizeof(int32_t) * ns * h->world);
        if (rc != TP_OK) return rc;
    }
    hipEvent_t g0 = nullptr, g1 = nullptr;
    HIP_TRY(h, hipEventCreate(&g0));
    HIP_TRY(h, hipEventCreate(&g1));
    HIP_TRY(h, hipEventRecord(g0, h->stream));
    // one gather of the weights (and one of the statuses) to root, on the stream of the kernel
    NCCL_TRY(h, ncclGroupStart());
    NCCL_TRY(h, ncclGather(b->out_weights(), is_root ? b->gather_w.p : nullptr, nw, ncclDouble, root, h->comm, h->stream));
    NCCL_TRY(h, ncclGather(b->out_status(), is_root ? b->gather_s.p : nullptr, ns, ncclInt32, root, h->comm, h->stream));
    NCCL_TRY(h, ncclGroupEnd());
    HIP_TRY(h, hipEventRecord(g1, h->stream));
    HIP_TRY(h, hipEventSynchronize(g1));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, g0, g1));
    h->gather_ms = ms;
    (void)hipEventDestroy(g0);
    (void)hipEventDestroy(g1);
    int rc = harvest_kernel_time(h);
    if (rc != TP_OK) return rc;
    b->gathered = true;
    if (is_root && (weights_all || status_all)) return tp_batch_download_gathered(b, weights_all, status_all);
    return TP_OK;
}

// Put the requested gather (tp_batch_gather_async) on the gather stream.  Called with the NEXT kernel already
// queued (tp_batch_run) or when the caller waits anyway: the host waits for the end of the run whose results
// are gathered, so the gather stream needs no device-side wait for the kernel stream.
static int flush_gather(tp_handle_t h) {
    tp_batch_t b = h->deferred;
    if (!b || !b->gather_req) { h->deferred = nullptr; return TP_OK; }
    HIP_TRY(h, hipSetDevice(h->device));
    const int root = b->gather_root, par = b->gather_req_parity;
    const size_t nw = (size_t)b->W * b->p.k, ns = (size_t)b->W;
    const bool is_root = h->rank == root;
    HIP_TRY(h, hipEventSynchronize(b->snap));
    // timing events: re-record them only when the previous pair has been read or is already complete
    bool time_this = true;
    if (h->gather_timed) {
        if (hipEventQuery(h->cg1) == hipSuccess) { int rc = harvest_gather_time(h); if (rc != TP_OK) return rc; }
        else time_this = false;
    }
    const double* sw = (const double*)(par ? b->weights2.p : b->weights.p);
    const int32_t* ss = (const int32_t*)(par ? b->status2.p : b->status.p);
    if (time_this) HIP_TRY(h, hipEventRecord(h->cg0, h->comm_stream));
    NCCL_TRY(h, ncclGroupStart());
    NCCL_TRY(h, ncclGather(sw, is_root ? b->gather_w.p : nullptr, nw, ncclDouble, root, h->comm, h->comm_stream));
    NCCL_TRY(h, ncclGather(ss, is_root ? b->gather_s.p : nullptr, ns, ncclInt32, root, h->comm, h->comm_stream));
    NCCL_TRY(h, ncclGroupEnd());
    if (time_this) { HIP_TRY(h, hipEventRecord(h->cg1, h->comm_stream)); h->gather_timed = true; }
    HIP_TRY(h, hipEventRecord(b->gather_done[par], h->comm_stream));
    b->gather_pending[par] = true;
    b->gather_req = false;
    h->deferred = nullptr;
    return TP_OK;
}

int tp_batch_gather_async(tp_batch_t b, int root) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    if (!h->comm) return fail(h, TP_ERR_INVALID, "tp_batch_gather_async without tp_comm_init");
    if (root < 0 || root >= h->world) return fail(h, TP_ERR_INVALID, "bad root %d", root);
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->comm_stream) {
        int lo = 0, hi = 0;                            // numerically lower = higher priority
        HIP_TRY(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(h, hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, hi));
        HIP_TRY(h, hipEventCreate(&h->cg0));
        HIP_TRY(h, hipEventCreate(&h->cg1));
    }
    const size_t nw = (size_t)b->W * b->p.k, ns = (size_t)b->W;
    const bool is_root = h->rank == root;
    int rc = TP_OK;
    if (h->deferred) { rc = flush_gather(h); if (rc != TP_OK) return rc; }     // an earlier request nobody ran after
    if (!b->pingpong) {                                // first use: the second result pair and its events
        rc = ensure(h, b->weights2, sizeof(double) * nw);
        if (rc == TP_OK) rc = ensure(h, b->status2, sizeof(int32_t) * ns);
        if (rc != TP_OK) return rc;
        for (hipEvent_t& e : b->gather_done) HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&b->snap, hipEventDisableTiming));
        b->pingpong = true;
    }
    if (is_root) rc = ensure(h, b->gather_w, sizeof(double) * nw * h->world);
    if (rc == TP_OK && is_root) rc = ensure(h, b->gather_s, sizeof(int32_t) * ns * h->world);
    if (rc != TP_OK) return rc;
    // Only a request: the gather goes onto its stream inside the NEXT tp_batch_run, after that run's kernel
    // is queued (or in tp_synchronize / tp_batch_download_gathered) - see flush_gather.
    HIP_TRY(h, hipEventRecord(b->snap, h->stream));
    b->gather_req = true;
    b->gather_req_parity = b->parity;
    b->gather_root = root;
    h->deferred = b;
    b->gathered = true;
    return TP_OK;
}

int tp_batch_download_gathered(tp_batch_t b, double* weights_all, int32_t* status_all) {
    if (!b) return TP_ERR_INVALID;
    tp_handle_t h = b->h;
    if (!b->gathered || !b->gather_w.p) return fail(h, TP_ERR_INVALID, "nothing gathered on this rank (root only, after tp_batch_gather)");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->deferred) { int rcf = flush_gather(h); if (rcf != TP_OK) return rcf; }
    if (h->comm_stream) {                              // an asynchronous gather may still be filling gather_w
        HIP_TRY(h, hipStreamSynchronize(h->comm_stream));
    }
    const size_t nw = (size_t)b->W * b->p.k, ns = (size_t)b->W;
    if (weights_all) HIP_TRY(h, hipMemcpyAsync(weights_all, b->gather_w.p, sizeof(double) * nw * h->world, hipMemcpyDeviceToHost, h->stream));
    if (status_all) HIP_TRY(h, hipMemcpyAsync(status_all, b->gather_s.p, sizeof(int32_t) * ns * h->world, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TP_OK;
}

}  // extern "C"
