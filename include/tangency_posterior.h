/* tangency_posterior.h - C-ABI of libtangency.so: the MI355X (gfx950) implementation of the
 * rolling-window Bayesian tangency-portfolio posterior.
 *
 * The reference (vilnik/incorporating-different-sources) has no FFI: its boundary is the Python
 * module surface of src/portfolio_calculations.py.  The entry points below are what that module's
 * hot path binds through ctypes in this build (incorporating_different_sources_amd/_native.py);
 * each cites the reference interface it replaces as ref:LINE of src/portfolio_calculations.py.
 *
 * One "window" = one rebalancing date.  For every window w the library computes, in fp64,
 *
 *   X_w  (n_r x k)  rows of the daily excess-log-return panel            ref:31-62, 136-161
 *   T = X'X, t = X'1                                                     ref:163-245
 *   Y_w  (m x k)    rows of the intraday log-return panel                ref:310-314
 *   S0 = n0 * m/(m-1) * (Y-Ybar)'(Y-Ybar)                                ref:317-318, 333
 *   c  = 2 n0 / (a + sqrt(a^2 + 4 n0 w0'S0 w0)),  a = n0 + k + 2         ref:415-418
 *   S1 = S0 + T ;  w1 = S1^-1 (c S0 w0 + t) ;  n1 = n0 + N               ref:358, 485-489, 282
 *   weights = (n1 + k + 2) w1 / (n1 - w1'S1 w1) / gamma                  ref:572-575, 836
 * or, for TP_STRATEGY_JEFFREYS,
 *   weights = (T - t t'/N)^-1 t / gamma                                  ref:600-606, 849
 *
 * Everything is plain C: caller-allocated buffers, int return codes, no exceptions.  All arrays are
 * row-major with the asset index contiguous.  A handle owns one GPU, one HIP stream and (optionally)
 * one RCCL communicator; calls on one handle must be serialised by the caller.
 * There is NO CPU fallback: tp_create fails with TP_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef TANGENCY_POSTERIOR_H
#define TANGENCY_POSTERIOR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes */
#define TP_OK 0
#define TP_ERR_INVALID (-1)     /* bad argument (the message is in tp_last_error) */
#define TP_ERR_NO_DEVICE (-2)   /* no usable HIP device */
#define TP_ERR_HIP (-3)         /* a HIP runtime call failed */
#define TP_ERR_UNSUPPORTED (-4) /* shape outside what the kernels cover (see tp_max_assets) */
#define TP_ERR_RCCL (-5)        /* an RCCL call failed */

/* weighting strategies on the path (ref:1012-1034 dispatch) */
#define TP_STRATEGY_CONJUGATE 0 /* calculate_conjugate_hf_mcm_portfolio, ref:819-836 */
#define TP_STRATEGY_JEFFREYS 1  /* calculate_jeffreys_portfolio,         ref:838-849 */

/* per-window status written to status[w] */
#define TP_STATUS_OK 0
#define TP_STATUS_NOT_PD 1    /* non-positive pivot: S1 (or J) is not positive definite (rank-deficient
                                 window; the reference returns finite garbage) */
#define TP_STATUS_NONFINITE 2 /* NaN/Inf in the weights (ref:492-494 raises ValueError) */
#define TP_STATUS_BAD_DENOM 3 /* n1 - w1'S1 w1 <= 0 (ref:573 has no guard, Appendix B-Q9) */

#define TP_AUX_STRIDE 8 /* doubles per window in `aux`: n0, n1, c, q0, q1, n1-q1, 0, 0 */

/* tp_params_t.flags */
#define TP_FLAG_CENTER_BY_ROWS 1 /* Jeffreys scatter T - t t'/n_w with n_w = the rows window w actually
                                   uses instead of N: (n_w - 1) x the sample covariance (ref:876, 917) */
#define TP_FLAG_NO_CENTER 2      /* Jeffreys strategy on the plain Gram matrix T = X'X (no - t t'/N term):
                                   (n-1) S + n xbar xbar' of ref:924 is exactly T */

#define TP_FLAG_NO_SHARED_GRAM 4  /* do not share Gram sums between overlapping windows (contiguous layout): every
                                   window pushes all its rows through the MFMAs, as the index layout always does */

typedef struct tp_handle_s* tp_handle_t;
typedef struct tp_batch_s* tp_batch_t;

/* The reference's portfolio_spec dict (src/portfolio_specs.py:80-90) reduced to what the path reads. */
typedef struct tp_params {
    int32_t k;        /* portfolio_spec["size"]            (ref:415, 572) */
    int32_t N;        /* portfolio_spec["rolling_window"]  (ref:265, 282, 600) */
    int32_t n_r;      /* max rows of excess returns per window (= N-1 without NaN drops, ref:60) */
    int32_t m;        /* max intraday returns per window (conjugate only; ref:314) */
    int32_t strategy; /* TP_STRATEGY_* */
    int32_t flags;    /* TP_FLAG_* */
    double gamma;     /* portfolio_spec["risk_aversion"]   (ref:836, 849) */
} tp_params_t;

/* Host-side description of W windows (all pointers are HOST pointers; optional ones may be NULL).
 * Window w reads
 *   daily rows   row_idx ? row_idx[w*n_r + r] : start[w] + r          for r < (n_rows ? n_rows[w] : n_r)
 *   asset column col_idx ? col_idx[w*k + j]   : j                      for j < k
 *   x[r][j] = panel[row*panel_ld + col] - (rf_adj ? rf_adj[w*n_r + r] : 0)     (ref:57)
 *   intraday rows hf_row_idx ? hf_row_idx[w*m + r] : hf_start[w] + r   for r < (hf_count ? hf_count[w] : m)
 */
typedef struct tp_inputs {
    const double* panel;       /* [panel_rows x panel_ld] daily log-return panel */
    int64_t panel_rows;
    int32_t panel_ld;
    int32_t hf_ld;
    const int64_t* start;      /* [W] first panel row of each window (contiguous mode) */
    const int32_t* row_idx;    /* optional [W x n_r] explicit panel rows (overrides start) */
    const int32_t* n_rows;     /* optional [W] rows actually used (<= n_r) */
    const int32_t* col_idx;    /* optional [W x k] panel columns of the k selected assets */
    const double* rf_adj;      /* optional [W x n_r] per-row risk-free adjustment, ref:48-57 */
    const double* hf_panel;    /* [hf_rows x hf_ld] intraday log-return panel (conjugate) */
    int64_t hf_rows;
    const int64_t* hf_start;   /* [W] */
    const int32_t* hf_row_idx; /* optional [W x m] */
    const int32_t* hf_count;   /* optional [W] intraday returns actually used (<= m, >= 2) */
    const double* w0;          /* [W x k] prior weights, ref:361-380 */
    const double* n0;          /* [W] prior strength, ref:247-267 */
    /* Optional price front-end (ref:31-62, 136-161, 299-314).  With `ret_num` given, `panel` holds PRICES
     * [panel_rows x panel_ld] and the library forms the log-return panel on the device,
     *     R[i][c] = log(P[ret_num[i]][c] / P[ret_den[i]][c])   for i < ret_rows   (NaN -> 0, +-inf -> +-DBL_MAX),
     * e.g. (i, i-1) for daily returns, (bin end, previous bin end) for weekly / monthly resampled windows and
     * (date, last complete bin end) for the running bin.  start / row_idx then address rows of R (and are
     * checked against ret_rows).  `hf_ret_num` does the same for the intraday panel.  Zero-initialise the
     * struct to leave the front-end off: the panels are then log-returns, as above. */
    const int32_t* ret_num;    /* optional [ret_rows] price row in the numerator */
    const int32_t* ret_den;    /* [ret_rows] price row in the denominator (required with ret_num) */
    int64_t ret_rows;
    const int32_t* hf_ret_num; /* optional [hf_ret_rows] */
    const int32_t* hf_ret_den;
    int64_t hf_ret_rows;
} tp_inputs_t;

const char* tp_version(void);
/* largest portfolio_spec["size"] the built kernels cover */
int tp_max_assets(void);
/* number of visible HIP devices (0 when there is none: tp_create will then fail) */
int tp_device_count(void);

/* Contexts.  tp_create binds device `device_id`, creates a stream and timing events.
 * Replaces: nothing in the reference (it has no device); corresponds to process start-up. */
int tp_create(int device_id, tp_handle_t* out);
/* Destroys the handle AND every batch of it that is still alive (their tp_batch_t become invalid).  Lifetime rule
 * for host languages with finalisers: whatever is still alive when the process exits is taken down by the library's
 * own exit handler (registered by the first tp_create, so it runs before the HIP / RCCL runtimes unload: batches
 * first, then the communicator, streams, events); tp_destroy / tp_batch_destroy calls that arrive after that -
 * from finalisers that run during interpreter shutdown - return TP_OK without touching the device or the handle. */
int tp_destroy(tp_handle_t h);
/* Tuning switches of a handle (A/B measurements, tests).  The environment variables of the same meaning
 * (TP_WAVE_KERNEL, TP_TILED_WAVE, TP_TILED_FUSE, TP_NO_SHARED_GRAM, TP_TILED_ARENA_GIB, TP_TILED_ARENA_MIB) are read
 * ONCE, in tp_create; afterwards only this call changes them - no launch reads the environment.
 *   "wave_kernel"      -1 automatic | 0 multi-wave register-tile kernel | 1 one-wave kernel | 2 two-wave kernel
 *   "tiled_wave"       -1 automatic | 0 four-wave Gram / diagonal-block kernels of the large-k path | 2 Gram with
 *                      two super-tiles per wavefront
 *   "tiled_fuse"       -1 automatic | 0 / 1 three-kernel / fused left-looking update of the large-k path
 *   "no_shared_gram"   1 = as if every batch carried TP_FLAG_NO_SHARED_GRAM (takes effect at the next upload)
 *   "tiled_arena_gib" / "tiled_arena_mib"  in-flight arena of the large-k path, per lane (0: default)
 *   "tiled_lanes"      sub-batches of the large-k path in flight at once, each on a stream of its own (0 / 1: one)
 *   "hf_share_min_blocks"  large-k path, conjugate: intraday windows that advance by a fixed stride share the Grams of
 *                      their whole stride-long blocks from this many whole blocks per window on (default 6; takes effect
 *                      at the next upload; "no_shared_gram" switches the scheme off)
 * Replaces nothing in the reference. */
int tp_set_option(tp_handle_t h, const char* name, int value);
const char* tp_last_error(tp_handle_t h); /* h may be NULL: last error of a failed tp_create */
int tp_device_info(tp_handle_t h, char* name, int name_len, int* compute_units, int* clock_mhz,
                   int64_t* hbm_bytes);

/* The price front-end on its own: out[i][c] = log(prices[num[i]][c] / prices[den[i]][c]) for i < n_out, c < ld
 * (NaN -> 0, +-inf -> +-DBL_MAX), host buffers in and out.  Replaces the np.log(prices / prices.shift(1)) of
 * ref:44 (daily or resampled prices) and ref:311 (intraday bars); tp_batch_upload runs the same kernel when
 * tp_inputs_t.ret_num is set, without the copy back. */
int tp_log_returns(tp_handle_t h, const double* prices, int64_t price_rows, int32_t ld, const int32_t* num,
                   const int32_t* den, int64_t n_out, double* out /* [n_out x ld] */);

/* A batch is W windows resident in HBM: inputs uploaded once, run any number of times.
 * Replaces the per-date loop body of Portfolio.update_portfolio -> calculate_portfolio_weights
 * (ref:1184 -> ref:941) for all rebalancing dates of a backtest at once. */
int tp_batch_create(tp_handle_t h, const tp_params_t* p, int64_t W, tp_batch_t* out);
int tp_batch_upload(tp_batch_t b, const tp_inputs_t* in);   /* H2D, synchronous */
/* The same upload queued on the handle's copy stream and not waited for: with the host arrays in page-locked
 * memory (tp_host_alloc) the copies of the NEXT batch run under the kernel of the current one (a backtest that
 * streams batches through tp_batch_run, ref:1232 loop over dates in chunks).  The host arrays must stay valid
 * until tp_batch_upload_wait returns (or the following tp_batch_run has been synchronised); the next tp_batch_run
 * of this batch waits for the copies on the device.  Validation of the index arrays still happens in the call. */
int tp_batch_upload_async(tp_batch_t b, const tp_inputs_t* in);
int tp_batch_upload_wait(tp_batch_t b);                     /* host wait for the queued copies; sets h2d_ms */
/* After an upload: the number of aligned row blocks of the daily panel whose Gram sums the windows of this batch share
 * (rolling windows in the contiguous layout); 0 when every window sums all its own rows. */
int tp_batch_shared_gram_blocks(tp_batch_t b);
/* After an upload, large-k path, conjugate: whole stride-long blocks of intraday rows per window whose Grams the windows
 * of a sub-batch share (0: every intraday row of every window goes through the MFMAs). */
int tp_batch_shared_intraday_blocks(tp_batch_t b);
/* Page-locked host memory for panels and result arrays (hipHostMalloc): DMA at PCIe rate, asynchronous. */
int tp_host_alloc(void** out, int64_t bytes);
int tp_host_free(void* p);
/* Optional right-hand side [W x k] replacing the border column (t for Jeffreys, c S0 w0 + t for the
 * conjugate posterior) in every later tp_batch_run: the weights become (matrix)^-1 rhs / gamma.  NULL
 * restores the default.  Binds the V^-1 1 / V^-1 mu solves of calculate_jorion_portfolio (ref:880-891). */
int tp_batch_set_rhs(tp_batch_t b, const double* rhs);
/* Optional per-window shift [W x 2] = (d_w, e_w), Jeffreys strategy only: the matrix that is factorised
 * becomes J_w + d_w I + e_w 1 1' in every later tp_batch_run (NULL restores the default).  d_w >= 0,
 * e_w >= 0 keep it positive definite.  Binds the scale matrix D_h of calculate_greyserman_portfolio
 * (ref:924: eta_b S_h = eta_b/2 (I + 1 1'), kappa_h xi_b^2 1 1'), one window per posterior draw. */
int tp_batch_set_shift(tp_batch_t b, const double* shift);
int tp_batch_run(tp_batch_t b);                             /* async on the handle's stream; HIP-event timed */
/* Keep (on != 0) the right-hand side each window is solved for in every later tp_batch_run (default: t = X'1,
 * ref:222, resp. c S0 w0 + t, ref:489): tp_batch_download_rhs then copies out what the LAST run used.  It never
 * launches anything itself: without a run after tp_batch_keep_rhs it fails with TP_ERR_INVALID. */
int tp_batch_keep_rhs(tp_batch_t b, int on);
int tp_batch_download_rhs(tp_batch_t b, double* rhs_out /* [W x k] */);
int tp_batch_download(tp_batch_t b, double* weights /* [W x k] */, int32_t* status /* [W] */,
                      double* aux /* optional [W x TP_AUX_STRIDE] */); /* waits for the stream, D2H */
int tp_batch_download_S1(tp_batch_t b, int64_t w, double* S1 /* [k x k] */); /* posterior scale matrix
                      S1 (ref:358) / Jeffreys J (ref:600) of window w, recomputed by a debug launch */
/* Read back one window's k x k matrix (symmetric, full storage) and its k-vector, recomputed by a
 * one-window launch; `rhs` may be NULL.  The reference's same-named helper functions bind these. */
#define TP_MATRIX_PRIOR 1      /* S0 (ref:285-333)            and c S0 w0                     */
#define TP_MATRIX_GRAM 2       /* T  (ref:163-204)            and t (ref:206-245)             */
#define TP_MATRIX_POSTERIOR 3  /* S1 (ref:358) / J (ref:600)  and c S0 w0 + t (ref:489) / t   */
int tp_batch_download_matrix(tp_batch_t b, int64_t w, int what, double* M /* [k x k] */, double* rhs /* [k] */);
/* Diagnostic builds only (make TP_STAMP=1; otherwise TP_ERR_UNSUPPORTED): run the batch once and return
 * 40 values per window: eight shader-clock stamps at the kernel's phase boundaries, then per wave (4)
 * the summed cycles of the four segments of the daily Gram loop (loads | MFMA | LDS write | barrier),
 * then for waves 0 and 1 those of a factorisation block step (hand-over | elimination | TRSM | trailing). */
int tp_batch_debug_stamps(tp_batch_t b, int64_t* stamps /* [W x 40] */);
int tp_batch_destroy(tp_batch_t b);

/* One-shot convenience: upload + run + download.  Replaces
 * calculate_conjugate_hf_mcm_portfolio (ref:819) / calculate_jeffreys_portfolio (ref:838) over W dates. */
int tp_posterior_batch(tp_handle_t h, const tp_params_t* p, int64_t W, const tp_inputs_t* in,
                       double* weights, int32_t* status, double* aux);

int tp_synchronize(tp_handle_t h);
/* Timings of the most recent call of each kind on this handle, in milliseconds (HIP events on the
 * handle's stream): posterior kernel, H2D upload, D2H download, RCCL gather. */
int tp_last_timing(tp_handle_t h, double* kernel_ms, double* h2d_ms, double* d2h_ms, double* gather_ms);
/* HIP-event bracket on the handle's stream around any sequence of tp_batch_run calls (bench.py's
 * timed region): begin records an event, end records another, waits for it and returns the span. */
int tp_region_begin(tp_handle_t h);
int tp_region_end(tp_handle_t h, double* ms);
/* Kernel time of every tp_batch_run inside the last region (each launch is bracketed by its own pair of HIP events on
 * the kernel stream; nothing waits between the steps; at most 512 steps are kept): bench.py's per-step median.
 * step_ms may be NULL (n_steps only). */
int tp_region_steps(tp_handle_t h, double* step_ms, int capacity, int* n_steps);
/* Launch geometry of the most recent tp_batch_run: grid size, threads per workgroup, LDS bytes,
 * 16-column tile count per side. */
int tp_last_launch(tp_handle_t h, int* grid, int* block, int* lds_bytes, int* ntile);

/* Multi-GPU: one process (handle) per GPU; windows are sharded by the caller; the only data-path
 * collective is one gather of the [W_local x k] weights (and statuses) to `root` over RCCL/xGMI.
 * The 128-byte id comes from rank 0 (tp_comm_unique_id) and is distributed by the launcher. */
#define TP_UNIQUE_ID_BYTES 128
int tp_comm_unique_id(void* id /* [TP_UNIQUE_ID_BYTES] */);
int tp_comm_init(tp_handle_t h, const void* id, int rank, int world);
int tp_comm_destroy(tp_handle_t h);
int tp_comm_count(tp_handle_t h, int* ranks);              /* ncclCommCount of the handle's communicator */
/* Single-process form (main.py is ONE process, src/main.py:26): one communicator over the n handles of this
 * process, rank i = handles[i], one handle per GPU (ncclCommInitAll; no id exchange, no launcher).
 * tp_group_gather is tp_batch_gather for it: batches[i] on rank i, all with the same W (pad the last shard),
 * every rank's gather issued in one group by the calling thread; waits; optional host copy-out on the root. */
int tp_comm_init_all(tp_handle_t* handles, int n);
int tp_group_gather(tp_batch_t* batches, int n, int root, double* weights_all /* [n x W x k] or NULL */,
                    int32_t* status_all /* [n x W] or NULL */);
/* Every rank calls it with the same W_local.  The gathered [world x W x k] weights and [world x W]
 * statuses stay in root's HBM; weights_all / status_all are optional HOST buffers on root (NULL: no
 * copy-out, fetch later with tp_batch_download_gathered). */
int tp_batch_gather(tp_batch_t b, int root, double* weights_all, int32_t* status_all);
/* The same gather without waiting for it, on a second, high-priority stream: from its first use the batch
 * keeps TWO result buffers and tp_batch_run alternates between them, so the gather of step i reads one while
 * the kernel of step i+1 writes the other (a rebalancing schedule that streams batches); no copy is made.
 * The call only REQUESTS the gather: it is put on its stream inside the next tp_batch_run, once that run's
 * kernel is queued and the host has seen the previous kernel finish (or in tp_synchronize /
 * tp_batch_download_gathered), so that no stream waits for another stream's event on the device.  Returns
 * at once; tp_synchronize (or tp_batch_download_gathered) waits for the gather.  Gathers complete in the
 * order they were requested; tp_batch_download always reads the results of the last run. */
int tp_batch_gather_async(tp_batch_t b, int root);
int tp_batch_download_gathered(tp_batch_t b, double* weights_all, int32_t* status_all); /* root only */

#ifdef __cplusplus
}
#endif
#endif /* TANGENCY_POSTERIOR_H */
