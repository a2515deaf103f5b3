"""ctypes binding of libtangency.so (include/tangency_posterior.h).

This is the ONLY compute path of the package: there is no CPU fallback.  Importing the module loads
the shared library (so that a missing build fails loudly); creating a `Device` needs a gfx950 GPU.
"""
from __future__ import annotations

import atexit
import ctypes
import os
import sys
import threading
import weakref
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TANGENCY_LIB selects another build of the same library (tuning experiments); default: the in-tree one
LIB_PATH = os.environ.get("TANGENCY_LIB") or os.path.join(_HERE, "libtangency.so")

TP_OK = 0
TP_ERR_INVALID = -1
TP_ERR_NO_DEVICE = -2
TP_ERR_HIP = -3
TP_ERR_UNSUPPORTED = -4
TP_ERR_RCCL = -5
STRATEGY_CONJUGATE = 0
STRATEGY_JEFFREYS = 1
STATUS_OK, STATUS_NOT_PD, STATUS_NONFINITE, STATUS_BAD_DENOM = 0, 1, 2, 3
AUX_STRIDE = 8
FLAG_CENTER_BY_ROWS = 1
FLAG_NO_CENTER = 2
FLAG_NO_SHARED_GRAM = 4
UNIQUE_ID_BYTES = 128

# every symbol include/tangency_posterior.h declares (checked by tests/test_cabi_symbols.py)
EXPORTS = [
    "tp_version", "tp_max_assets", "tp_device_count", "tp_create", "tp_destroy", "tp_set_option", "tp_last_error",
    "tp_device_info",
    "tp_log_returns", "tp_batch_create", "tp_batch_upload", "tp_batch_upload_async", "tp_batch_upload_wait",
    "tp_batch_shared_gram_blocks",
    "tp_batch_shared_intraday_blocks",
    "tp_host_alloc", "tp_host_free", "tp_batch_set_rhs", "tp_batch_set_shift", "tp_batch_keep_rhs",
    "tp_batch_download_rhs", "tp_batch_run", "tp_batch_download", "tp_batch_download_S1", "tp_batch_download_matrix",
    "tp_batch_debug_stamps", "tp_batch_destroy", "tp_posterior_batch", "tp_synchronize", "tp_last_timing",
    "tp_region_begin", "tp_region_end", "tp_region_steps", "tp_last_launch", "tp_comm_unique_id", "tp_comm_init", "tp_comm_destroy",
    "tp_comm_count", "tp_comm_init_all", "tp_group_gather",
    "tp_batch_gather", "tp_batch_gather_async", "tp_batch_download_gathered",
]


class TangencyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libtangency error {code}: {msg}")
        self.code = code


class tp_params_t(ctypes.Structure):
    _fields_ = [("k", c_int32), ("N", c_int32), ("n_r", c_int32), ("m", c_int32), ("strategy", c_int32),
                ("flags", c_int32), ("gamma", c_double)]


class tp_inputs_t(ctypes.Structure):
    _fields_ = [("panel", POINTER(c_double)), ("panel_rows", c_int64), ("panel_ld", c_int32), ("hf_ld", c_int32),
                ("start", POINTER(c_int64)), ("row_idx", POINTER(c_int32)), ("n_rows", POINTER(c_int32)),
                ("col_idx", POINTER(c_int32)), ("rf_adj", POINTER(c_double)),
                ("hf_panel", POINTER(c_double)), ("hf_rows", c_int64),
                ("hf_start", POINTER(c_int64)), ("hf_row_idx", POINTER(c_int32)), ("hf_count", POINTER(c_int32)),
                ("w0", POINTER(c_double)), ("n0", POINTER(c_double)),
                ("ret_num", POINTER(c_int32)), ("ret_den", POINTER(c_int32)), ("ret_rows", c_int64),
                ("hf_ret_num", POINTER(c_int32)), ("hf_ret_den", POINTER(c_int32)), ("hf_ret_rows", c_int64)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C incorporating_different_sources_amd/csrc`.  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.tp_version.restype = c_char_p
    lib.tp_last_error.restype = c_char_p
    lib.tp_last_error.argtypes = [c_void_p]
    lib.tp_max_assets.restype = c_int
    lib.tp_create.argtypes = [c_int, POINTER(c_void_p)]
    lib.tp_destroy.argtypes = [c_void_p]
    lib.tp_set_option.argtypes = [c_void_p, c_char_p, c_int]
    lib.tp_device_info.argtypes = [c_void_p, c_char_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int64)]
    lib.tp_log_returns.argtypes = [c_void_p, POINTER(c_double), c_int64, c_int32, POINTER(c_int32), POINTER(c_int32),
                                   c_int64, POINTER(c_double)]
    lib.tp_batch_create.argtypes = [c_void_p, POINTER(tp_params_t), c_int64, POINTER(c_void_p)]
    lib.tp_batch_upload.argtypes = [c_void_p, POINTER(tp_inputs_t)]
    lib.tp_batch_upload_async.argtypes = [c_void_p, POINTER(tp_inputs_t)]
    lib.tp_batch_upload_wait.argtypes = [c_void_p]
    lib.tp_batch_shared_gram_blocks.argtypes = [c_void_p]
    lib.tp_batch_shared_intraday_blocks.argtypes = [c_void_p]
    lib.tp_host_alloc.argtypes = [POINTER(c_void_p), c_int64]
    lib.tp_host_free.argtypes = [c_void_p]
    lib.tp_batch_keep_rhs.argtypes = [c_void_p, c_int]
    lib.tp_comm_count.argtypes = [c_void_p, POINTER(c_int)]
    lib.tp_comm_init_all.argtypes = [POINTER(c_void_p), c_int]
    lib.tp_group_gather.argtypes = [POINTER(c_void_p), c_int, c_int, POINTER(c_double), POINTER(c_int32)]
    lib.tp_batch_set_rhs.argtypes = [c_void_p, POINTER(c_double)]
    lib.tp_batch_set_shift.argtypes = [c_void_p, POINTER(c_double)]
    lib.tp_batch_download_rhs.argtypes = [c_void_p, POINTER(c_double)]
    lib.tp_batch_run.argtypes = [c_void_p]
    lib.tp_batch_download.argtypes = [c_void_p, POINTER(c_double), POINTER(c_int32), POINTER(c_double)]
    lib.tp_batch_download_S1.argtypes = [c_void_p, c_int64, POINTER(c_double)]
    lib.tp_batch_download_matrix.argtypes = [c_void_p, c_int64, c_int, POINTER(c_double), POINTER(c_double)]
    lib.tp_batch_debug_stamps.argtypes = [c_void_p, POINTER(c_int64)]
    lib.tp_batch_destroy.argtypes = [c_void_p]
    lib.tp_posterior_batch.argtypes = [c_void_p, POINTER(tp_params_t), c_int64, POINTER(tp_inputs_t),
                                       POINTER(c_double), POINTER(c_int32), POINTER(c_double)]
    lib.tp_synchronize.argtypes = [c_void_p]
    lib.tp_last_timing.argtypes = [c_void_p] + [POINTER(c_double)] * 4
    lib.tp_region_begin.argtypes = [c_void_p]
    lib.tp_region_end.argtypes = [c_void_p, POINTER(c_double)]
    lib.tp_region_steps.argtypes = [c_void_p, POINTER(c_double), c_int, POINTER(c_int)]
    lib.tp_last_launch.argtypes = [c_void_p] + [POINTER(c_int)] * 4
    lib.tp_comm_unique_id.argtypes = [c_void_p]
    lib.tp_comm_init.argtypes = [c_void_p, c_void_p, c_int, c_int]
    lib.tp_comm_destroy.argtypes = [c_void_p]
    lib.tp_batch_gather.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(c_int32)]
    lib.tp_batch_gather_async.argtypes = [c_void_p, c_int]
    lib.tp_batch_download_gathered.argtypes = [c_void_p, POINTER(c_double), POINTER(c_int32)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is not c_char_p:
            fn.restype = c_int
    return lib


lib = _load()


def version() -> str:
    return lib.tp_version().decode()


def max_assets() -> int:
    return int(lib.tp_max_assets())


def device_count() -> int:
    return int(lib.tp_device_count())


def _ptr(a, ct):
    return None if a is None else a.ctypes.data_as(POINTER(ct))


def _arr(a, dtype, shape=None, name=""):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


# ---- lifetime --------------------------------------------------------------------------------------------------------
# Device memory, streams, events and RCCL communicators must be released while the HIP / RCCL runtimes are still up.
# Objects that outlive their scope - locals of a frame kept by a traceback after a failed backtest, module globals -
# used to reach __del__ during interpreter finalisation or never, and the process then aborted in the runtimes' own
# static teardown (round 2: `std::bad_variant_access`, core dumps after a failing test).  Rule: every live Device and
# pinned block is tracked here and closed by ONE atexit hook (atexit hooks run before module teardown and before any
# C-level exit handler: batches first, then the communicator, streams, the handle); finalisers that run once the
# interpreter is finalising do nothing (the library's own exit handler covers whatever is left, see tp_destroy).
_live_devices: "weakref.WeakSet" = weakref.WeakSet()
_live_pinned: "weakref.WeakSet" = weakref.WeakSet()
_live_lock = threading.Lock()
_closed_for_exit = False


def _finalizing() -> bool:
    return _closed_for_exit or sys.is_finalizing()


def shutdown() -> None:
    """Close every live Device (its batches first) and free every pinned block.  Runs at interpreter exit; safe to
    call earlier (objects created afterwards are tracked again)."""
    with _live_lock:
        devices = list(_live_devices)
        pinned = list(_live_pinned)
    for d in devices:
        try:
            d.close()
        except Exception:
            pass
    for p in pinned:
        try:
            p.free()
        except Exception:
            pass
    global _default_device, _default_group
    _default_device = None
    _default_group = None


def _shutdown_at_exit() -> None:
    global _closed_for_exit
    try:
        shutdown()
    finally:
        _closed_for_exit = True


atexit.register(_shutdown_at_exit)


class _PinnedBlock:
    """Owner of one tp_host_alloc block; numpy arrays made over it keep it alive through `.base`."""

    def __init__(self, nbytes: int):
        self.ptr = c_void_p()
        rc = lib.tp_host_alloc(ctypes.byref(self.ptr), int(nbytes))
        if rc != TP_OK:
            raise TangencyError(rc, f"tp_host_alloc({nbytes}) failed")
        self.nbytes = int(nbytes)
        self.buf = (ctypes.c_char * max(1, self.nbytes)).from_address(self.ptr.value)
        with _live_lock:
            _live_pinned.add(self)

    def free(self):
        if self.ptr:
            lib.tp_host_free(self.ptr)
            self.ptr = c_void_p()

    def __del__(self):
        if _finalizing():        # the runtime may be unloading: leak (the process is ending)
            return
        try:
            self.free()
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """An uninitialised array in page-locked host memory (`tp_host_alloc`): uploads from it and downloads into it
    run at PCIe rate and, with `Batch.upload_async`, without blocking the host."""
    dtype = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    n = int(np.prod(shape)) if shape else 1
    block = _PinnedBlock(n * dtype.itemsize)
    arr = np.frombuffer(block.buf, dtype=dtype, count=n).reshape(shape)
    # np.frombuffer keeps `block.buf` (and through a reference cycle-free attribute the block) alive
    block.buf._owner = block
    return arr


def pinned_copy(a) -> np.ndarray:
    a = np.asarray(a)
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


class Device:
    """One GPU (`tp_handle_t`): a HIP stream, timing events and optionally an RCCL communicator."""

    def __init__(self, device_id: int = 0):
        self._h = c_void_p()
        self._batches = weakref.WeakSet()
        rc = lib.tp_create(int(device_id), ctypes.byref(self._h))
        if rc != TP_OK:
            raise TangencyError(rc, lib.tp_last_error(None).decode())
        self.device_id = device_id
        with _live_lock:
            _live_devices.add(self)

    def set_option(self, name: str, value: int):
        """`tp_set_option`: kernel-selection switches of this handle ("wave_kernel", "tiled_wave", "tiled_fuse",
        "no_shared_gram", "tiled_arena_gib", "tiled_arena_mib"); the TP_* environment variables are read once, when the
        Device is created."""
        self._check(lib.tp_set_option(self._h, name.encode(), int(value)))
        return self

    def _check(self, rc):
        if rc != TP_OK:
            raise TangencyError(rc, lib.tp_last_error(self._h).decode())

    def close(self):
        if self._h:
            for b in list(self._batches):      # a batch must not outlive its handle (tp_destroy would take it down too)
                b.close()
            lib.tp_destroy(self._h)
            self._h = c_void_p()
        with _live_lock:
            _live_devices.discard(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        if _finalizing():        # see "lifetime" above: never call into HIP / RCCL from a finaliser at exit
            return
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        name = ctypes.create_string_buffer(256)
        cu, mhz, hbm = c_int(), c_int(), c_int64()
        self._check(lib.tp_device_info(self._h, name, 256, ctypes.byref(cu), ctypes.byref(mhz), ctypes.byref(hbm)))
        return dict(name=name.value.decode(), compute_units=cu.value, clock_mhz=mhz.value, hbm_bytes=hbm.value)

    def synchronize(self):
        self._check(lib.tp_synchronize(self._h))

    def last_timing(self) -> dict:
        v = [c_double() for _ in range(4)]
        self._check(lib.tp_last_timing(self._h, *[ctypes.byref(x) for x in v]))
        return dict(kernel_ms=v[0].value, h2d_ms=v[1].value, d2h_ms=v[2].value, gather_ms=v[3].value)

    def last_launch(self) -> dict:
        v = [c_int() for _ in range(4)]
        self._check(lib.tp_last_launch(self._h, *[ctypes.byref(x) for x in v]))
        return dict(grid=v[0].value, block=v[1].value, lds_bytes=v[2].value, ntile=v[3].value)

    def region_begin(self):
        self._check(lib.tp_region_begin(self._h))

    def region_end(self) -> float:
        ms = c_double()
        self._check(lib.tp_region_end(self._h, ctypes.byref(ms)))
        return ms.value

    def region_steps(self) -> np.ndarray:
        """Kernel milliseconds of every `run` inside the last region_begin / region_end bracket (HIP events per launch)."""
        n = c_int()
        self._check(lib.tp_region_steps(self._h, None, 0, ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.float64)
        if n.value:
            self._check(lib.tp_region_steps(self._h, _ptr(out, c_double), n.value, ctypes.byref(n)))
        return out

    # ---- RCCL -------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        rc = lib.tp_comm_unique_id(buf)
        if rc != TP_OK:
            raise TangencyError(rc, "ncclGetUniqueId failed")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = ctypes.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        self._check(lib.tp_comm_init(self._h, buf, int(rank), int(world)))
        self.rank, self.world = rank, world

    def comm_destroy(self):
        self._check(lib.tp_comm_destroy(self._h))

    def comm_count(self) -> int:
        """Ranks of the handle's RCCL communicator (ncclCommCount)."""
        n = c_int()
        self._check(lib.tp_comm_count(self._h, ctypes.byref(n)))
        return n.value

    # ---- price front-end ----------------------------------------------------------------------
    def log_returns(self, prices, num, den) -> np.ndarray:
        """out[i] = log(prices[num[i]] / prices[den[i]]) on the device (NaN -> 0), ref:44 / ref:311."""
        P = _arr(prices, np.float64)
        if P.ndim != 2:
            raise ValueError("prices must be 2-D [rows x assets]")
        num, den = _arr(num, np.int32), _arr(den, np.int32)
        if num.ndim != 1 or num.shape != den.shape:
            raise ValueError("num / den: two equally long 1-D index arrays expected")
        out = np.empty((num.size, P.shape[1]), dtype=np.float64)
        self._check(lib.tp_log_returns(self._h, _ptr(P, c_double), P.shape[0], P.shape[1], _ptr(num, c_int32),
                                       _ptr(den, c_int32), num.size, _ptr(out, c_double)))
        return out

    # ---- batches ------------------------------------------------------------------------------
    def batch(self, strategy, k, N, n_r, gamma, W, m=0, flags=0) -> "Batch":
        return Batch(self, strategy, k, N, n_r, gamma, W, m, flags)


class Batch:
    """W windows resident in HBM (`tp_batch_t`)."""

    def __init__(self, dev: Device, strategy, k, N, n_r, gamma, W, m=0, flags=0):
        self.dev = dev
        strat = {"conjugate": STRATEGY_CONJUGATE, "jeffreys": STRATEGY_JEFFREYS}.get(strategy, strategy)
        self.params = tp_params_t(int(k), int(N), int(n_r), int(m), int(strat), int(flags), float(gamma))
        self.W, self.k, self.n_r, self.m = int(W), int(k), int(n_r), int(m)
        self._b = c_void_p()
        dev._check(lib.tp_batch_create(dev._h, ctypes.byref(self.params), self.W, ctypes.byref(self._b)))
        dev._batches.add(self)
        self._keep = None

    def close(self):
        if self._b:
            if self.dev._h:                    # a closed Device took its batches with it
                lib.tp_batch_destroy(self._b)
            self._b = c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        if _finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def upload(self, panel, start=None, hf_panel=None, hf_start=None, w0=None, n0=None, row_idx=None,
               n_rows=None, col_idx=None, rf_adj=None, hf_row_idx=None, hf_count=None, ret_pairs=None,
               hf_ret_pairs=None, asynchronous=False):
        """H2D.  `ret_pairs=(num, den)`: `panel` holds PRICES and the device forms the log-return panel
        R[i] = log(P[num[i]] / P[den[i]]) that start / row_idx address; `hf_ret_pairs` likewise for `hf_panel`."""
        W, k, n_r, m = self.W, self.k, self.n_r, self.m
        panel = _arr(panel, np.float64)
        if panel.ndim != 2:
            raise ValueError("panel must be 2-D [rows x assets]")
        hf_panel = _arr(hf_panel, np.float64)
        if hf_panel is not None and hf_panel.ndim != 2:
            raise ValueError("hf_panel must be 2-D [rows x assets]")
        a = dict(
            panel=panel, hf_panel=hf_panel,
            start=_arr(start, np.int64, (W,), "start"), row_idx=_arr(row_idx, np.int32, (W, n_r), "row_idx"),
            n_rows=_arr(n_rows, np.int32, (W,), "n_rows"), col_idx=_arr(col_idx, np.int32, (W, k), "col_idx"),
            rf_adj=_arr(rf_adj, np.float64, (W, n_r), "rf_adj"),
            hf_start=_arr(hf_start, np.int64, (W,), "hf_start"),
            hf_row_idx=_arr(hf_row_idx, np.int32, (W, m), "hf_row_idx"),
            hf_count=_arr(hf_count, np.int32, (W,), "hf_count"),
            w0=_arr(w0, np.float64, (W, k), "w0"), n0=_arr(n0, np.float64, (W,), "n0"))
        for name, pairs in (("ret", ret_pairs), ("hf_ret", hf_ret_pairs)):
            num = _arr(pairs[0], np.int32) if pairs is not None else None
            den = _arr(pairs[1], np.int32) if pairs is not None else None
            if pairs is not None and (num.ndim != 1 or num.shape != den.shape or num.size < 1):
                raise ValueError(f"{name}_pairs: two equally long 1-D index arrays expected")
            a[name + "_num"], a[name + "_den"] = num, den
        inp = tp_inputs_t(
            _ptr(a["panel"], c_double), panel.shape[0], panel.shape[1],
            hf_panel.shape[1] if hf_panel is not None else 0,
            _ptr(a["start"], c_int64), _ptr(a["row_idx"], c_int32), _ptr(a["n_rows"], c_int32),
            _ptr(a["col_idx"], c_int32), _ptr(a["rf_adj"], c_double),
            _ptr(a["hf_panel"], c_double), hf_panel.shape[0] if hf_panel is not None else 0,
            _ptr(a["hf_start"], c_int64), _ptr(a["hf_row_idx"], c_int32), _ptr(a["hf_count"], c_int32),
            _ptr(a["w0"], c_double), _ptr(a["n0"], c_double),
            _ptr(a["ret_num"], c_int32), _ptr(a["ret_den"], c_int32), a["ret_num"].size if a["ret_num"] is not None else 0,
            _ptr(a["hf_ret_num"], c_int32), _ptr(a["hf_ret_den"], c_int32),
            a["hf_ret_num"].size if a["hf_ret_num"] is not None else 0)
        self._keep = a     # host arrays stay alive for the duration of the upload
        if asynchronous:
            self.dev._check(lib.tp_batch_upload_async(self._b, ctypes.byref(inp)))   # `upload_wait` releases them
        else:
            self.dev._check(lib.tp_batch_upload(self._b, ctypes.byref(inp)))
            self._keep = None
        return self

    def upload_async(self, panel, **kw):
        """`upload` queued on the device's copy stream and not waited for (arrays from `pinned_empty` make the
        copies truly asynchronous).  The next `run` waits for them on the device; `upload_wait` on the host."""
        return self.upload(panel, asynchronous=True, **kw)

    def upload_wait(self):
        self.dev._check(lib.tp_batch_upload_wait(self._b))
        self._keep = None
        return self

    def shared_gram_blocks(self) -> int:
        """Aligned row blocks of the daily panel whose Gram sums this batch's windows share (0: none)."""
        return int(lib.tp_batch_shared_gram_blocks(self._b))

    def shared_intraday_blocks(self) -> int:
        """Large-k path, conjugate: whole intraday blocks (days) per window taken from shared block Grams (0: none)."""
        return int(lib.tp_batch_shared_intraday_blocks(self._b))

    def keep_rhs(self, on=True):
        """Keep the right-hand side every window is solved for in later runs (`download_rhs` reads the last run's)."""
        self.dev._check(lib.tp_batch_keep_rhs(self._b, 1 if on else 0))
        return self

    def set_rhs(self, rhs):
        """Right-hand side [W x k] in place of the border column (None: default t / c S0 w0 + t)."""
        r = _arr(rhs, np.float64, (self.W, self.k), "rhs")
        self.dev._check(lib.tp_batch_set_rhs(self._b, _ptr(r, c_double)))
        return self

    def set_shift(self, shift):
        """Per-window (d, e) [W x 2]: the Jeffreys matrix becomes J + d I + e 1 1' (None: no shift)."""
        sh = _arr(shift, np.float64, (self.W, 2), "shift")
        self.dev._check(lib.tp_batch_set_shift(self._b, _ptr(sh, c_double)))
        return self

    def download_rhs(self) -> np.ndarray:
        """[W x k] right-hand sides the windows were solved for in the last run (t = X'1 by default); needs
        `keep_rhs()` before that run."""
        out = np.empty((self.W, self.k), dtype=np.float64)
        self.dev._check(lib.tp_batch_download_rhs(self._b, _ptr(out, c_double)))
        return out

    def run(self):
        self.dev._check(lib.tp_batch_run(self._b))
        return self

    def download(self, want_aux=True, out=None):
        """(weights, status, aux).  `out=(weights, status[, aux])`: write into these arrays (e.g. `pinned_empty`)."""
        if out is not None:
            weights, status = out[0], out[1]
            aux = out[2] if len(out) > 2 else None
            if weights.shape != (self.W, self.k) or weights.dtype != np.float64 or not weights.flags.c_contiguous:
                raise ValueError("out[0]: C-contiguous float64 [W x k] expected")
            if status.shape != (self.W,) or status.dtype != np.int32:
                raise ValueError("out[1]: int32 [W] expected")
            if aux is not None and (aux.shape != (self.W, AUX_STRIDE) or aux.dtype != np.float64):
                raise ValueError("out[2]: float64 [W x 8] expected")
            self.dev._check(lib.tp_batch_download(self._b, _ptr(weights, c_double), _ptr(status, c_int32),
                                                  _ptr(aux, c_double)))
            return weights, status, aux
        weights = np.empty((self.W, self.k), dtype=np.float64)
        status = np.empty(self.W, dtype=np.int32)
        aux = np.empty((self.W, AUX_STRIDE), dtype=np.float64) if want_aux else None
        self.dev._check(lib.tp_batch_download(self._b, _ptr(weights, c_double), _ptr(status, c_int32),
                                              _ptr(aux, c_double)))
        return weights, status, aux

    def download_S1(self, w: int) -> np.ndarray:
        S1 = np.empty((self.k, self.k), dtype=np.float64)
        self.dev._check(lib.tp_batch_download_S1(self._b, int(w), _ptr(S1, c_double)))
        return S1

    def download_matrix(self, w: int, what):
        """(M [k x k], rhs [k]) of window w: what = 'prior' (S0, c S0 w0), 'gram' (T, t) or
        'posterior' (S1 or J, right-hand side)."""
        code = {"prior": 1, "gram": 2, "posterior": 3}.get(what, what)
        M = np.empty((self.k, self.k), dtype=np.float64)
        rhs = np.empty(self.k, dtype=np.float64)
        self.dev._check(lib.tp_batch_download_matrix(self._b, int(w), int(code), _ptr(M, c_double), _ptr(rhs, c_double)))
        return M, rhs

    def debug_stamps(self) -> np.ndarray:
        """[W x 8] shader-clock stamps at the kernel's phase boundaries (TP_STAMP builds only)."""
        st = np.zeros((self.W, 40), dtype=np.int64)
        self.dev._check(lib.tp_batch_debug_stamps(self._b, _ptr(st, c_int64)))
        return st

    def gather(self, root=0, to_host=True):
        """One RCCL gather of every rank's [W x k] weights (and statuses) to `root`.  With
        `to_host=False` the result stays in root's HBM (`download_gathered` fetches it later)."""
        world, rank = self.dev.world, self.dev.rank
        self._gather_root = root
        if rank == root and to_host:
            wall = np.empty((world, self.W, self.k), dtype=np.float64)
            sall = np.empty((world, self.W), dtype=np.int32)
            self.dev._check(lib.tp_batch_gather(self._b, root, _ptr(wall, c_double), _ptr(sall, c_int32)))
            return wall, sall
        self.dev._check(lib.tp_batch_gather(self._b, root, None, None))
        return None, None

    def gather_async(self, root=0):
        """The same gather on the handle's second stream, not waited for: it overlaps the next `run`.
        `Device.synchronize()` or `download_gathered()` wait for it."""
        self._gather_root = root
        self.dev._check(lib.tp_batch_gather_async(self._b, root))
        return self

    def download_gathered(self):
        world = self.dev.world
        wall = np.empty((world, self.W, self.k), dtype=np.float64)
        sall = np.empty((world, self.W), dtype=np.int32)
        self.dev._check(lib.tp_batch_download_gathered(self._b, _ptr(wall, c_double), _ptr(sall, c_int32)))
        return wall, sall


class DeviceGroup:
    """Every visible GPU of THIS process under one RCCL communicator (`tp_comm_init_all`): the single-process
    multi-device mode behind `backtest_portfolio` (the reference's main.py is one process, src/main.py:26).
    Windows shard contiguously over the devices (`shard.run_sharded`); one grouped gather brings the weights to
    device 0."""

    def __init__(self, device_ids=None):
        ids = list(range(device_count())) if device_ids is None else list(device_ids)
        if not ids:
            raise TangencyError(TP_ERR_NO_DEVICE, "no HIP device available; there is no CPU fallback")
        self.devices = [Device(i) for i in ids]
        self.world = len(self.devices)
        if self.world > 1:
            arr = (c_void_p * self.world)(*[d._h for d in self.devices])
            rc = lib.tp_comm_init_all(arr, self.world)
            if rc != TP_OK:
                raise TangencyError(rc, lib.tp_last_error(self.devices[0]._h).decode())
            for r, d in enumerate(self.devices):
                d.rank, d.world = r, self.world

    def gather(self, batches, root=0):
        """One grouped RCCL gather of the equally sized batches (batches[i] on devices[i]) to `root`: returns
        (weights [world x W x k], status [world x W]) on the host."""
        n = len(batches)
        if n != self.world:
            raise ValueError("one batch per device expected")
        if n == 1:
            w, s, _ = batches[0].download(want_aux=False)
            return w[None], s[None]
        W, k = batches[0].W, batches[0].k
        wall = np.empty((n, W, k), dtype=np.float64)
        sall = np.empty((n, W), dtype=np.int32)
        arr = (c_void_p * n)(*[b._b for b in batches])
        rc = lib.tp_group_gather(arr, n, int(root), _ptr(wall, c_double), _ptr(sall, c_int32))
        if rc != TP_OK:
            raise TangencyError(rc, lib.tp_last_error(self.devices[root]._h).decode())
        return wall, sall

    def close(self):
        for d in self.devices:
            d.close()
        self.devices = []


_default_device = None
_default_group = None


def default_group() -> DeviceGroup:
    """All visible GPUs of this process (created on first use; `TP_DEVICES=0,1,..` restricts them)."""
    global _default_group
    if _default_group is None:
        ids = os.environ.get("TP_DEVICES")
        _default_group = DeviceGroup([int(x) for x in ids.split(",")] if ids else None)
    return _default_group


def default_device() -> Device:
    global _default_device
    if _default_device is None:
        _default_device = Device(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("TP_DEVICE") is None
                                 else int(os.environ["TP_DEVICE"]))
    return _default_device


def posterior_batch(strategy, k, N, gamma, panel, start=None, n_r=None, hf_panel=None, hf_start=None, m=0,
                    w0=None, n0=None, row_idx=None, n_rows=None, col_idx=None, rf_adj=None,
                    hf_row_idx=None, hf_count=None, device: Device | None = None, want_aux=True, rhs=None, flags=0,
                    shift=None, ret_pairs=None, hf_ret_pairs=None):
    """Upload + run + download.  Same argument meaning as `oracle.posterior_batch` (tests compare them)."""
    dev = device or default_device()
    W = len(start) if start is not None else len(row_idx)
    b = Batch(dev, strategy, k, N, n_r, gamma, W, m or 0, flags)
    try:
        if rhs is not None:
            b.set_rhs(rhs)
        if shift is not None:
            b.set_shift(shift)
        b.upload(panel, start=start, hf_panel=hf_panel, hf_start=hf_start, w0=w0, n0=n0, row_idx=row_idx,
                 n_rows=n_rows, col_idx=col_idx, rf_adj=rf_adj, hf_row_idx=hf_row_idx, hf_count=hf_count,
                 ret_pairs=ret_pairs, hf_ret_pairs=hf_ret_pairs)
        b.run()
        return b.download(want_aux=want_aux)
    finally:
        b.close()
