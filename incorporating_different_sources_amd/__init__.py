"""MI355X-native rolling-window Bayesian tangency-portfolio posterior.

Host modules keep the reference's call surface (`portfolio_specs`, `portfolio_calculations`); the
arithmetic of the hot path runs in hand-written HIP kernels behind a C-ABI (`libtangency.so`,
`include/tangency_posterior.h`).  There is no CPU fallback.
"""
__version__ = "0.6.0"   # = the number in tp_version() (tests/test_cabi_symbols.py checks)
