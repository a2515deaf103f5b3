"""Portfolio-spec grid with the reference's interface (`/root/reference/src/portfolio_specs.py`).

Same three entry points, same dict keys and key-string format, so the reference's `main.py` and
`portfolio_evaluation.py` can import this module in place of theirs:

* `create_portfolio_specs()`                (ref portfolio_specs.py:51-92)
* `get_display_name_from_full_name(name)`   (ref portfolio_specs.py:22-49)
* `get_color_from_display_name(name)`       (ref portfolio_specs.py:4-19)

The spec dict is the parameter block of the device path: `size` -> k, `rolling_window` -> N,
`risk_aversion` -> gamma, `mcm_scaling`, and `weighting_strategy` selects conjugate / Jeffreys and the
prior weights (`include/tangency_posterior.h: tp_params_t`).
"""
from __future__ import annotations

import itertools

# display name <- substring of the full spec key, tested in this order (the first hit wins; the
# conjugate variants must come before the bare "vw"/"ew" they contain)
_DISPLAY_RULES = (
    ("conjugate_hf_vix_vw", "Conjugate HF-VIX VW"),
    ("conjugate_hf_vix_ew", "Conjugate HF-VIX EW"),
    ("conjugate_hf_epu_vw", "Conjugate HF-EPU VW"),
    ("conjugate_hf_epu_ew", "Conjugate HF-EPU EW"),
    ("jeffreys", "Jeffreys"),
    ("black_litterman", "Black-Litterman"),
    ("shrinkage", "Shrinkage"),
    ("jorion", "Jorion Hyperpar."),
    ("greyserman", "Greyserman Hiera."),
    ("vw", "VW"),
    ("ew", "EW"),
)

_COLORS = {
    "S&P 500": "#FFD700",
    "VW": "#E63946",
    "EW": "#A8DADC",
    "Conjugate HF-VIX VW": "#457B9D",
    "Conjugate HF-VIX EW": "#4D85A6",
    "Conjugate HF-EPU VW": "#FF69B4",
    "Conjugate HF-EPU EW": "#FF7F50",
    "Jeffreys": "#1D3557",
    "Shrinkage": "#F4A261",
    "Jorion Hyperpar.": "#2A9D8F",
    "Black-Litterman": "#9370DB",
    "Greyserman Hiera.": "#9DC209",
}

CONJUGATE_STRATEGIES = ("conjugate_hf_vix_vw", "conjugate_hf_vix_ew", "conjugate_hf_epu_vw", "conjugate_hf_epu_ew")
PASSIVE_STRATEGIES = ("vw", "ew")

# the grid the reference ships (ref portfolio_specs.py:52-62)
DEFAULT_GRID = dict(
    weighting_strategies=["vw", "ew", "conjugate_hf_vix_vw", "conjugate_hf_epu_vw", "jeffreys", "shrinkage",
                          "jorion", "black_litterman", "greyserman"],
    sizes=[50],
    risk_aversions=[5],
    turnover_costs=[15],
    rebalancing_frequencies=["monthly"],
    rolling_windows=[250],
    rolling_window_frequencies=["weekly"],
    mcm_scalings=[1],
)


def get_color_from_display_name(display_name):
    return _COLORS[display_name]


def get_display_name_from_full_name(full_name):
    for needle, shown in _DISPLAY_RULES:
        if needle in full_name:
            return shown
    return None


def spec_key(weighting_strategy, size, risk_aversion, turnover_cost, rebalancing_frequency, rolling_window,
             rolling_window_frequency, mcm_scaling):
    """The reference's spec-name format (ref portfolio_specs.py:77); None prints as NA."""
    na = lambda v: "NA" if v is None else v
    return (f"weighting_strategy_{weighting_strategy}_size_{size}_risk_aversion_{na(risk_aversion)}"
            f"_turnover_cost_{turnover_cost}_rebalancing_frequency_{rebalancing_frequency}"
            f"_rolling_window_{rolling_window}_rolling_window_frequency_{rolling_window_frequency}"
            f"_mcm_scaling_{na(mcm_scaling)}")


_LAST_GRID = {}


def last_grid():
    """The spec dict most recently returned by `create_portfolio_specs` ({} before the first call).
    `portfolio_calculations.backtest_portfolio` looks here for the conjugate siblings of the spec it is given, so
    that the reference's spec loop (src/main.py:48) costs ONE packed upload and ONE device batch for all of them."""
    return _LAST_GRID


def create_portfolio_specs(grid=None):
    """Cartesian product of the grid; passive strategies have no risk aversion, only the conjugate
    strategies have an MCM scaling (ref portfolio_specs.py:66-70)."""
    g = dict(DEFAULT_GRID)
    if grid:
        g.update(grid)
    specs = {}
    for strategy in g["weighting_strategies"]:
        risks = [None] if strategy in PASSIVE_STRATEGIES else g["risk_aversions"]
        scalings = g["mcm_scalings"] if strategy in CONJUGATE_STRATEGIES else [None]
        for size, risk, cost, freq, window, window_freq, scaling in itertools.product(
                g["sizes"], risks, g["turnover_costs"], g["rebalancing_frequencies"], g["rolling_windows"],
                g["rolling_window_frequencies"], scalings):
            key = spec_key(strategy, size, risk, cost, freq, window, window_freq, scaling)
            specs[key] = {
                "weighting_strategy": strategy,
                "size": size,
                "risk_aversion": risk,
                "turnover_cost": cost,
                "rebalancing_frequency": freq,
                "rolling_window": window,
                "rolling_window_frequency": window_freq,
                "mcm_scaling": scaling,
                "display_name": get_display_name_from_full_name(key),
            }
    global _LAST_GRID
    _LAST_GRID = specs
    return specs
