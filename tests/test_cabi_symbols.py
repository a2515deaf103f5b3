"""The C-ABI library loads on a CPU-only box and exports every symbol include/tangency_posterior.h
declares; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import REPO, have_gpu


def declared_functions():
    text = open(os.path.join(REPO, "include", "tangency_posterior.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tp_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_lists():
    from incorporating_different_sources_amd import _native
    assert declared_functions() == sorted(_native.EXPORTS)


def test_library_exports_every_declared_symbol():
    from incorporating_different_sources_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"libtangency.so does not export {name}"
    lib.tp_version.restype = ctypes.c_char_p
    assert b"tangency-posterior" in lib.tp_version()
    lib.tp_max_assets.restype = ctypes.c_int
    assert lib.tp_max_assets() >= 100


def test_struct_layout_matches_header():
    from incorporating_different_sources_amd import _native
    # tp_params_t: 6 x int32 + double; tp_inputs_t: 11 pointers + 2 x int64 + 2 x int32
    assert ctypes.sizeof(_native.tp_params_t) == 32
    assert ctypes.sizeof(_native.tp_inputs_t) == 8 + 8 + 4 + 4 + 5 * 8 + 8 + 8 + 5 * 8
    assert _native.tp_inputs_t.hf_ld.offset == 20 and _native.tp_inputs_t.start.offset == 24


@pytest.mark.skipif(have_gpu(), reason="CPU-only behaviour")
def test_no_gpu_fails_loudly():
    from incorporating_different_sources_amd import _native
    with pytest.raises(_native.TangencyError) as e:
        _native.Device(0)
    assert e.value.code == _native.TP_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, load or link it."""
    pkg = os.path.join(REPO, "incorporating_different_sources_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                continue
            for line in open(os.path.join(root, f)).read().splitlines():
                low = line.lower()
                if "oracle" in low and any(t in low for t in ("import", "#include", "cdll", "-l", "dlopen")):
                    raise AssertionError(f"{f}: product code references the oracle: {line.strip()}")
