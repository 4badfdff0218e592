"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/micloc_hip.h declares, and the product path fails loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

from haghighatshoarmuir2024_amd import _lib


def header_symbols():
    text = open(os.path.join(ROOT, "include", "micloc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(micloc_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"libmicloc_hip.so does not export {n}"
        assert n in _lib.SYMBOLS, f"{n} is declared in the header but not bound in _lib.SYMBOLS"
    assert sorted(_lib.SYMBOLS) == names
    assert lib.micloc_abi_version() == 1


def test_plain_helpers_without_gpu():
    lib = _lib.load()
    assert lib.micloc_padded_T(4799) == 4800 and lib.micloc_padded_T(8) == 8 and lib.micloc_padded_T(1) == 8
    assert lib.micloc_status_string(0) == b"ok"
    assert b"shape" in lib.micloc_status_string(_lib.MICLOC_ERR_SHAPE)
    assert lib.micloc_workspace_bytes(None, 4, 100) == 0
    assert lib.micloc_rzcc_workspace_bytes(2, 100, 3) > 2 * 100 * 3 * 8
    assert lib.micloc_rzcc_workspace_bytes(0, 100, 3) == 0
    # argument validation happens before any device call
    assert lib.micloc_plan_create(None, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_rzcc_encode_f64(None, 1, 1, 1, 1, 0, None, None, 0, None) == _lib.MICLOC_ERR_INVALID


def test_no_silent_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    from haghighatshoarmuir2024_amd.runtime import Plan
    from haghighatshoarmuir2024_amd.spike_encoder import ZeroCrossingSpikeEncoder

    with pytest.raises(_lib.MiclocError):
        Plan(7, np.ones(16), [1.0], [1.0], 3, True)
    with pytest.raises(_lib.MiclocError):
        ZeroCrossingSpikeEncoder(48_000, 3, True).evolve(np.zeros((10, 2)))
    cfg = _lib.MiclocConfig(device=0, num_mic=7, stht_len=4, stht_kernel=(ctypes.c_double * 4)(0, 1, 0, -1), iir_len=1,
                            iir_b=(ctypes.c_double * 1)(1.0), iir_a=(ctypes.c_double * 1)(1.0), robust_width=1, bipolar=0)
    handle = ctypes.c_void_p()
    assert _lib.load().micloc_plan_create(ctypes.byref(cfg), ctypes.byref(handle)) == _lib.MICLOC_ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "haghighatshoarmuir2024_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "libmicloc_oracle", '#include "micloc_oracle', "oracle/_build"):
                    assert needle not in text, (f, needle)
