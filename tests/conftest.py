import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_usable():
    """A HIP device is visible.  (With a device but without libmicloc_hip.so the gpu tests RUN and fail loudly:
    the product path has no CPU fallback and a GPU box must never pass on a skip.)"""
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without an MI355X: gpu-marked tests are skipped (not failed); `-m gpu` on the GPU box
    runs them.  Nothing is skipped when a device is present, so a missing library still fails loudly there."""
    if any(item.get_closest_marker("gpu") for item in items) and not _gpu_usable():
        skip = pytest.mark.skip(reason="needs an MI355X (no HIP device visible)")
        for item in items:
            if item.get_closest_marker("gpu"):
                item.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def cfg2():
    """Parameters of BASELINE config 2 (paper_plots/target_snn_localization.py:319-342) from golden data."""
    k = golden("kat_init.npz")
    bf = golden("bf_mat_chirp449_bipolar.npz")
    return dict(
        fs=48_000,
        kernel=k["kernel_48k"],
        b=k["b_48k"],
        a=k["a_48k"],
        robust_width=int(k["robust_width_48k"]),
        nir=k["nir_48k"],
        bf_mat=bf["bf_mat"],
        doa_list=bf["doa_list"],
        r_vec=k["ccirc_r"],
        theta_vec=k["ccirc_theta"],
    )


def campaign_seeds(name, default):
    """Seeds of a random parity campaign as pytest params whose ids carry the seed COUNT (`seed17-of-600`): the driver's `pytest -m gpu`
    log says how many configurations ran.  `MICLOC_SEEDS_<NAME>` overrides one campaign, `MICLOC_RANDOM_SEEDS` all of them (the
    builder's long runs: DESIGN.md section 2)."""
    n = int(os.environ.get(f"MICLOC_SEEDS_{name.upper()}", os.environ.get("MICLOC_RANDOM_SEEDS", str(default))))
    return [pytest.param(s, id=f"seed{s}-of-{n}") for s in range(n)]
