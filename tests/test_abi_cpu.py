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


def test_round2_entry_points_validate_before_touching_the_device():
    """The entry points added in round 2 reject bad arguments on the host (no device call is made for them), and their
    size queries are pure functions."""
    lib = _lib.load()
    vp = ctypes.c_void_p
    one = vp(256)  # a non-null, 256-byte aligned dummy: validation must fail before it is ever dereferenced
    assert lib.micloc_awgn_workspace_bytes(0, 10, 7) == 0
    assert lib.micloc_awgn_workspace_bytes(3, 4799, 7) >= (3 * 5 + 3) * 8
    assert lib.micloc_awgn_f64(None, 1, 1, 1, None, None, 0, 0, None, 0, None, 0, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_awgn_f64(one, 1, 10, 7, None, None, 0, 0, None, 0, None, 0, None) == _lib.MICLOC_ERR_INVALID  # neither snr nor sigma
    assert lib.micloc_awgn_f64(one, 1, 10, 7, one, None, 0, 0, None, 0, vp(8), 1 << 20, None) == _lib.MICLOC_ERR_WORKSPACE  # misaligned ws
    assert lib.micloc_uniform_f64(None, 10, 0, 0, None, 0.0, 1.0, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_counter_add_u32(None, 1, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_synth_targets_f64(None, None) == _lib.MICLOC_ERR_INVALID
    args = _lib.MiclocSynthArgs()
    args.time = args.sig = args.slopes = args.x = 256
    args.T, args.B, args.K, args.M, args.fs, args.mode = 100, 1, 1, 7, 48000.0, 2
    assert lib.micloc_synth_targets_f64(ctypes.byref(args), None) == _lib.MICLOC_ERR_INVALID  # unknown mode
    args.mode = 0
    assert lib.micloc_synth_targets_f64(ctypes.byref(args), None) == _lib.MICLOC_ERR_INVALID  # neither delays nor (doa, geometry)
    assert lib.micloc_delay_min_f64(one, 1, 1, 1, one, one, 7, 0.0, one, None) == _lib.MICLOC_ERR_INVALID  # speed must be positive
    assert lib.micloc_design_vectors_f64(one, 1, 130, 1, 1e-8, one, 4, 0, None) == _lib.MICLOC_ERR_SHAPE     # more than 128 channels
    assert lib.micloc_design_vectors_f64(one, 1, 41, 0, 1e-8, one, 4, 0, None) == _lib.MICLOC_ERR_SHAPE      # the wide kernel pairs all columns: even count
    assert lib.micloc_design_vectors_f64(one, 1, 13, 1, 1e-8, one, 4, 0, None) == _lib.MICLOC_ERR_SHAPE      # bipolar needs an even count
    assert lib.micloc_design_vectors_f64(one, 3, 14, 1, 1e-8, one, 4, 2, None) == _lib.MICLOC_ERR_INVALID    # columns past G
    assert lib.micloc_peak_location_i32(one, 1, 449, 1, 14, one, None) == _lib.MICLOC_ERR_INVALID            # even window (utils.py:100)
    assert lib.micloc_peak_location_i32(one, 1, 20, 1, 15, one, None) == _lib.MICLOC_ERR_INVALID             # window > G / 2 (utils.py:105)
    assert lib.micloc_xylo_upload(65, one, 10, one, one, one, one, 1 << 20, None) == _lib.MICLOC_ERR_SHAPE
    assert lib.micloc_xylo_lif_resident_i16(one, 7, 1, 10, 28, 10, 0, 31, None, one, one, 1 << 20, None) == _lib.MICLOC_ERR_SHAPE  # Cin != 2 x ternary
    assert lib.micloc_stream_state_bytes(None, 4) == 0
    assert lib.micloc_stream_overflow(None, None, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_lif_beamform_workspace_bytes(None, 1, 10) == 0


def test_round3_entry_points_validate_before_touching_the_device():
    """Round-3 entry points: bad arguments are rejected on the host, size queries are pure functions."""
    lib = _lib.load()
    vp = ctypes.c_void_p
    one = vp(256)
    # Gram matrix of a planar signal
    assert lib.micloc_planar_gram_workspace_bytes(0, 100, 14, 0) == 0 and lib.micloc_planar_gram_workspace_bytes(1, 100, 14, 100) == 0
    assert lib.micloc_planar_gram_workspace_bytes(2, 4799, 14, 480) >= 2 * 3 * 256 * 8  # 2 trials x 3 chunks x one 16 x 16 tile
    assert lib.micloc_planar_gram_workspace_bytes(1, 2048, 128, 0) == 36 * 256 * 8       # 8 channel tiles: 36 tile pairs
    assert lib.micloc_planar_gram_f64(None, 1, 14, 100, 104, 0, 1, one, one, 1 << 20, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_planar_gram_f64(one, 1, 129, 100, 104, 0, 1, one, one, 1 << 20, None) == _lib.MICLOC_ERR_INVALID  # more than 128 channels
    assert lib.micloc_planar_gram_f64(one, 1, 14, 100, 96, 0, 1, one, one, 1 << 20, None) == _lib.MICLOC_ERR_SHAPE      # row stride < T
    assert lib.micloc_planar_gram_f64(one, 1, 14, 100, 104, 0, 1, one, one, 16, None) == _lib.MICLOC_ERR_WORKSPACE
    # fused synthesis + noise
    assert lib.micloc_synth_awgn_workspace_bytes(0, 10, 7, 1) == 0
    assert lib.micloc_synth_awgn_workspace_bytes(3, 4799, 7, 2) >= lib.micloc_awgn_workspace_bytes(3, 4799, 7) + 3 * 2 * 7 * 8
    assert lib.micloc_synth_awgn_f64(None, one, 0, 0, None, 0, one, 1 << 20, None) == _lib.MICLOC_ERR_INVALID
    args = _lib.MiclocSynthArgs()
    args.time = args.sig = args.slopes = args.x = args.doa = args.r_vec = args.theta_vec = 256
    args.T, args.B, args.K, args.M, args.fs, args.mode, args.speed = 100, 1, 1, 7, 48000.0, 0, 340.0
    assert lib.micloc_synth_awgn_f64(ctypes.byref(args), None, 0, 0, None, 0, one, 1 << 20, None) == _lib.MICLOC_ERR_INVALID  # no SNR
    assert lib.micloc_synth_awgn_f64(ctypes.byref(args), one, 0, 0, None, 0, one, 16, None) == _lib.MICLOC_ERR_WORKSPACE
    assert lib.micloc_synth_awgn_f64(ctypes.byref(args), one, 0, 0, None, 0xFFFFFFFF, one, 1 << 20, None) == _lib.MICLOC_ERR_INVALID  # trial ids
    # the reserved trial word of the uniform generator / the pair index is one counter word
    assert lib.micloc_awgn_f64(one, 2, 10, 7, one, None, 0, 0, None, 0xFFFFFFFE, one, 1 << 20, None) == _lib.MICLOC_ERR_INVALID
    # streaming localisation
    assert lib.micloc_stream_localize_state_bytes(None, 4) == 0 and lib.micloc_stream_localize_workspace_bytes(None, 4, 512) == 0
    assert lib.micloc_stream_chunk_frames(None) == _lib.MICLOC_ERR_NOT_SET
    # the clocked tile calls (no absolute time by value)
    assert lib.micloc_stream_reset(None, 1, one, 1 << 20, one, 1 << 20, one, 512, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_stream_begin_tile(None, one, one, vp(512), 1, 16, 512, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_stream_wrap_rows_f64(None, one, one, 1, 512, 480, 16, None, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_stream_encode_tile_f64(None, one, 1, 16, 16, 0, one, 512, one, 1 << 20, one, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_stream_localize_tile_f64(None, one, one, 1 << 20, one, 1, 512, 0, None, None, one, 1 << 20, None) == _lib.MICLOC_ERR_INVALID
    # the sweep's Xylo call and the Demo's channel bookkeeping
    assert lib.micloc_xylo_sweep_scratch_bytes(0) == 0 and lib.micloc_xylo_sweep_scratch_bytes(4) > 0
    assert lib.micloc_xylo_lif_sweep_i16(None, 7, 1, 10, 14, 10, 31, one, one, 1 << 20, one, 1 << 20, 0, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_xylo_lif_sweep_i16(one, 7, 1, 10, 28, 10, 31, one, one, 1 << 20, one, 1 << 20, 0, None) == _lib.MICLOC_ERR_SHAPE  # Cin != 2 x ternary
    assert lib.micloc_xylo_lif_sweep_i16(one, 7, 1, 10, 14, 10, 31, one, one, 1 << 20, one, 1 << 20, 9, None) == _lib.MICLOC_ERR_INVALID  # workers per CU
    assert lib.micloc_xylo_sweep_status(None, None, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_pack_events_u8(one, 1, 10, 14, 2, 2, 2, one, None) == _lib.MICLOC_ERR_INVALID  # band >= bands
    assert lib.micloc_pack_events_u8(one, 1, 10, 14, 0, 1, 3, one, None) == _lib.MICLOC_ERR_INVALID  # unknown mode
    assert lib.micloc_rate_from_counts_f64(None, 1, 10, 1, 100, 48000.0, one, None) == _lib.MICLOC_ERR_INVALID
    assert lib.micloc_stream_localize_status(None, None, None) == _lib.MICLOC_ERR_INVALID


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


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/micloc_hip.h is the drop-in boundary: it must be usable from plain C (C99, -pedantic), and a C program must be able to
    link against libmicloc_hip.so and call the entry points that need no GPU."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this host")
    src = tmp_path / "abi_c.c"
    src.write_text(
        '#include "micloc_hip.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        '    printf("%d %s %d %zu\\n", micloc_abi_version(), micloc_status_string(MICLOC_ERR_SHAPE), micloc_padded_T(4799),\n'
        "           micloc_rzcc_workspace_bytes(2, 100, 14));\n"
        "    return micloc_abi_version() == MICLOC_ABI_VERSION ? 0 : 1;\n}\n")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-fsyntax-only", str(src)])
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "abi_c"
    # (the HIP runtime the library needs is the one PyTorch-ROCm ships or /opt/rocm's: let the dynamic loader find either)
    import torch

    rpaths = [libdir, os.path.join(os.path.dirname(torch.__file__), "lib"), "/opt/rocm/lib"]
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lmicloc_hip", "-Wl,--allow-shlib-undefined"]
                       + [f"-Wl,-rpath,{p}" for p in rpaths], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    fields = out.stdout.decode().split()
    assert fields[0] == "1" and fields[-2] == "4800" and int(fields[-1]) > 0
