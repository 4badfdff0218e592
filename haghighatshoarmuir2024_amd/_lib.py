"""ctypes binding of libmicloc_hip.so (the C-ABI declared in include/micloc_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmicloc_hip.so")

MICLOC_OK = 0
MICLOC_ERR_INVALID = -1
MICLOC_ERR_SHAPE = -2
MICLOC_ERR_WORKSPACE = -3
MICLOC_ERR_NOT_SET = -4
MICLOC_ERR_HIP = -5
MICLOC_ERR_NO_DEVICE = -6
MICLOC_MAX_IIR = 9

c_double_p = ctypes.POINTER(ctypes.c_double)
c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t


class MiclocConfig(ctypes.Structure):
    _fields_ = [
        ("device", c_int),
        ("num_mic", c_int),
        ("stht_len", c_int),
        ("stht_kernel", c_double_p),
        ("iir_len", c_int),
        ("iir_b", c_double_p),
        ("iir_a", c_double_p),
        ("robust_width", c_int),
        ("bipolar", c_int),
    ]


class MiclocSynthArgs(ctypes.Structure):
    _fields_ = [
        ("time", c_void_p), ("sig", c_void_p), ("slopes", c_void_p),
        ("T", c_int), ("B", c_int), ("K", c_int), ("M", c_int),
        ("delays", c_void_p),
        ("doa", c_void_p),
        ("moving", c_int),
        ("r_vec", c_void_p), ("theta_vec", c_void_p),
        ("speed", ctypes.c_double),
        ("shift", c_void_p),
        ("gain", c_void_p),
        ("mode", c_int),
        ("fs", ctypes.c_double),
        ("x", c_void_p),
    ]


# every symbol include/micloc_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "micloc_plan_create": (c_int, [ctypes.POINTER(MiclocConfig), ctypes.POINTER(c_void_p)]),
    "micloc_plan_destroy": (None, [c_void_p]),
    "micloc_plan_set_neuron_kernel": (c_int, [c_void_p, c_double_p, c_int]),
    "micloc_plan_set_bf_mat": (c_int, [c_void_p, c_double_p, c_int, c_int]),
    "micloc_plan_set_bf_mat_c128": (c_int, [c_void_p, c_double_p, c_double_p, c_int, c_int]),
    "micloc_plan_generation": (c_int, [c_void_p]),
    "micloc_plan_set_encoder_chunk": (c_int, [c_void_p, c_int]),
    "micloc_plan_encoder_chunks": (c_int, [c_void_p, c_int, c_int]),
    "micloc_stream_create_cu_range": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "micloc_stream_destroy": (c_int, [c_void_p]),
    "micloc_padded_T": (c_int, [c_int]),
    "micloc_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "micloc_stht_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "micloc_bandpass_rzcc_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_lif_beamform_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_beamform_c128_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_snn_pipeline_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_snn_pipeline_stages_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int]),
    "micloc_beamformer_pipeline_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_rzcc_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "micloc_rzcc_encode_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_rzcc_workspace_bytes_ex": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "micloc_rzcc_encode_ex_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_lfilter_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "micloc_lfilter_f64": (c_int, [c_double_p, c_double_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_lif_beamform_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_lif_covariance_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_planar_gram_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "micloc_planar_gram_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_snn_pipeline_cov_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_synth_delay_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, ctypes.c_double, c_void_p, c_void_p]),
    "micloc_synth_targets_f64": (c_int, [c_void_p, c_void_p]),
    "micloc_delay_min_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, ctypes.c_double, c_void_p, c_void_p]),
    "micloc_uniform_f64": (c_int, [c_void_p, c_size_t, ctypes.c_uint64, ctypes.c_uint32, c_void_p, ctypes.c_double, ctypes.c_double, c_void_p]),
    "micloc_counter_add_u32": (c_int, [c_void_p, ctypes.c_uint32, c_void_p]),
    "micloc_awgn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "micloc_awgn_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.c_uint64, ctypes.c_uint32, c_void_p, ctypes.c_uint32,
                                c_void_p, c_size_t, c_void_p]),
    "micloc_synth_awgn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "micloc_synth_awgn_f64": (c_int, [c_void_p, c_void_p, ctypes.c_uint64, ctypes.c_uint32, c_void_p, ctypes.c_uint32, c_void_p, c_size_t, c_void_p]),
    "micloc_doa_error_f64": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "micloc_xylo_workspace_bytes": (c_size_t, [c_int, c_int]),
    "micloc_xylo_lif_i16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_xylo_upload": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_xylo_lif_resident_i16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "micloc_xylo_sweep_scratch_bytes": (c_size_t, [c_int]),
    "micloc_xylo_sweep_status": (c_int, [c_void_p, ctypes.POINTER(c_int), c_void_p]),
    "micloc_pack_events_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "micloc_rate_from_counts_f64": (c_int, [c_void_p, c_int, c_int, c_int, c_int, ctypes.c_double, c_void_p, c_void_p]),
    "micloc_xylo_lif_sweep_i16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p]),
    "micloc_lif_beamform_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "micloc_stream_state_bytes": (c_size_t, [c_void_p, c_int]),
    "micloc_stream_overflow": (c_int, [c_void_p, ctypes.POINTER(c_int), c_void_p]),
    "micloc_stream_localize_state_bytes": (c_size_t, [c_void_p, c_int]),
    "micloc_stream_chunk_frames": (c_int, [c_void_p]),
    "micloc_stream_localize_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "micloc_stream_reset": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    "micloc_stream_begin_tile": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "micloc_stream_wrap_rows_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "micloc_stream_encode_tile_f64": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "micloc_stream_localize_tile_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                c_size_t, c_void_p]),
    "micloc_stream_localize_status": (c_int, [c_void_p, ctypes.POINTER(c_int), c_void_p]),
    "micloc_design_vectors_f64": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_double, c_void_p, c_int, c_int, c_void_p]),
    "micloc_peak_location_i32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "micloc_envelope_track_f64": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p, c_void_p, c_void_p]),
    "micloc_envelope_track_any": (c_int, [c_void_p, c_int, c_int, c_int, c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p, c_void_p, c_void_p]),
    "micloc_abi_version": (c_int, []),
    "micloc_status_string": (ctypes.c_char_p, [c_int]),
    "micloc_last_hip_error": (c_int, []),
}

_lib = None


class MiclocError(RuntimeError):
    pass


def load():
    """Load libmicloc_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiclocError(
                f"{LIB_PATH} not found: build it with `make -C haghighatshoarmuir2024_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback"
            )
        # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64; it must be in the process BEFORE this library
        # is loaded so that both resolve to ONE HIP runtime (same SONAME).  Loaded the other way round, the two
        # runtimes each try to own the device and hipGetDeviceCount() fails in the second one.
        import torch  # noqa: F401

        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status, what=""):
    if status == MICLOC_OK:
        return
    lib = load()
    msg = lib.micloc_status_string(status).decode()
    if status == MICLOC_ERR_HIP:
        msg += f" [hipError_t {lib.micloc_last_hip_error()}]"
    if status == MICLOC_ERR_SHAPE:
        raise ValueError(f"micloc {what}: {msg}")
    raise MiclocError(f"micloc {what}: {msg} (status {status})")
