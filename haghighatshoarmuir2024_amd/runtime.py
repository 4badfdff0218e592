"""Device runtime under the micloc class surface: one `Plan` per (beamformer instance, device).

PyTorch-ROCm is used only as the device-memory / stream provider: tensors are allocated with torch,
their raw pointers are handed to the C-ABI, and kernels run on torch's current HIP stream.
"""
import ctypes

import numpy as np

from . import _lib


def _torch():
    import torch

    return torch


def require_gpu(device=None):
    torch = _torch()
    if not torch.cuda.is_available():
        raise _lib.MiclocError("no HIP device visible: the micloc hot path runs on MI355X only (no CPU fallback)")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.MiclocError(f"device must be a HIP ('cuda') device, got {device}")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def to_host(t):
    """Device tensor -> NumPy array owned by the caller, through page-locked memory: the array is backed by a pinned block of
    torch's caching host allocator (one DMA, no staging copy; the block goes back to the cache when the caller drops the array --
    the script's loop, paper_plots/target_snn_localization.py:455-464, recycles it every trial).  Synchronises the current stream."""
    torch = _torch()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy()


def _dptr(a):
    return a.ctypes.data_as(_lib.c_double_p)


def _stream(device):
    return ctypes.c_void_p(_torch().cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def pad_ba(b, a):
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    n = max(len(b), len(a))
    if n > _lib.MICLOC_MAX_IIR:
        raise ValueError(f"IIR filters with more than {_lib.MICLOC_MAX_IIR} coefficients are not supported")
    bb = np.zeros(n)
    aa = np.zeros(n)
    bb[: len(b)] = b
    aa[: len(a)] = a
    return np.ascontiguousarray(bb), np.ascontiguousarray(aa), n


class Workspace:
    """Grow-only scratch buffer on one device (256-byte aligned by the torch allocator)."""

    def __init__(self, device):
        self.device = device
        self.buf = None
        self.generation = 0  # bumped on every (re)allocation: graphs captured against the old buffer are stale

    def get(self, nbytes):
        torch = _torch()
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = None
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self.generation += 1
        return self.buf


class Plan:
    """Owns a micloc_plan (filter taps, neuron kernel, beamforming matrix on the device) + workspace."""

    def __init__(self, num_mic, stht_kernel, b, a, robust_width, bipolar, device=None):
        self.lib = _lib.load()
        self.device = require_gpu(device)
        self.num_mic = int(num_mic)
        self.C = 2 * self.num_mic
        ker = np.ascontiguousarray(stht_kernel, dtype=np.float64)
        bb, aa, n = pad_ba(b, a)
        cfg = _lib.MiclocConfig(
            device=self.device.index,
            num_mic=self.num_mic,
            stht_len=len(ker),
            stht_kernel=_dptr(ker),
            iir_len=n,
            iir_b=_dptr(bb),
            iir_a=_dptr(aa),
            robust_width=int(robust_width),
            bipolar=int(bool(bipolar)),
        )
        handle = ctypes.c_void_p()
        _lib.check(self.lib.micloc_plan_create(ctypes.byref(cfg), ctypes.byref(handle)), "plan_create")
        self.handle = handle
        self.ws = Workspace(self.device)
        self._nir_key = None
        self._w_key = None
        self.G = None
        self.w_complex = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.micloc_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- plan state ------------------------------------------------------------------------------
    def set_neuron_kernel(self, nir):
        nir = np.ascontiguousarray(nir, dtype=np.float64)
        key = nir.tobytes()
        if key != self._nir_key:
            _lib.check(self.lib.micloc_plan_set_neuron_kernel(self.handle, _dptr(nir), len(nir)), "set_neuron_kernel")
            self._nir_key = key

    def set_bf_mat(self, W):
        W = np.asarray(W)
        if np.iscomplexobj(W):
            key = (True, W.shape, W.tobytes())
            if key != self._w_key:
                Wre = np.ascontiguousarray(W.real, dtype=np.float64)
                Wim = np.ascontiguousarray(W.imag, dtype=np.float64)
                _lib.check(self.lib.micloc_plan_set_bf_mat_c128(self.handle, _dptr(Wre), _dptr(Wim), W.shape[0], W.shape[1]), "set_bf_mat_c128")
                self._w_key = key
            self.w_complex = True
        else:
            W = np.ascontiguousarray(W, dtype=np.float64)
            key = (False, W.shape, W.tobytes())
            if key != self._w_key:
                _lib.check(self.lib.micloc_plan_set_bf_mat(self.handle, _dptr(W), W.shape[0], W.shape[1]), "set_bf_mat")
                self._w_key = key
            self.w_complex = False
        self.G = W.shape[1]

    def set_encoder_chunk(self, chunk_frames):
        """Time chunking of the band-pass / RZCC stage (micloc_plan_set_encoder_chunk): 0 automatic, < 0 off,
        > 0 owned frames per chunk.  Bit-identical results for every setting."""
        _lib.check(self.lib.micloc_plan_set_encoder_chunk(self.handle, int(chunk_frames)), "set_encoder_chunk")

    def encoder_chunks(self, B, T):
        return self.lib.micloc_plan_encoder_chunks(self.handle, int(B), int(T))

    def workspace(self, B, T):
        n = self.lib.micloc_workspace_bytes(self.handle, B, T)
        return self.ws.get(n), n

    @property
    def generation(self):
        """Changes whenever a device pointer a captured hipGraph may hold was re-allocated: the plan's tables
        (micloc_plan_generation) or this plan's workspace.  StreamPipeline.capture records it, replay() checks it."""
        return (self.lib.micloc_plan_generation(self.handle), self.ws.generation)

    def padded_T(self, T):
        return self.lib.micloc_padded_T(T)

    # ---- helpers -----------------------------------------------------------------------------------
    def to_device(self, x):
        """numpy / torch [B, T, M] float64 -> contiguous device tensor."""
        torch = _torch()
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
        if x.dtype != torch.float64:
            x = x.to(torch.float64)
        if x.device != self.device:
            x = x.to(self.device)
        return x.contiguous()

    # ---- pipelines -----------------------------------------------------------------------------------
    def snn_pipeline(self, x, want_spikes=False, want_y=False, want_power=True, stages=7, out=None):
        """x: device tensor [B, T, M]. Returns dict of device tensors (spikes int8, y, power, argmax).
        `stages` (MICLOC_STAGE_* bits: 1 STHT, 2 band-pass + RZCC, 4 LIF + beamforming + power; 8 / 16: only the serial scan of a
        chunked band-pass + RZCC stage / the stage without it) launches a part of the pipeline; the parts of one batch share this
        plan's workspace and, via `out`, the output tensors."""
        torch = _torch()
        B, T, M = x.shape
        if M != self.num_mic:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {self.num_mic}!")
        G = self.G
        if out is None:
            out = dict(spikes=torch.empty((B, T, self.C), dtype=torch.int8, device=self.device) if want_spikes else None,
                       y=torch.empty((B, T, G), dtype=torch.float64, device=self.device) if want_y else None,
                       power=torch.empty((B, G), dtype=torch.float64, device=self.device) if want_power else None,
                       argmax=torch.empty((B,), dtype=torch.int32, device=self.device) if want_power else None)
        ws, nbytes = self.workspace(B, T)
        _lib.check(
            self.lib.micloc_snn_pipeline_stages_f64(self.handle, _ptr(x), B, T, _ptr(out["spikes"]), _ptr(out["y"]), _ptr(out["power"]),
                                                     _ptr(out["argmax"]), _ptr(ws), nbytes, _stream(self.device), int(stages)),
            "snn_pipeline",
        )
        return out

    def snn_pipeline_f32bf(self, x, want_spikes=False):
        """VARIANT: fp64 up to the spikes (STHT, band-pass, RZCC), then LIF + beamforming + power on the fp32 MFMA units
        (micloc_lif_beamform_f32).  Returns spikes / power / argmax like snn_pipeline; power agrees to ~1e-6 relative."""
        torch = _torch()
        B, T, M = x.shape
        if M != self.num_mic:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {self.num_mic}!")
        ws, nbytes = self.workspace(B, T)
        Ts = self.padded_T(T)
        key = (B, T)
        if getattr(self, "_f32_key", None) != key:  # stage buffers, allocated once per shape
            self._f32_h = torch.empty((B, self.C, Ts), dtype=torch.float64, device=self.device)
            self._f32_spk = torch.empty((B, T, self.C), dtype=torch.int8, device=self.device)
            self._f32_key = key
        h, spikes = self._f32_h, self._f32_spk
        st = _stream(self.device)
        _lib.check(self.lib.micloc_stht_f64(self.handle, _ptr(x), B, T, _ptr(h), Ts, st), "stht")
        _lib.check(self.lib.micloc_bandpass_rzcc_f64(self.handle, _ptr(h), B, T, Ts, None, _ptr(spikes), _ptr(ws), nbytes, st), "bandpass_rzcc")
        power = torch.empty((B, self.G), dtype=torch.float64, device=self.device)
        argmax = torch.empty((B,), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.micloc_lif_beamform_f32(self.handle, _ptr(spikes), B, T, _ptr(power), _ptr(argmax), _ptr(ws), nbytes, st),
                   "lif_beamform_f32")
        return dict(spikes=spikes.clone() if want_spikes else None, y=None, power=power, argmax=argmax)

    def snn_pipeline_cov(self, x, t_start=0, want_spikes=False, want_cov=False, want_power=True, cov_out=None):
        """Covariance-form tail: power = w^T (V^T V / T') w, optionally the membrane covariance itself
        (frames t >= t_start).  Algebraically identical to snn_pipeline's power; supports up to 64 channels."""
        torch = _torch()
        B, T, M = x.shape
        if M != self.num_mic:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {self.num_mic}!")
        spikes = torch.empty((B, T, self.C), dtype=torch.int8, device=self.device) if want_spikes else None
        cov = torch.empty((B, self.C, self.C), dtype=torch.float64, device=self.device) if want_cov and cov_out is None else cov_out
        if cov is not None and (tuple(cov.shape) != (B, self.C, self.C) or cov.dtype != torch.float64 or not cov.is_contiguous()):
            raise ValueError("cov_out must be a contiguous float64 [B, 2M, 2M] device tensor")
        power = torch.empty((B, self.G), dtype=torch.float64, device=self.device) if want_power else None
        argmax = torch.empty((B,), dtype=torch.int32, device=self.device) if want_power else None
        ws, nbytes = self.workspace(B, T)
        _lib.check(
            self.lib.micloc_snn_pipeline_cov_f64(self.handle, _ptr(x), B, T, int(t_start), _ptr(spikes), _ptr(cov), _ptr(power), _ptr(argmax),
                                                  _ptr(ws), nbytes, _stream(self.device)),
            "snn_pipeline_cov",
        )
        return dict(spikes=spikes, cov=cov, power=power, argmax=argmax)

    def beamformer_pipeline(self, x, want_y=False, want_power=True):
        torch = _torch()
        B, T, M = x.shape
        if M != self.num_mic:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {self.num_mic}!")
        G = self.G
        y = torch.empty((B, T, G), dtype=torch.complex128, device=self.device) if want_y else None
        power = torch.empty((B, G), dtype=torch.float64, device=self.device) if want_power else None
        argmax = torch.empty((B,), dtype=torch.int32, device=self.device) if want_power else None
        ws, nbytes = self.workspace(B, T)
        _lib.check(
            self.lib.micloc_beamformer_pipeline_f64(self.handle, _ptr(x), B, T, _ptr(y), _ptr(power), _ptr(argmax), _ptr(ws), nbytes,
                                                     _stream(self.device)),
            "beamformer_pipeline",
        )
        return dict(y=y, power=power, argmax=argmax)

    def beamform_c128(self, pre, T, want_y=False, want_power=True, out=None):
        """The contraction stage of the complex Beamformer alone (micloc_beamform_c128_f64): planar band-passed rows
        pre [B, 2M, Ts] -> y [B, T, G] complex128 and / or power [B, G], argmax [B].  `out` reuses a previous result dict."""
        torch = _torch()
        B, C, Ts = pre.shape
        G = self.G
        if out is None:
            out = dict(y=torch.empty((B, T, G), dtype=torch.complex128, device=self.device) if want_y else None,
                       power=torch.empty((B, G), dtype=torch.float64, device=self.device) if want_power else None,
                       argmax=torch.empty((B,), dtype=torch.int32, device=self.device) if want_power else None)
        nbytes = self.lib.micloc_lif_beamform_workspace_bytes(self.handle, B, T)
        ws = self.ws.get(nbytes)
        _lib.check(self.lib.micloc_beamform_c128_f64(self.handle, _ptr(pre), B, T, Ts, _ptr(out["y"]), _ptr(out["power"]), _ptr(out["argmax"]),
                                                     _ptr(ws), nbytes, _stream(self.device)), "beamform_c128")
        return out

    # ---- single stages (used by tests and by Demo.spike_encoding-style callers) --------------------------
    def stht(self, x):
        torch = _torch()
        B, T, M = x.shape
        Ts = self.padded_T(T)
        h = torch.empty((B, self.C, Ts), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.micloc_stht_f64(self.handle, _ptr(x), B, T, _ptr(h), Ts, _stream(self.device)), "stht")
        return h

    def bandpass_rzcc(self, h, T, want_pre=True, want_spikes=True):
        torch = _torch()
        B, C, Ts = h.shape
        pre = torch.empty_like(h) if want_pre else None
        spikes = torch.empty((B, T, C), dtype=torch.int8, device=self.device) if want_spikes else None
        ws, nbytes = self.workspace(B, T)
        _lib.check(self.lib.micloc_bandpass_rzcc_f64(self.handle, _ptr(h), B, T, Ts, _ptr(pre), _ptr(spikes), _ptr(ws), nbytes,
                                                     _stream(self.device)), "bandpass_rzcc")
        return pre, spikes

    def lif_beamform(self, spikes, want_y=False, want_power=True):
        torch = _torch()
        B, T, C = spikes.shape
        G = self.G
        y = torch.empty((B, T, G), dtype=torch.float64, device=self.device) if want_y else None
        power = torch.empty((B, G), dtype=torch.float64, device=self.device) if want_power else None
        argmax = torch.empty((B,), dtype=torch.int32, device=self.device) if want_power else None
        ws, nbytes = self.workspace(B, T)
        _lib.check(self.lib.micloc_lif_beamform_f64(self.handle, _ptr(spikes), B, T, _ptr(y), _ptr(power), _ptr(argmax), _ptr(ws), nbytes,
                                                    _stream(self.device)), "lif_beamform")
        return dict(y=y, power=power, argmax=argmax)


# ---- plan-less operators -------------------------------------------------------------------------------
_op_ws = {}


def _op_workspace(device, nbytes):
    ws = _op_ws.get(device)
    if ws is None:
        ws = _op_ws[device] = Workspace(device)
    return ws.get(nbytes)


def rzcc_encode(sig, robust_width, bipolar, device=None, chunk_frames=0):
    """ZeroCrossingSpikeEncoder.evolve on the device. sig: numpy/torch [T, C] or [B, T, C] -> int8 device tensor.
    chunk_frames: time chunking (0 automatic, < 0 off, > 0 frames per chunk); the result does not depend on it."""
    torch = _torch()
    lib = _lib.load()
    device = require_gpu(device)
    if isinstance(sig, np.ndarray):
        sig = torch.from_numpy(np.ascontiguousarray(sig, dtype=np.float64))
    sig = sig.to(device=device, dtype=torch.float64).contiguous()
    squeeze = sig.dim() == 2
    if squeeze:
        sig = sig.unsqueeze(0)
    B, T, C = sig.shape
    spikes = torch.zeros((B, T, C), dtype=torch.int8, device=device)
    if B * T * C > 0:
        nbytes = lib.micloc_rzcc_workspace_bytes_ex(B, T, C, int(robust_width), int(chunk_frames))
        ws = _op_workspace(device, nbytes)
        _lib.check(lib.micloc_rzcc_encode_ex_f64(_ptr(sig), B, T, C, int(robust_width), int(bool(bipolar)), int(chunk_frames), _ptr(spikes),
                                                 _ptr(ws), nbytes, _stream(device)), "rzcc_encode")
    return spikes[0] if squeeze else spikes


def lfilter(b, a, x, device=None, out=None):
    """scipy.signal.lfilter(b, a, x, axis=0) for real x [T, C] or [B, T, C] on the device (`out`: a contiguous float64 device tensor of
    x's shape to write into -- e.g. one band's slice of a filterbank's [F, T, M] result)."""
    torch = _torch()
    lib = _lib.load()
    device = require_gpu(device)
    bb, aa, n = pad_ba(b, a)
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
    x = x.to(device=device, dtype=torch.float64).contiguous()
    squeeze = x.dim() == 2
    if squeeze:
        x = x.unsqueeze(0)
    B, T, C = x.shape
    if out is None:
        y = torch.empty_like(x)
    else:
        y = out.unsqueeze(0) if squeeze else out
        if tuple(y.shape) != (B, T, C) or y.dtype != torch.float64 or not y.is_contiguous() or y.device != x.device:
            raise ValueError("out must be a contiguous float64 device tensor of the input's shape")
    if B * T * C > 0:
        nbytes = lib.micloc_lfilter_workspace_bytes(B, T, C)
        ws = _op_workspace(device, nbytes)
        _lib.check(lib.micloc_lfilter_f64(_dptr(bb), _dptr(aa), n, _ptr(x), B, T, C, _ptr(y), _ptr(ws), nbytes, _stream(device)), "lfilter")
    return y[0] if squeeze else y


def synth_delay(time_in, sig_in, delays, fs, device=None):
    """Device synthesis of noise-free array signals: time_in/sig_in numpy [T] (already on the fs grid), delays numpy
    [B, M] (min-shifted) -> device tensor [B, T, M], bit-exact with np.interp (see csrc/synth.hip)."""
    torch = _torch()
    lib = _lib.load()
    device = require_gpu(device)
    time_in = np.ascontiguousarray(time_in, dtype=np.float64)
    sig_in = np.ascontiguousarray(sig_in, dtype=np.float64)
    slopes = np.diff(sig_in) / np.diff(time_in)
    delays = np.ascontiguousarray(delays, dtype=np.float64)
    B, M = delays.shape
    T = len(time_in)
    d_time = torch.from_numpy(time_in).to(device)
    d_sig = torch.from_numpy(sig_in).to(device)
    d_slopes = torch.from_numpy(np.ascontiguousarray(slopes)).to(device)
    d_delays = torch.from_numpy(delays).to(device)
    x = torch.empty((B, T, M), dtype=torch.float64, device=device)
    _lib.check(lib.micloc_synth_delay_f64(_ptr(d_time), _ptr(d_sig), _ptr(d_slopes), T, _ptr(d_delays), B, M, float(fs), _ptr(x),
                                          _stream(device)), "synth_delay")
    return x


class Template:
    """A source signal on the fs grid, resident on the device (time, values, np.interp's slope table)."""

    def __init__(self, time_in, sig_in, fs, device=None):
        torch = _torch()
        self.device = require_gpu(device)
        time_in = np.ascontiguousarray(time_in, dtype=np.float64)
        sig_in = np.ascontiguousarray(sig_in, dtype=np.float64)
        if time_in.ndim != 1 or time_in.shape != sig_in.shape or len(time_in) < 2:
            raise ValueError("template needs a time vector and a signal of the same length >= 2")
        self.T = len(time_in)
        self.fs = float(fs)
        self.time_host = time_in
        self.time = torch.from_numpy(time_in).to(self.device)
        self.sig = torch.from_numpy(sig_in).to(self.device)
        self.slopes = torch.from_numpy(np.ascontiguousarray(np.diff(sig_in) / np.diff(time_in))).to(self.device)


class Geometry:
    """Microphone polar coordinates on the device (for delays computed in the synthesis kernel)."""

    def __init__(self, geometry, device=None):
        torch = _torch()
        self.device = require_gpu(device)
        self.M = len(geometry.r_vec)
        self.r_vec = torch.from_numpy(np.ascontiguousarray(geometry.r_vec, dtype=np.float64)).to(self.device)
        self.theta_vec = torch.from_numpy(np.ascontiguousarray(geometry.theta_vec, dtype=np.float64)).to(self.device)
        self.speed = float(geometry.speed)


def _as_dev(a, device):
    torch = _torch()
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
    return a.to(device=device, dtype=torch.float64).contiguous()


def _synth_args(template, mode, delays, doa, geometry, moving, shift, gain, out):
    """MiclocSynthArgs for the generators below; returns (args, x, tensors to keep alive until the call has been issued)."""
    torch = _torch()
    dev = template.device
    T = template.T
    args = _lib.MiclocSynthArgs()
    keep = []
    if delays is not None:
        d = _as_dev(delays, dev)
        if d.dim() != (4 if moving else 3):
            raise ValueError("delays must be [B, K, M] (constant DoA) or [B, K, T, M] (moving)")
        B, K, M = d.shape[0], d.shape[1], d.shape[-1]
        if moving and d.shape[2] != T:
            raise ValueError("moving delays need one row per template sample")
        args.delays = d.data_ptr()
        keep.append(d)
    else:
        if doa is None or geometry is None:
            raise ValueError("either delays or (doa, geometry) must be given")
        a = _as_dev(doa, dev)
        if a.dim() != (3 if moving else 2):
            raise ValueError("doa must be [B, K] (constant) or [B, K, T] (moving)")
        B, K, M = a.shape[0], a.shape[1], geometry.M
        if moving and a.shape[2] != T:
            raise ValueError("moving DoAs need one value per template sample")
        args.doa = a.data_ptr()
        args.r_vec, args.theta_vec, args.speed = geometry.r_vec.data_ptr(), geometry.theta_vec.data_ptr(), geometry.speed
        keep.append(a)
    sh = _as_dev(shift, dev)
    gn = _as_dev(gain, dev)
    if sh is not None and tuple(sh.shape) != (B,):
        raise ValueError("shift must be [B]")
    if gn is not None and tuple(gn.shape) != (B, K, T):
        raise ValueError("gain must be [B, K, T]")
    x = out if out is not None else torch.empty((B, T, M), dtype=torch.float64, device=dev)
    if tuple(x.shape) != (B, T, M) or x.dtype != torch.float64 or not x.is_contiguous():
        raise ValueError("out must be a contiguous float64 [B, T, M] tensor")
    args.time, args.sig, args.slopes = template.time.data_ptr(), template.sig.data_ptr(), template.slopes.data_ptr()
    args.T, args.B, args.K, args.M = T, B, K, M
    args.moving = int(bool(moving))
    args.shift = sh.data_ptr() if sh is not None else None
    args.gain = gn.data_ptr() if gn is not None else None
    args.mode = {"apply_to_template": 0, "signal_from_template": 1}[mode]
    args.fs = template.fs
    args.x = x.data_ptr()
    return args, x, (keep, sh, gn)


def synth_targets(template, mode, delays=None, doa=None, geometry=None, moving=False, shift=None, gain=None, out=None):
    """micloc_synth_targets_f64: x [B, T, M] = sum_k gain * interp(...) (see include/micloc_hip.h).

    delays [B, K, (T,) M] (host NumPy -> bit-exact parity) or doa [B, K(, T)] + `geometry` (runtime.Geometry: delays
    computed in the kernel).  mode: "apply_to_template" (t - (d - shift), clamped) or "signal_from_template" (t + d)."""
    args, x, keep = _synth_args(template, mode, delays, doa, geometry, moving, shift, gain, out)
    _lib.check(_lib.load().micloc_synth_targets_f64(ctypes.byref(args), _stream(template.device)), "synth_targets")
    del keep
    return x


def synth_awgn(template, mode, snr_db, seed=0, substream=0, first_trial=0, epoch=None, ws=None, delays=None, doa=None, geometry=None,
               moving=False, shift=None, gain=None, out=None):
    """micloc_synth_awgn_f64: synth_targets(...) followed by awgn_(x, snr_db=...) in two passes that never store the noise-free
    signal (one trip through HBM instead of four); the same bits as the two calls."""
    torch = _torch()
    lib = _lib.load()
    dev = template.device
    args, x, keep = _synth_args(template, mode, delays, doa, geometry, moving, shift, gain, out)
    B, T, M = x.shape
    if not isinstance(snr_db, torch.Tensor):
        snr_db = np.array(np.broadcast_to(np.asarray(snr_db, dtype=np.float64), (B,)))
    s_db = _as_dev(snr_db, dev)
    nbytes = lib.micloc_synth_awgn_workspace_bytes(B, T, M, int(args.K))
    if ws is None:
        ws = _op_workspace(dev, nbytes)
    elif ws.numel() < nbytes:
        raise ValueError("synth_awgn workspace too small (micloc_synth_awgn_workspace_bytes)")
    _lib.check(lib.micloc_synth_awgn_f64(ctypes.byref(args), _ptr(s_db), int(seed), int(substream), _ptr(epoch), int(first_trial), _ptr(ws), nbytes,
                                         _stream(dev)), "synth_awgn")
    del keep
    return x


def delay_min(doa, geometry, moving=False, out=None):
    """shift[b] = min over targets / time / microphones of the un-normalised delays (snn_beamformer.py:257) on the device."""
    torch = _torch()
    lib = _lib.load()
    a = _as_dev(doa, geometry.device)
    B, K = a.shape[0], a.shape[1]
    Td = a.shape[2] if moving else 1
    sh = out if out is not None else torch.empty((B,), dtype=torch.float64, device=a.device)
    _lib.check(lib.micloc_delay_min_f64(_ptr(a), B, K, Td, _ptr(geometry.r_vec), _ptr(geometry.theta_vec), geometry.M, geometry.speed,
                                        _ptr(sh), _stream(a.device)), "delay_min")
    return sh


def uniform(n, seed, substream=0, lo=0.0, hi=1.0, device=None, out=None, epoch=None):
    """n Philox-4x32-10 uniforms in [lo, hi) on the device (np.random.rand's range, 53 bits)."""
    torch = _torch()
    lib = _lib.load()
    device = require_gpu(device) if out is None else out.device
    u = out if out is not None else torch.empty((int(n),), dtype=torch.float64, device=device)
    _lib.check(lib.micloc_uniform_f64(_ptr(u), u.numel(), int(seed), int(substream), _ptr(epoch), float(lo), float(hi), _stream(device)), "uniform")
    return u


def awgn_(x, snr_db=None, sigma=None, seed=0, substream=0, first_trial=0, ws=None, epoch=None):
    """In place: x[b] += sigma_b N(0, 1), sigma_b = sqrt(mean(x[b]^2)) / sqrt(10^(snr_db[b]/10)) (snn_beamformer.py:270-275),
    Philox-4x32-10 + Box-Muller on the device.  x [B, T, M] device tensor; snr_db / sigma [B] (device tensor or NumPy)."""
    lib = _lib.load()
    B, T, M = x.shape
    dev = x.device
    torch = _torch()
    if snr_db is not None and not isinstance(snr_db, torch.Tensor):
        snr_db = np.array(np.broadcast_to(np.asarray(snr_db, dtype=np.float64), (B,)))
    s_db = _as_dev(snr_db, dev)
    sg = _as_dev(sigma, dev)
    if s_db is None and sg is None:
        raise ValueError("awgn_ needs snr_db or sigma")
    nbytes = lib.micloc_awgn_workspace_bytes(B, T, M)
    if ws is None:
        ws = _op_workspace(dev, nbytes)  # (a caller that captures the call into a HIP graph passes its own buffer)
    elif ws.numel() < nbytes:
        raise ValueError("awgn workspace too small")
    _lib.check(lib.micloc_awgn_f64(_ptr(x), B, T, M, _ptr(s_db), _ptr(sg), int(seed), int(substream), _ptr(epoch), int(first_trial), _ptr(ws), nbytes,
                                   _stream(dev)), "awgn")
    return x


def design_vectors(cov, bipolar, bf_mat, g0, rel_prec=1e-8):
    """micloc_design_vectors_f64: cov [n, C, C] device tensor -> columns g0 .. g0 + n of the device tensor bf_mat [C, G]."""
    n, C, _ = cov.shape
    _lib.check(_lib.load().micloc_design_vectors_f64(_ptr(cov), n, C, int(bool(bipolar)), float(rel_prec), _ptr(bf_mat), bf_mat.shape[1], int(g0),
                                                     _stream(cov.device)), "design_vectors")
    return bf_mat


def planar_gram(planar, T, t_start=0, normalise=True):
    """micloc_planar_gram_f64: planar [B, C, Ts] device tensor -> gram [B, C, C] = sum over frames t_start <= t < T of
    x[:, t] x[:, t]^T (/ (T - t_start)).  The complex covariance of Beamformer.design_from_template is a fold of it."""
    torch = _torch()
    B, C, Ts = planar.shape
    lib = _lib.load()
    nbytes = lib.micloc_planar_gram_workspace_bytes(B, int(T), C, int(t_start))
    if nbytes == 0:
        raise ValueError("planar_gram: bad shape or t_start")
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=planar.device)
    gram = torch.empty((B, C, C), dtype=torch.float64, device=planar.device)
    _lib.check(lib.micloc_planar_gram_f64(_ptr(planar), B, C, int(T), Ts, int(t_start), int(bool(normalise)), _ptr(gram), _ptr(ws), nbytes,
                                          _stream(planar.device)), "planar_gram")
    return gram


def peak_location(counts, G, win_size, out=None):
    """micloc_peak_location_i32: counts int32 [B, bands * G] -> index int32 [B] (find_peak_location of the per-DoA counts)."""
    torch = _torch()
    B, FG = counts.shape
    if FG % G != 0:
        raise ValueError("the number of hidden neurons must be a multiple of the DoA grid size")
    if win_size % 2 != 1:
        raise ValueError("averaging window size should be odd to not create confusion in peak index!")
    if win_size > G // 2:
        raise ValueError("size of averaging window is larger than half the length of input signal!")
    idx = out if out is not None else torch.empty((B,), dtype=torch.int32, device=counts.device)
    _lib.check(_lib.load().micloc_peak_location_i32(_ptr(counts), B, int(G), FG // int(G), int(win_size), _ptr(idx), _stream(counts.device)),
               "peak_location")
    return idx


def envelope_track(y, win_fall, win_rise, want_index=True, env_out=None):
    """micloc_envelope_track_any: y [T, G] or [B, T, G] device tensor (float64: the SNN beamformer's output; complex128: the complex
    Beamformer's, paper_plots/target_localization.py:597-600; uint8 / int32 / int64: a spike raster, paper_plots/target_xylo_localization.py:757-768)
    -> (env float64 of y's shape, index int32 [T] / [B, T] or None).
    Envelope.evolve (micloc/utils.py:36-81) + np.argmax(axis=1) (paper_plots/target_snn_localization.py:599-622) on the device."""
    torch = _torch()
    kinds = {torch.float64: 0, torch.complex128: 1, torch.uint8: 2, torch.int32: 3, torch.int64: 4}  # MICLOC_ENV_*
    if y.dtype not in kinds or y.dim() not in (2, 3) or not y.is_cuda:
        raise ValueError("envelope_track: a device tensor [T, G] or [B, T, G] of float64 / complex128 / uint8 / int32 / int64 is required")
    if int(win_fall) < 1 or int(win_rise) < 1:
        raise ValueError("envelope windows must hold at least one sample (int(fs * time) >= 1)")
    yc = y.contiguous()
    B, T, G = (1, *yc.shape) if yc.dim() == 2 else yc.shape
    # NumPy's own arithmetic for the two filter constants (utils.py:70-76: `1 / win_len_state`, `1 - 1 / win_len_state`)
    wl = np.asarray([int(win_fall), int(win_rise)])
    inv = 1 / wl
    a = 1 - inv
    env = env_out if env_out is not None else torch.empty(yc.shape, dtype=torch.float64, device=yc.device)
    idx = torch.empty(yc.shape[:-1], dtype=torch.int32, device=yc.device) if want_index else None
    _lib.check(_lib.load().micloc_envelope_track_any(_ptr(yc), kinds[y.dtype], int(B), int(T), int(G), float(a[1]), float(inv[1]), float(a[0]),
                                                     _ptr(env), _ptr(idx) if idx is not None else None, _stream(yc.device)), "envelope_track")
    return env, idx


def counter_add_(counter, inc=1):
    """*counter += inc on the device (counter: uint32/int32 device tensor of one element): the `epoch` of the generators."""
    _lib.check(_lib.load().micloc_counter_add_u32(_ptr(counter), int(inc), _stream(counter.device)), "counter_add")
    return counter


def awgn_workspace(B, T, M, device, K=1):
    """Workspace of awgn_ and synth_awgn (K targets) for [B, T, M] signals."""
    torch = _torch()
    return torch.empty(int(_lib.load().micloc_synth_awgn_workspace_bytes(int(B), int(T), int(M), int(K))), dtype=torch.uint8, device=device)


def doa_error(argmax, doa_list, doa_true, groups=1, want_err=True):
    """Device tensors argmax [B] int32, doa_list [G] f64, doa_true [B] f64 -> (err [B] or None, mae [groups]):
    arcsin|sin(doa_list[argmax] - doa_true)| and its mean per SNR group (target_snn_localization.py:464-467, :520)."""
    torch = _torch()
    lib = _lib.load()
    B = argmax.shape[0]
    dev = argmax.device
    err = torch.empty((B,), dtype=torch.float64, device=dev) if want_err else None
    mae = torch.empty((groups,), dtype=torch.float64, device=dev)
    _lib.check(lib.micloc_doa_error_f64(_ptr(argmax), _ptr(doa_list), doa_list.shape[0], _ptr(doa_true), B, int(groups), _ptr(err), _ptr(mae),
                                        _stream(dev)), "doa_error")
    return err, mae


STAGE_STHT, STAGE_ENCODE, STAGE_BEAMFORM, STAGE_ENCODE_SCAN, STAGE_ENCODE_REST = 1, 2, 4, 8, 16  # include/micloc_hip.h MICLOC_STAGE_*


class CuRangeStream:
    """A HIP stream restricted to compute units [cu_lo, cu_hi) of every XCD (micloc_stream_create_cu_range), as a torch stream
    (`.stream`).  Take them from cu_range_streams(): a masked stream holds a hardware queue for as long as it exists (a process that
    keeps creating them slows down launch by launch), and torch's caching allocator remembers the stream of every block allocated
    under it -- destroying one while such a block is alive ends in a crash at the block's release.  Streams of the pool therefore
    live as long as the process; close() is for a caller that owns everything allocated under the stream."""

    def __init__(self, device, cu_lo, cu_hi):
        torch = _torch()

        self.lib = _lib.load()
        dev = require_gpu(device)
        h = ctypes.c_void_p()
        _lib.check(self.lib.micloc_stream_create_cu_range(dev.index, int(cu_lo), int(cu_hi), ctypes.byref(h)), "stream_create_cu_range")
        self.handle = h
        self.stream = torch.cuda.ExternalStream(h.value, device=dev)

    def close(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            self.stream.synchronize()
            _lib.check(self.lib.micloc_stream_destroy(h), "stream_destroy")
        self.handle = None


_cu_range_pool = {}


def cu_range_streams(device, cu_lo, cu_hi, n=1):
    """The first n streams of the process-wide pool of streams restricted to compute units [cu_lo, cu_hi) of every XCD of `device`
    (created on demand, never destroyed: see CuRangeStream).  Pipelines that are alive at the same time share them."""
    dev = require_gpu(device)
    pool = _cu_range_pool.setdefault((dev.index, int(cu_lo), int(cu_hi)), [])
    while len(pool) < n:
        pool.append(CuRangeStream(dev, cu_lo, cu_hi))
    return [m.stream for m in pool[:n]]


class StreamPipeline:
    """Round-robin dispatch of consecutive batches over several HIP streams, one Plan (= workspace) per stream.

    The band-pass/RZCC kernel is latency bound (one wave per 64 streams, the time axis is sequential) and leaves
    most of the chip idle; running batch i+1's STHT / batch i-1's beamformer next to it on other streams fills
    the machine.  Results are device tensors owned by the stream that produced them: call synchronize() (or make
    the consumer stream wait) before reading them elsewhere.

    scan_lane = k > 0 (long recordings, whose encoder is time-chunked: Plan.encoder_chunks(B, T) > 1): the serial checkpoint scans
    of ALL plans go through one extra stream that owns compute units [0, k) of every XCD, the plans' own streams get the other
    32 - k (cu_range_streams: a process-wide pool).  A scan is a chain of dependent fp64 operations on few workgroups (speech sweep: 28): on
    a SIMD that also issues another kernel's matrix instructions it runs at half its pace or less, and streams left to themselves
    fall into step -- all scans together, the chip idle beside them.  With the lane the scans of consecutive batches run back to
    back at full pace and the throughput stages of the other batches fill the rest of the chip (snn_pipeline below; eager launches
    with event dependencies -- a captured graph has no streams).  Speech sweep, 125 trials x 332 157 frames per step: 15.6 -> 14.1
    ms per step with four plans (tools/dev/speech_lane.py).
    """

    def __init__(self, plans, scan_lane=0):
        torch = _torch()
        self.plans = list(plans)
        self.device = self.plans[0].device
        self.scan_lane = int(scan_lane)
        if self.scan_lane > 0:
            cus = torch.cuda.get_device_properties(self.device).multi_processor_count // 8
            if not 0 < self.scan_lane < cus:
                raise ValueError(f"scan_lane must leave compute units to both sides: 1 .. {cus - 1} per XCD")
            self.lane = cu_range_streams(self.device, 0, self.scan_lane)[0]
            self.streams = cu_range_streams(self.device, self.scan_lane, cus, len(self.plans))
            self._ev = [(torch.cuda.Event(), torch.cuda.Event()) for _ in self.plans]
        else:
            self.lane = None
            self.streams = [torch.cuda.Stream(device=self.device) for _ in self.plans]
        self._next = 0

    def snn_pipeline(self, x_of, before=None, after=None, index=None, **kw):
        """One batch through plan i's snn_pipeline on stream i (round-robin, or `index`), launched eagerly: x_of(i) is the input
        tensor, before(i) / after(i, out) run on the same stream in front of / behind it (synthesis; DoA error).  With a scan lane the
        encoder's serial scan is enqueued on the lane between two events; without one this is submit().  Returns (out, after's
        result); keyword arguments go to Plan.snn_pipeline (`out` may be a list with one entry per plan)."""
        torch = _torch()
        if index is None:
            i = self._next % len(self.plans)
            self._next += 1
        else:
            i = int(index) % len(self.plans)
        plan, s = self.plans[i], self.streams[i]
        outs = kw.pop("out", None)
        out = outs[i] if isinstance(outs, list) else outs
        cur = torch.cuda.current_stream(self.device)
        # hipExtStreamCreateWithCUMask makes BLOCKING streams: the legacy default stream already orders them behind its earlier work,
        # and an event recorded on it would in turn wait for every one of them -- the batches in flight would run one after the other
        if self.lane is None or cur != torch.cuda.default_stream(self.device):
            s.wait_stream(cur)
        with torch.cuda.stream(s):
            if before is not None:
                before(i)
            x = x_of(i)
            B, T, _ = x.shape
            if self.lane is None or plan.encoder_chunks(B, T) <= 1:
                out = plan.snn_pipeline(x, out=out, **kw)
            else:
                ev_a, ev_b = self._ev[i]
                out = plan.snn_pipeline(x, out=out, stages=STAGE_STHT, **kw)
                ev_a.record(s)
                self.lane.wait_event(ev_a)
                with torch.cuda.stream(self.lane):
                    plan.snn_pipeline(x, out=out, stages=STAGE_ENCODE_SCAN, **kw)
                    ev_b.record(self.lane)
                s.wait_event(ev_b)
                plan.snn_pipeline(x, out=out, stages=STAGE_ENCODE_REST | STAGE_BEAMFORM, **kw)
            if isinstance(outs, list):
                outs[i] = out
            res = after(i, out) if after is not None else None
        return out, res

    def submit(self, fn):
        """Run fn(plan) on the next stream; returns fn's result."""
        torch = _torch()
        i = self._next % len(self.plans)
        self._next += 1
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            return fn(self.plans[i])

    def capture(self, fn):
        """Capture fn(plan) once per stream into a HIP graph (the C-ABI stage calls neither allocate nor synchronise,
        include/micloc_hip.h).  Returns a callable `replay()` that launches the next stream's graph round-robin and
        returns that stream's (static) outputs.  fn must read its inputs from tensors that outlive the graphs."""
        torch = _torch()
        graphs = []
        for plan, s in zip(self.plans, self.streams):
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                fn(plan)  # warm-up outside capture: workspaces and lazily-set kernel attributes
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread_local: other threads (e.g. the RCCL watchdog) may touch the runtime while this thread captures
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                out = fn(plan)
            graphs.append((g, out, plan.generation))
        state = {"next": 0}
        plans = self.plans

        def replay(index=None):
            """Launch the next stream's graph (round-robin), or stream `index`'s without advancing the rotation."""
            if index is None:
                i = state["next"] % len(graphs)
                state["next"] += 1
            else:
                i = int(index) % len(graphs)
            if plans[i].generation != graphs[i][2]:
                # the graph holds raw device pointers of tables / workspace that have been re-allocated since
                raise _lib.MiclocError("stale HIP graph: the plan's tables or workspace were re-allocated, or a table changed its "
                                       "shape, after capture (set_bf_mat / set_neuron_kernel with another size, or a larger batch); "
                                       "capture again")
            with torch.cuda.stream(self.streams[i]):
                graphs[i][0].replay()
            return graphs[i][1]

        return replay

    def synchronize(self):
        torch = _torch()
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)
        cur.synchronize()
