"""Xylo-based localisation (BASELINE config 4) with the call surface of the reference's
micloc/xylo_snn_localization.py, without rockpool / samna / hardware.

reference                                   -> here
  signal_from_template          :44-71      -> same arithmetic, vectorised over time (host)
  Demo.__init__                 :75-171     -> per-band SNNBeamformer designs (device chain + host SVD), filterbank
  Demo._initialize_snn_module   :173-313    -> `xylo_specification`: NumPy restatement of the network the reference
                                               hands to rockpool (weights [bf_mat; -bf_mat], tau * fs/1000, shared
                                               w_rec = -0.1/N, threshold 1) + global 8-bit quantisation
  Demo.spike_encoding           :315-356    -> STHT -> order-1 band-pass -> RZCC on the GPU, +/- split     (PINNED:
                                               tests/golden/filterbank.npz comes from the reference's own classes)
  Demo.xylo_process             :358-377    -> micloc_xylo_lif_i16 (integer LIF kernel)                   (UNPINNED)
  Demo.extract_rate / estimate_doa_from_rate :379-444 -> same formulas (host)
  live demo / power measurement :446-682    -> out of scope (needs the physical Xylo board and microphone array)

PARITY UNPINNED for xylo_process and the quantiser: in the reference both are rockpool code
(`mapper`, `global_quantize`, `config_from_specification`, `XyloSim`), an un-vendored, un-pinned third-party
dependency that is not installed in the build image, and the reference has no tests at that boundary.  The
restatement follows the published Xylo-A2 update rule and rockpool's documented global quantisation; it has not
been checked against XyloSim output.
"""
import ctypes
from numbers import Number

import numpy as np

from . import _lib, runtime
from .filterbank import ButterworthFilterbank
from .snn_beamformer import SNNBeamformer


def signal_from_template(geometry, template):
    """`T x num_mic` array signal: template delayed by +delays(doa) (no min-shift, no clamping; np.interp saturates)."""
    time_temp, sig_temp, doa_temp = template
    time_temp = np.asarray(time_temp, dtype=np.float64)
    if isinstance(doa_temp, Number):
        doa_temp = doa_temp * np.ones_like(time_temp)
    delays = geometry.delays(np.asarray(doa_temp, dtype=np.float64), normalized=False)  # [T, M]
    time_delays = time_temp.reshape(-1, 1) + delays
    return np.interp(time_delays.ravel(), time_temp, sig_temp).reshape(*time_delays.shape)


def xylo_specification(bf_mats, tau_vecs, fs, target_dt, bipolar_spikes, threshold=1.0, w_rec_coef=-0.1, bits_per_weight=8):
    """Quantised hidden-layer specification (what mapper + global_quantize produce for the reference's network).

    Returns dict(W_in int8 [Cin, N], w_rec int, dash_syn uint8 [N], dash_mem uint8 [N], threshold int16 [N], scaling)."""
    num_freq = len(bf_mats)
    cin, cout = bf_mats[0].shape
    W = np.zeros((num_freq * cin, num_freq * cout))
    for ch in range(num_freq):
        W[ch * cin : (ch + 1) * cin, ch * cout : (ch + 1) * cout] = bf_mats[ch]
    if bipolar_spikes:
        W = np.vstack([W, -W])
    W = W.astype(np.float32).astype(np.float64)  # the reference stores the weights as float32 tensors
    N = W.shape[1]
    w_rec = np.float64(np.float32(w_rec_coef / N))
    scale_t = fs / (1.0 / target_dt)
    tau = np.asarray(tau_vecs, dtype=np.float64) * scale_t  # [F, 2] (tau_syn, tau_mem)
    tau_syn = np.repeat(tau[:, 0], cout)
    tau_mem = np.repeat(tau[:, 1], cout)
    # bit-shift decay constants: dash = round(log2(tau / dt))
    dash_syn = np.clip(np.round(np.log2(tau_syn / target_dt)), 0, 15).astype(np.uint8)
    dash_mem = np.clip(np.round(np.log2(tau_mem / target_dt)), 0, 15).astype(np.uint8)
    # one global scale maps the largest |weight| (input and recurrent) to the int8 range; thresholds share it
    max_w = max(np.abs(W).max(), abs(w_rec))
    max_q = 2 ** (bits_per_weight - 1) - 1
    scaling = max_q / max_w if max_w > 0 else 1.0
    W_q = np.round(W * scaling).astype(np.int8)
    w_rec_q = int(np.round(w_rec * scaling))
    thr_q = np.full(N, int(np.clip(np.round(threshold * scaling), 1, 32767)), dtype=np.int16)
    return dict(W_in=W_q, w_rec=w_rec_q, dash_syn=dash_syn, dash_mem=dash_mem, threshold=thr_q, scaling=scaling)


def xylo_lif(spikes_in, spec, max_spikes=31, want_spikes=True, device=None):
    """Run the integer LIF kernel. spikes_in: [T, Cin] or [B, T, Cin] (numpy / device tensor, small non-negative ints).
    Returns (spikes_out uint8 device tensor or None, rate int32 device tensor [.., N])."""
    import torch

    lib = _lib.load()
    device = runtime.require_gpu(device)
    if isinstance(spikes_in, np.ndarray):
        spikes_in = torch.from_numpy(np.ascontiguousarray(spikes_in.astype(np.uint8)))
    s = spikes_in.to(device=device, dtype=torch.uint8).contiguous()
    squeeze = s.dim() == 2
    if squeeze:
        s = s.unsqueeze(0)
    B, T, Cin = s.shape
    W = np.ascontiguousarray(spec["W_in"], dtype=np.int8)
    if W.shape[0] != Cin:
        raise ValueError(f"number of input spike channels {Cin} should match the weight matrix {W.shape[0]}")
    N = W.shape[1]
    ds = np.ascontiguousarray(spec["dash_syn"], dtype=np.uint8)
    dm = np.ascontiguousarray(spec["dash_mem"], dtype=np.uint8)
    th = np.ascontiguousarray(spec["threshold"], dtype=np.int16)
    out = torch.empty((B, T, N), dtype=torch.uint8, device=device) if want_spikes else None
    rate = torch.empty((B, N), dtype=torch.int32, device=device)
    nbytes = lib.micloc_xylo_workspace_bytes(Cin, N)
    ws = runtime._op_workspace(device, nbytes)
    vp = ctypes.c_void_p
    _lib.check(
        lib.micloc_xylo_lif_i16(runtime._ptr(s), B, T, Cin, vp(W.ctypes.data), N, int(spec["w_rec"]), vp(ds.ctypes.data), vp(dm.ctypes.data),
                                vp(th.ctypes.data), int(max_spikes), runtime._ptr(out), runtime._ptr(rate), runtime._ptr(ws), nbytes,
                                runtime._stream(device)),
        "xylo_lif",
    )
    if squeeze:
        return (out[0] if out is not None else None), rate[0]
    return out, rate


class XyloNetwork:
    """The quantised hidden layer resident on the device (micloc_xylo_upload once, then micloc_xylo_lif_resident_i16 per
    batch: no host access, no synchronisation -- capturable into a HIP graph).  PARITY UNPINNED like xylo_lif."""

    def __init__(self, spec, device=None):
        import torch

        self.lib = _lib.load()
        self.device = runtime.require_gpu(device)
        W = np.ascontiguousarray(spec["W_in"], dtype=np.int8)
        self.Cin, self.N = W.shape
        self.w_rec = int(spec["w_rec"])
        ds = np.ascontiguousarray(spec["dash_syn"], dtype=np.uint8)
        dm = np.ascontiguousarray(spec["dash_mem"], dtype=np.uint8)
        th = np.ascontiguousarray(spec["threshold"], dtype=np.int16)
        self.last_scratch = None
        self.nbytes = self.lib.micloc_xylo_workspace_bytes(self.Cin, self.N)
        self.ws = torch.empty(int(self.nbytes), dtype=torch.uint8, device=self.device)
        vp = ctypes.c_void_p
        _lib.check(self.lib.micloc_xylo_upload(self.Cin, vp(W.ctypes.data), self.N, vp(ds.ctypes.data), vp(dm.ctypes.data), vp(th.ctypes.data),
                                               runtime._ptr(self.ws), self.nbytes, runtime._stream(self.device)), "xylo_upload")

    def run(self, spikes, ternary=False, want_spikes=False, max_spikes=31, queued=True, workers_per_cu=0):
        """spikes: uint8 events [B, T, Cin], or (ternary=True) the encoder's int8 raster [B, T, Cin / 2] in {-1, 0, +1}.
        Returns (spikes_out uint8 [B, T, N] or None, rate int32 [B, N]) as device tensors.  queued=False: counts-only calls also take the
        one-workgroup-per-trial launch (micloc_xylo_lif_resident_i16) instead of the ticket queue -- same counts (tests, A/B timing)."""
        import torch

        B, T, C = spikes.shape
        if (2 * C if ternary else C) != self.Cin:
            raise ValueError(f"number of input spike channels {2 * C if ternary else C} should match the weight matrix {self.Cin}")
        if spikes.dtype != (torch.int8 if ternary else torch.uint8) or not spikes.is_contiguous():
            raise ValueError("spikes must be a contiguous int8 (ternary) / uint8 (events) device tensor")
        out = torch.empty((B, T, self.N), dtype=torch.uint8, device=self.device) if want_spikes else None
        rate = torch.empty((B, self.N), dtype=torch.int32, device=self.device)
        if ternary and not want_spikes and self.w_rec == 0 and queued:
            # the sweep's call: counts only -> persistent workgroups on a (trial, time chunk) ticket queue (micloc_xylo_lif_sweep_i16)
            nb = self.lib.micloc_xylo_sweep_scratch_bytes(B)
            scratch = torch.empty(int(nb), dtype=torch.uint8, device=self.device)  # one per call: calls of several streams run side by side
            _lib.check(self.lib.micloc_xylo_lif_sweep_i16(runtime._ptr(spikes), C, B, T, self.Cin, self.N, int(max_spikes), runtime._ptr(rate),
                                                          runtime._ptr(self.ws), self.nbytes, runtime._ptr(scratch), nb, int(workers_per_cu), runtime._stream(self.device)),
                       "xylo_lif_sweep")
            self.last_scratch = scratch  # (queue_status reads its control words)
            return None, rate
        _lib.check(self.lib.micloc_xylo_lif_resident_i16(runtime._ptr(spikes), C if ternary else 0, B, T, self.Cin, self.N, self.w_rec, int(max_spikes),
                                                         runtime._ptr(out), runtime._ptr(rate), runtime._ptr(self.ws), self.nbytes,
                                                         runtime._stream(self.device)), "xylo_lif_resident")
        return out, rate


    def queue_status(self):
        """dict(tickets, gave_up) of the last ticket-queue launch (micloc_xylo_sweep_status; synchronises): gave_up must be 0."""
        st = (ctypes.c_int * 2)()
        _lib.check(self.lib.micloc_xylo_sweep_status(runtime._ptr(self.last_scratch), st, runtime._stream(self.device)), "xylo_sweep_status")
        return dict(tickets=int(st[0]), gave_up=int(st[1]))

    def check(self):
        """Raise if a worker of the last ticket-queue launch gave up waiting for its predecessor (a broken launch: its counts are
        marked -1, never plausible numbers).  Synchronises; callers use it where they synchronise anyway (results to the host)."""
        if getattr(self, "last_scratch", None) is not None and self.queue_status()["gave_up"]:
            raise _lib.MiclocError("xylo ticket queue: a worker gave up waiting for its predecessor's state; the counts of this launch are invalid")


class Demo:
    def __init__(self, geometry, freq_bands, doa_list, recording_duration=0.25, kernel_duration=10e-3, bipolar_spikes=True,
                 xylosim_version=True, fs=48_000, device=None):
        self.freq_bands = np.asarray(freq_bands)
        if self.freq_bands.ndim == 1:
            self.freq_bands = self.freq_bands.reshape(1, -1)
        self.device = device
        self.beamfs, self.bf_mats, self.tau_vecs = [], [], []
        for freq_range in self.freq_bands:
            freq_mid = np.mean(freq_range)
            tau = 1 / (2 * np.pi * freq_mid)
            tau_vec = [tau, tau]
            self.tau_vecs.append(tau_vec)
            beamf = SNNBeamformer(geometry=geometry, kernel_duration=kernel_duration, freq_range=freq_range, tau_vec=tau_vec,
                                  bipolar_spikes=bipolar_spikes, fs=fs, device=device)
            self.beamfs.append(beamf)
            time_temp = np.arange(0, recording_duration, step=1 / fs)
            sig_temp = np.sin(2 * np.pi * freq_mid * time_temp)
            self.bf_mats.append(beamf.design_from_template(template=(time_temp, sig_temp), doa_list=doa_list))
        self.tau_vecs = np.asarray(self.tau_vecs)
        self.filterbank = ButterworthFilterbank(freq_bands=self.freq_bands, order=1, fs=fs, device=device)
        self.doa_list = np.asarray(doa_list)
        self.recording_duration = recording_duration
        self.kernel_duration = kernel_duration
        self.bipolar_spikes = bipolar_spikes
        self.xylosim_version = True  # there is no hardware path here
        self.fs = fs
        self.dt = 1.0 / fs
        self._band_plans = None
        self._net = None
        self._initialize_snn_module(target_dt=1e-3)

    def _initialize_snn_module(self, target_dt):
        self.spec = xylo_specification(self.bf_mats, self.tau_vecs, self.fs, target_dt, self.bipolar_spikes)

    # ---- spike encoding (pinned) -------------------------------------------------------------------------------
    def _plans(self):
        if self._band_plans is None:
            enc = self.beamfs[0].spk_encoder
            self._band_plans = [runtime.Plan(len(self.beamfs[0].geometry), self.beamfs[0].kernel, b, a, enc.robust_width, enc.bipolar,
                                             device=self.device) for (b, a) in self.filterbank.ba_list]
        return self._band_plans

    def _band_rasters(self, sig_batch):
        """The encoder's int8 raster of every band: STHT + order-1 band-pass + RZCC of the fused pipeline (only the quadrature channels
        go through HBM, the encoder reads the in-phase ones -- the rolled input frames -- from x itself)."""
        out = []
        for plan in self._plans():
            x = plan.to_device(sig_batch)
            out.append(plan.snn_pipeline(x, want_spikes=True, want_power=False, stages=3)["spikes"])
        return out

    def _pack(self, rasters, mode):
        """micloc_pack_events_u8 per band: the channel bookkeeping of spike_encoding (:339-354) in one kernel per band -- torch only
        allocates the result."""
        import torch

        B, T, C = rasters[0].shape
        F = len(rasters)
        width = F * C * (2 if mode == 2 else 1)
        out = torch.empty((B, T, width), dtype=torch.uint8 if mode else torch.int8, device=rasters[0].device)
        for f, r in enumerate(rasters):
            _lib.check(_lib.load().micloc_pack_events_u8(runtime._ptr(r), B, T, C, f, F, mode, runtime._ptr(out), runtime._stream(r.device)),
                       "pack_events")
        return out

    def spike_encoding_device(self, sig_batch):
        """[B, T, M] -> uint8 device tensor [B, T, 2M * F * (2 if bipolar else 1)] of 0/1 events."""
        return self._pack(self._band_rasters(sig_batch), 2 if self.bipolar_spikes else 1)

    def spike_encoding(self, sig_in):
        sig_in = np.ascontiguousarray(sig_in, dtype=np.float64)
        return self.spike_encoding_device(sig_in[None])[0].cpu().numpy().astype(np.int64)

    # ---- integer LIF (unpinned) ---------------------------------------------------------------------------------
    def xylo_process(self, spikes_in):
        out, _ = xylo_lif(np.asarray(spikes_in), self.spec, want_spikes=True, device=self.device)
        return out.cpu().numpy().astype(np.int64)

    def network(self):
        if self._net is None:
            self._net = XyloNetwork(self.spec, device=self.device)
        return self._net

    def raster_device(self, sig_batch):
        """[B, T, M] -> the encoder's int8 raster [B, T, 2M * F] in {-1, 0, +1} (bands concatenated on the channel axis);
        the +/- split of spike_encoding happens inside the LIF kernel's staging loop."""
        per_band = self._band_rasters(sig_batch)
        return per_band[0] if len(per_band) == 1 else self._pack(per_band, 0)

    def counts_batch(self, sig_batch):
        """[B, T, M] noisy array signals -> output spike counts per hidden neuron, int32 device tensor [B, F * G]; nothing of
        size T x N is materialised and only micloc kernels run."""
        import torch

        raster = self.raster_device(sig_batch)
        if not self.bipolar_spikes:  # unipolar encoder: the raster holds 0 / +1 only and IS the event tensor (same bytes, read as uint8)
            return self.network().run(raster.view(dtype=torch.uint8), ternary=False)[1]
        return self.network().run(raster, ternary=True)[1]

    def rate_batch(self, sig_batch):
        """[B, T, M] noisy array signals -> spike rate per DoA [B, G] (device tensor): mean(spikes_out) * fs averaged over
        the bands (extract_rate, xylo_snn_localization.py:379-398)."""
        import torch

        counts = self.counts_batch(sig_batch)
        B, G = counts.shape[0], len(self.doa_list)
        rate = torch.empty((B, G), dtype=torch.float64, device=counts.device)
        _lib.check(_lib.load().micloc_rate_from_counts_f64(runtime._ptr(counts), B, G, counts.shape[1] // G, int(sig_batch.shape[1]), float(self.fs),
                                                           runtime._ptr(rate), runtime._stream(counts.device)), "rate_from_counts")
        return rate

    def peak_batch(self, sig_batch, win_size):
        """[B, T, M] -> find_peak_location(rate / rate.max(), win_size) per trial on the device (int32 tensor [B]):
        the estimator of paper_plots/target_xylo_localization.py:594-604."""
        counts = self.counts_batch(sig_batch)
        return runtime.peak_location(counts, len(self.doa_list), win_size)

    def extract_rate(self, spikes_in):
        rate_channels = np.mean(spikes_in, axis=0) * self.fs
        return rate_channels.reshape(-1, len(self.doa_list)).mean(0)

    def estimate_doa_from_rate(self, spike_rate, method):
        method_list = ["peak", "periodic_ml", "trimmed_periodic_ml"]
        if method not in method_list:
            raise ValueError(f"only the following estimation methods are supported:\n{method_list}")
        if method == "peak":
            return self.doa_list[np.argmax(spike_rate)]
        if method == "periodic_ml":
            return np.angle(np.mean(spike_rate * np.exp(1j * self.doa_list)))
        idx = np.argmax(spike_rate)
        num = len(self.doa_list) // 2
        rng = np.arange(-num // 2, num // 2 + 1) - idx
        return np.angle(np.mean(spike_rate[rng] * np.exp(1j * self.doa_list[rng])))
