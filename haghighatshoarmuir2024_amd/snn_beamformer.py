"""SNNBeamformer with the reference's call surface (micloc/snn_beamformer.py), running on MI355X.

reference method                         -> what runs here
  __init__            :25-80             -> same scipy designs (hilbert, butter); a device plan is created lazily
  apply_to_signal     :283-370           -> micloc_snn_pipeline_f64 (STHT, band-pass, RZCC, LIF, beamform) on the GPU
  apply_to_template   :213-281           -> host synthesis (vectorised, same draws from np.random) + the above
  design_from_template:82-211            -> the same device chain per DoA (bf_mat = I gives the membrane
                                            signal), covariance + SVD / secular equation on the host (LAPACK,
                                            like the reference)
plus a batched entry point the reference does not have:
  localize_batch(bf_mat, sig[B,T,M])     -> power [B,G] and arg-max [B] without materialising T x G

There is no CPU fallback: without the HIP library or a GPU every device call raises.
"""
from numbers import Number

import numpy as np

from . import runtime
from .array_geometry import ArrayGeometry
from .spike_encoder import ZeroCrossingSpikeEncoder

# sampling rate of the multi-mic board
Fs = 48_000


def neuron_impulse_response(time_vec, tau_vec):
    """Truncated, normalised alpha kernel (reference :342-361): (t/tau) exp(-t/tau), divided by its sum over the
    WHOLE time axis, cut where the cumulative sum reaches 0.999."""
    tau_syn, tau_mem = tau_vec[0], tau_vec[1]
    if tau_mem != tau_syn:
        # the reference's other branch has a sign error and trips its own assert
        raise ValueError("only tau_syn == tau_mem is supported")
    t = np.asarray(time_vec, dtype=np.float64)
    t = t - t[0]
    h = (t / tau_syn) * np.exp(-t / tau_syn)
    h = h / np.sum(h)
    return h[: int(np.sum(np.cumsum(h) < 0.999))]


def synthesize_array_signal(geometry, fs, time_temp, sig_temp, doa_temp):
    """Noise-free array signal of apply_to_template (reference :239-267), vectorised over time.
    Returns (time_in [T], sig_in_vec [T, M])."""
    time_temp = np.asarray(time_temp, dtype=np.float64)
    sig_temp = np.asarray(sig_temp, dtype=np.float64)
    time_in = np.arange(time_temp.min(), time_temp.max(), step=1 / fs)
    sig_in = np.interp(time_in, time_temp, sig_temp)
    if isinstance(doa_temp, Number):
        # a constant DoA: the reference interpolates a constant series (np.interp returns the constant itself) and calls `delays` with
        # that scalar once per time step (:254-256) -- the same seven numbers every time: computed once, as that very scalar call
        delays = geometry.delays(float(doa_temp), normalized=False).reshape(-1, 1)  # [M, 1]
    else:
        doa_in = np.interp(time_in, time_temp, doa_temp)
        delays = geometry.delays(doa_in, normalized=False).T  # [M, T]
    delays = delays - delays.min()
    time_delayed = time_in.reshape(1, -1) - delays
    np.maximum(time_delayed, time_in.min(), out=time_delayed)
    sig = np.interp(time_delayed.ravel(), time_in, sig_in).reshape(time_delayed.shape).T
    return time_in, np.ascontiguousarray(sig)


class SNNBeamformer:
    def __init__(self, geometry: ArrayGeometry, kernel_duration, freq_range, tau_vec, bipolar_spikes=False, fs=Fs, device=None):
        from scipy.signal import butter, hilbert

        self.geometry = geometry
        self.fs = fs
        self.device = device

        self.kernel_duration = kernel_duration
        self.kernel_length = int(self.fs * self.kernel_duration)
        impulse = np.zeros(self.kernel_length)
        impulse[0] = 1
        self.kernel = np.fft.fftshift(np.imag(hilbert(impulse)))
        self.tau_vec = tau_vec

        try:
            f_low, f_high = freq_range
            if f_low > f_high:
                raise Exception()
        except Exception:
            raise ValueError("freq_range should be a vector consisting of two frequencies f_low < f_high!")
        self.bandpass_filter = butter(2, freq_range, btype="bandpass", analog=False, output="ba", fs=fs)

        robust_width = int(fs / f_high) // 2
        self.bipolar_spikes = bipolar_spikes
        self.spk_encoder = ZeroCrossingSpikeEncoder(fs=self.fs, robust_width=robust_width, bipolar=bipolar_spikes, device=device)
        self._plan = None
        self._plan_key = None

    # ---- device plan ----------------------------------------------------------------------------------
    def plan(self):
        """Device plan for the CURRENT attribute values (kernel, band-pass, encoder settings)."""
        b, a = self.bandpass_filter
        key = (len(self.geometry), np.asarray(self.kernel).tobytes(), np.asarray(b).tobytes(), np.asarray(a).tobytes(),
               int(self.spk_encoder.robust_width), bool(self.spk_encoder.bipolar))
        if self._plan is None or key != self._plan_key:
            self._plan = runtime.Plan(len(self.geometry), self.kernel, b, a, self.spk_encoder.robust_width, self.spk_encoder.bipolar,
                                      device=self.device)
            self._plan_key = key
        return self._plan

    def new_plan(self):
        """A fresh, un-cached device plan (own coefficient tables and workspace): one per HIP stream when
        consecutive batches are pipelined across streams (see runtime.StreamPipeline)."""
        b, a = self.bandpass_filter
        return runtime.Plan(len(self.geometry), self.kernel, b, a, self.spk_encoder.robust_width, self.spk_encoder.bipolar, device=self.device)

    def _resample(self, time_vec, sig_in_vec, num_mic):
        """reference :309-321 (only taken when the time axis is not on the fs grid)."""
        if np.allclose(np.diff(time_vec), 1 / self.fs):
            return time_vec, sig_in_vec
        time_new = np.arange(time_vec[0], time_vec[-1], step=1 / self.fs)
        t_all = np.repeat(time_vec.reshape(1, -1), num_mic, axis=0)
        t_new_all = np.repeat(time_new.reshape(1, -1), num_mic, axis=0)
        sig = np.interp(t_new_all.ravel(), t_all.ravel(), sig_in_vec.ravel()).reshape(-1, num_mic)
        return time_new, sig

    # ---- reference call surface ---------------------------------------------------------------------------
    def apply_to_signal(self, bf_mat, sig_in_vec, to_host=True):
        """Reference :283-370.  to_host=False (not in the reference): the T x G result stays on the device (a torch tensor), e.g. for
        utils.Envelope.track -- the moving-target read-out without the T x G device -> host copy."""
        time_vec, sig_in_vec = sig_in_vec
        twice_num_mic, num_grid = bf_mat.shape
        num_mic = twice_num_mic // 2
        T, num_chan = sig_in_vec.shape
        if num_chan != num_mic:
            raise ValueError(f"number of channels in the input siganl {num_chan} should be the same as the number of microphones {num_mic}!")
        time_vec, sig_in_vec = self._resample(np.asarray(time_vec), np.asarray(sig_in_vec), num_mic)
        plan = self.plan()
        plan.set_neuron_kernel(neuron_impulse_response(time_vec, self.tau_vec))
        plan.set_bf_mat(np.asarray(bf_mat, dtype=np.float64))
        x = plan.to_device(np.asarray(sig_in_vec, dtype=np.float64)[None])
        out = plan.snn_pipeline(x, want_y=True, want_power=False)
        return runtime.to_host(out["y"][0]) if to_host else out["y"][0]

    def apply_to_template(self, bf_mat, template, snr_db, to_host=True):
        try:
            time_temp, sig_temp, doa_temp = template
        except Exception:
            raise ValueError("input template should be a tuple containing (time_in, sig_in, doa_in) of the template signal!")
        snr = 10 ** (snr_db / 10)
        time_in, sig_in_vec = synthesize_array_signal(self.geometry, self.fs, time_temp, sig_temp, doa_temp)
        # same draw from the global legacy stream as the reference (:270-275)
        noise = np.sqrt(np.mean(sig_in_vec**2)) / np.sqrt(snr) * np.random.randn(*sig_in_vec.shape)
        sig_in_vec += noise
        if to_host:  # (the reference's own call, with the reference's signature: an `apply_to_signal` replaced by a caller keeps working)
            return self.apply_to_signal(bf_mat=bf_mat, sig_in_vec=(time_in, sig_in_vec))
        return self.apply_to_signal(bf_mat=bf_mat, sig_in_vec=(time_in, sig_in_vec), to_host=False)

    def synthesize_batch(self, template, doas, device_delays=False):
        """Noise-free array signals for a batch of trials, synthesised on the device (synthesis.apply_to_template_batch).
        template = (time_temp, sig_temp); doas [B] (constant DoA per trial) or [B, len(time_temp)] (moving DoAs).
        device_delays=False: delays from NumPy's cos, bit-exact with the host path / the reference's np.interp;
        True: delays computed in the kernel (throughput runs).  Returns (time_in [T] numpy, x [B, T, M] device)."""
        from . import synthesis

        return synthesis.apply_to_template_batch(self.geometry, self.fs, template, doas, device=self.device, device_delays=device_delays)

    # ---- batched device entry points (not in the reference) --------------------------------------------------
    def localize_batch(self, bf_mat, sig_batch, time_vec=None, return_spikes=False, power_mode="direct"):
        """sig_batch [B, T, M] (numpy or device tensor, already on the fs grid) -> dict of device tensors:
        power [B, G] = mean_t |apply_to_signal|^2, argmax [B] (int32), optionally spikes [B, T, 2M] int8."""
        B, T, M = sig_batch.shape
        if bf_mat.shape[0] // 2 != M:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {bf_mat.shape[0] // 2}!")
        if time_vec is None:
            time_vec = np.arange(T) / self.fs
        plan = self.plan()
        plan.set_neuron_kernel(neuron_impulse_response(time_vec, self.tau_vec))
        plan.set_bf_mat(np.asarray(bf_mat, dtype=np.float64))
        x = plan.to_device(sig_batch)
        if power_mode == "covariance":
            # algebraically identical variant: w^T (V^T V / T) w instead of mean_t (V w)^2  
            return plan.snn_pipeline_cov(x, want_spikes=return_spikes, want_power=True)
        if power_mode != "direct":
            raise ValueError("power_mode must be 'direct' or 'covariance'")
        return plan.snn_pipeline(x, want_spikes=return_spikes, want_power=True)

    def membrane_covariance_batch(self, sig_batch, time_vec=None, t_start=0, out=None):
        """[B, T, M] -> device tensor [B, 2M, 2M] (`out`, if given): V^T V / (T - t_start) of the membrane signal over frames >= t_start."""
        B, T, M = sig_batch.shape
        if time_vec is None:
            time_vec = np.arange(T) / self.fs
        plan = self.plan()
        plan.set_neuron_kernel(neuron_impulse_response(time_vec, self.tau_vec))
        return plan.snn_pipeline_cov(plan.to_device(sig_batch), t_start=t_start, want_cov=True, want_power=False, cov_out=out)["cov"]

    def membrane_batch(self, sig_batch, time_vec=None):
        """[B, T, M] -> device tensor [B, T, 2M]: the membrane signal vmem (bf_mat = identity)."""
        B, T, M = sig_batch.shape
        if time_vec is None:
            time_vec = np.arange(T) / self.fs
        plan = self.plan()
        plan.set_neuron_kernel(neuron_impulse_response(time_vec, self.tau_vec))
        plan.set_bf_mat(np.eye(2 * M))
        return plan.snn_pipeline(plan.to_device(sig_batch), want_y=True, want_power=False)["y"]

    def design_from_template(self, template, doa_list, doa_batch=32, svd="host", device_synthesis=None):
        """Reference :82-211.  The per-DoA chain (delayed template -> STHT -> band-pass -> RZCC -> LIF -> covariance of the
        last 3/4) runs on the device for `doa_batch` DoAs at a time.  svd="host": the 2M x 2M decompositions by LAPACK like
        the reference (same singular-vector phases: bf_mat equals the reference's fixture); svd="device": one batched
        Jacobi kernel (micloc_design_vectors_f64, up to 64 microphones; unipolar columns equal the reference's, bipolar ones up to a residual unit phase
        of ~2e-4 rad: the kernel follows LAPACK's convention, first component real and negative, because the real-projected
        spectrum is not invariant to that phase -- 97 % of the reference's arg-maxima on its accuracy sweep, MAE within 0.03 deg),
        with the delayed templates synthesised on the device as well -- nothing but the template and bf_mat crosses PCIe."""
        if svd not in ("host", "device"):
            raise ValueError("svd must be 'host' or 'device'")
        if device_synthesis is None:
            device_synthesis = svd == "device"
        try:
            time_temp, sig_temp = template
        except Exception:
            raise ValueError("input template should be a tuple containing (time_in, sig_in) of the template signal!")
        time_temp = np.asarray(time_temp, dtype=np.float64)
        time_interp = np.arange(time_temp.min(), time_temp.max(), step=1 / self.fs)
        sig_interp = np.interp(time_interp, time_temp, sig_temp)
        sig_temp, time_temp = sig_interp, time_interp
        doa_list = np.asarray(doa_list, dtype=np.float64)

        bf_mat = []
        bf_dev = None
        if svd == "device":
            import torch

            if 2 * len(self.geometry) > 128:
                raise ValueError("svd='device' supports up to 64 microphones")
            dev = runtime.require_gpu(self.device)
            bf_dev = torch.empty((2 * len(self.geometry), len(doa_list)), dtype=torch.float64, device=dev)
            # every DoA's covariance stays on the device (config 5: 1440 x 128 x 128 doubles = 189 MB) and ALL decompositions
            # run in one launch -- they do not depend on the batching of the chain (48 of them fill a fifth of the chip)
            cov_all = torch.empty((len(doa_list), 2 * len(self.geometry), 2 * len(self.geometry)), dtype=torch.float64, device=dev)
        for start in range(0, len(doa_list), doa_batch):
            doas = doa_list[start : start + doa_batch]
            # delayed, clamped copies of the template, one trial per DoA (reference :141-154)
            delays = self.geometry.delays(doas, normalized=True)  # [n, M]
            delays = delays - delays.min(axis=1, keepdims=True)
            if device_synthesis:
                sig = runtime.synth_delay(time_temp, sig_temp, delays, self.fs, device=self.device)  # [n, T, M], == np.interp bit for bit
            else:
                time_delayed = time_temp.reshape(1, 1, -1) - delays[:, :, None]  # [n, M, T]
                np.maximum(time_delayed, time_temp.min(), out=time_delayed)
                sig = np.interp(time_delayed.ravel(), time_temp, sig_temp).reshape(time_delayed.shape)
                sig = np.ascontiguousarray(np.transpose(sig, (0, 2, 1)))  # [n, T, M]
            if svd == "device":
                self.membrane_covariance_batch(sig, time_vec=time_temp, t_start=sig.shape[1] // 4, out=cov_all[start : start + len(doas)])
                continue
            # membrane covariance over the last 3/4 on the device (MFMA Gram kernels: lif_cov_kernel up to 64 channels,
            # lif_cov_wide_kernel up to 128 -- the plan's limit)
            cov = self.membrane_covariance_batch(sig, time_vec=time_temp, t_start=sig.shape[1] // 4).cpu().numpy()
            for C in cov:
                if not self.spk_encoder.bipolar:
                    bf_mat.append(self._find_dc_removed_sing_vec(C, rel_prec=0.00000001))
                else:
                    d = C.shape[0] // 2
                    C_comp = (C[:d, :d] + C[d:, d:]) / 2 + 1j * ((C[:d, d:] + C[d:, :d].T) / 2)
                    U, _, _ = np.linalg.svd(C_comp)
                    bf_mat.append(np.concatenate([np.real(U[:, 0]), np.imag(U[:, 0])]))
        if svd == "device":
            runtime.design_vectors(cov_all, self.spk_encoder.bipolar, bf_dev, 0, rel_prec=0.00000001)
            return bf_dev.cpu().numpy()
        return np.asarray(bf_mat).T

    def _find_dc_removed_sing_vec(self, C, rel_prec=0.0001):
        """Singular vector of the PSD matrix C conditioned on being orthogonal to the all-one vector
        (reference :372-422): bisection on the secular equation sum_i theta_i^2 / (D_i - u) = 0 in (D_1, D_0)."""
        U, D, _ = np.linalg.svd(C)
        theta = U.T @ np.ones(C.shape[0])
        u_min, u_max = D[1], D[0]
        while (u_max - u_min) / u_min >= rel_prec:
            u_mid = (u_min + u_max) / 2
            if np.sum(theta**2 / (D - u_mid)) < 0.0:
                u_min = u_mid
            else:
                u_max = u_mid
        root = (u_min + u_max) / 2.0
        vec = U @ (theta / (D - root))
        return vec / np.linalg.norm(vec)
