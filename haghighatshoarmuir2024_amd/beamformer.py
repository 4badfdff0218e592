"""Beamformer (non-spiking, complex) with the reference's call surface (micloc/beamformer.py), on MI355X.

reference method                      -> what runs here
  __init__             :26-71         -> same scipy designs (hilbert, butter)
  apply_to_signal      :260-292       -> micloc_beamformer_pipeline_f64: STHT, band-pass, (T x 2M) @ stacked real
                                         form of conj(bf_mat) on the fp64 matrix cores
  apply_to_template    :194-258       -> host synthesis (same np.random draws) + the above
  design_from_template :73-192        -> STHT and the complex covariance (micloc_planar_gram_f64) on the GPU per DoA; the M x M
                                         SVD / generalised eigh on the host (LAPACK, like the reference: its phases), or --
                                         svd="device" -- the batched Jacobi kernel (micloc_design_vectors_f64)
  localize_batch (new)                -> power [B,G], arg-max [B] for a batch of trials, no T x G temporary
"""
from numbers import Number

import numpy as np

from . import runtime
from .array_geometry import ArrayGeometry
from .snn_beamformer import synthesize_array_signal

Fs = 48_000


class Beamformer:
    def __init__(self, geometry: ArrayGeometry, kernel_duration, freq_range, fs=Fs, device=None):
        from scipy.signal import butter, hilbert

        self.geometry = geometry
        self.kernel_duration = kernel_duration
        self.fs = fs
        self.device = device
        impulse = np.zeros(int(fs * kernel_duration))
        impulse[0] = 1
        self.kernel = np.fft.fftshift(np.imag(hilbert(impulse)))
        self.freq_range = np.asarray(freq_range)
        try:
            f_low, f_high = freq_range
            if f_low > f_high:
                raise Exception()
        except Exception:
            raise ValueError("freq_range should be a vector consisting of two frequencies f_low < f_high!")
        self.bandpass_filter = butter(2, freq_range, btype="bandpass", analog=False, output="ba", fs=fs)
        self._plan = None
        self._plan_key = None

    def plan(self):
        b, a = self.bandpass_filter
        key = (len(self.geometry), np.asarray(self.kernel).tobytes(), np.asarray(b).tobytes(), np.asarray(a).tobytes())
        if self._plan is None or key != self._plan_key:
            self._plan = runtime.Plan(len(self.geometry), self.kernel, b, a, robust_width=1, bipolar=False, device=self.device)
            self._plan_key = key
        return self._plan

    def apply_to_signal(self, bf_mat, sig_in, to_host=True):
        """Reference :260-292.  to_host=False (not in the reference): the complex T x G result stays on the device (a complex128 torch
        tensor), e.g. for utils.Envelope.track -- the moving-target read-out of paper_plots/target_localization.py:597-600."""
        num_mic, num_grid = bf_mat.shape
        T, num_chan = sig_in.shape
        if num_chan != num_mic:
            raise ValueError(f"number of channels in the input siganl {num_chan} should be the same as the number of microphones {num_mic}!")
        plan = self.plan()
        plan.set_bf_mat(np.asarray(bf_mat, dtype=np.complex128))
        x = plan.to_device(np.asarray(sig_in, dtype=np.float64)[None])
        y = plan.beamformer_pipeline(x, want_y=True, want_power=False)["y"][0]
        return runtime.to_host(y) if to_host else y

    def localize_batch(self, bf_mat, sig_batch):
        B, T, M = sig_batch.shape
        if bf_mat.shape[0] != M:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {bf_mat.shape[0]}!")
        plan = self.plan()
        plan.set_bf_mat(np.asarray(bf_mat, dtype=np.complex128))
        return plan.beamformer_pipeline(plan.to_device(sig_batch), want_y=False, want_power=True)

    def apply_to_template(self, bf_mat, template, snr_db, to_host=True):
        try:
            time_temp, sig_temp, doa_temp = template
        except Exception:
            raise ValueError("input template should be a tuple containing (time_in, sig_in, doa_in) of the template signal!")
        snr = 10 ** (snr_db / 10)
        _, sig_in_vec = synthesize_array_signal(self.geometry, self.fs, time_temp, sig_temp, doa_temp)
        noise = np.sqrt(np.mean(sig_in_vec**2)) / np.sqrt(snr) * np.random.randn(*sig_in_vec.shape)
        sig_in_vec += noise
        if to_host:  # (the reference's own call with its signature)
            return self.apply_to_signal(bf_mat=bf_mat, sig_in=sig_in_vec)
        return self.apply_to_signal(bf_mat=bf_mat, sig_in=sig_in_vec, to_host=False)

    def design_from_template(self, template, doa_list, interference_removal=False, doa_batch=32, svd="host"):
        """Reference :73-192.  svd="host" (default): the M x M decompositions by LAPACK like the reference (its singular-vector
        phases).  svd="device" (not with interference_removal): the leading singular vectors by the batched Jacobi kernel
        (micloc_design_vectors_f64 on the real embedding [[A, B], [-B, A]] of cov = A + jB), equal to LAPACK's up to the unit phase
        of each column -- the kernel makes the first component real and negative; |bf_mat^H bf_mat| and the beamformed power do not
        depend on it."""
        if svd not in ("host", "device"):
            raise ValueError("svd must be 'host' or 'device'")
        if svd == "device" and interference_removal:
            raise ValueError("interference_removal needs the generalised eigenproblem of the host path (svd='host')")
        try:
            time_temp, sig_temp = template
        except Exception:
            raise ValueError("input template should be a tuple containing (time_in, sig_in) of the template signal!")
        time_temp = np.asarray(time_temp, dtype=np.float64)
        time_interp = np.arange(time_temp.min(), time_temp.max(), step=1 / self.fs)
        sig_temp = np.interp(time_interp, time_temp, sig_temp)
        time_temp = time_interp
        doa_list = np.asarray(doa_list, dtype=np.float64)
        plan = self.plan()
        M = len(self.geometry)

        cov_mat_list = []
        for start in range(0, len(doa_list), doa_batch):
            doas = doa_list[start : start + doa_batch]
            delays = self.geometry.delays(doas, normalized=True)  # [n, M]
            time_delayed = time_temp.reshape(1, 1, -1) - delays[:, :, None]
            np.maximum(time_delayed, time_temp.min(), out=time_delayed)
            sig = np.interp(time_delayed.ravel(), time_temp, sig_temp).reshape(time_delayed.shape)
            sig = np.ascontiguousarray(np.transpose(sig, (0, 2, 1)))  # [n, T, M]
            T = sig.shape[1]
            # NOTE (reference :137-150): the covariance uses the STHT output *before* band-pass filtering
            h = plan.stht(plan.to_device(sig))  # planar [n, 2M, Ts]: rows re_0..re_{M-1}, im_0..im_{M-1}
            stable = min(len(self.kernel), T // 2)
            # conj(h)^T h / T' over the stable part = a fold of the real 2M x 2M Gram matrix (fp64 MFMA kernel, csrc/covariance.hip)
            R = runtime.planar_gram(h, T, t_start=stable, normalise=True).cpu().numpy()
            cov = (R[:, :M, :M] + R[:, M:, M:]) + 1j * (R[:, :M, M:] - R[:, M:, :M])  # [n, M, M]
            cov_mat_list.extend(list(cov))

        bf_mat = []
        if svd == "device":
            import torch

            if 2 * M > 128:
                raise ValueError("svd='device' supports up to 64 microphones")
            cov = np.asarray(cov_mat_list)
            A, B = cov.real, cov.imag
            emb = np.concatenate([np.concatenate([A, B], axis=2), np.concatenate([-B, A], axis=2)], axis=1)  # [n, 2M, 2M]
            dev = runtime.require_gpu(self.device)
            out = torch.empty((2 * M, len(cov_mat_list)), dtype=torch.float64, device=dev)
            runtime.design_vectors(torch.from_numpy(np.ascontiguousarray(emb)).to(dev), True, out, 0)
            W = out.cpu().numpy()
            return W[:M] + 1j * W[M:], cov_mat_list
        if not interference_removal:
            for cov_mat in cov_mat_list:
                U, _, _ = np.linalg.svd(cov_mat)
                bf_mat.append(U[:, 0])
        else:
            from scipy.linalg import eigh

            cov_sum = 0
            for cov_mat in cov_mat_list:
                cov_sum = cov_sum + cov_mat
            cov_sum = cov_sum + np.diag(np.mean(np.diag(cov_sum)) * np.ones(cov_sum.shape[0])) / 10
            for cov_mat in cov_mat_list:
                _, U = eigh(cov_mat, cov_sum - cov_mat)
                vec = U[:, -1]
                bf_mat.append(vec / np.linalg.norm(vec))
        return np.asarray(bf_mat).T, cov_mat_list
