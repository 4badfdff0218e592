"""Host utilities with the reference's call surface (micloc/utils.py: Envelope :15-81,
find_peak_location :84-121).  Post-processing outside the hot path; the moving-target read-out (Envelope over the T x G beamformer
output + per-step arg-max, paper_plots/target_snn_localization.py:599-622) has a device form so that T x G never crosses PCIe."""
import warnings

import numpy as np


class Envelope:
    def __init__(self, rise_time, fall_time, fs):
        if rise_time > fall_time:
            raise ValueError("for proper functioning, an envelope estimator should have a larger fall time!")
        self.rise_time = rise_time
        self.fall_time = fall_time
        self.fs = fs
        # index 0: falling, index 1: rising
        self.win_lens = np.asarray([int(fs * fall_time), int(fs * rise_time)])

    def evolve(self, sig_in):
        """`T x num_chan` in, the envelope of every channel out (reference :36-81).  A float64 DEVICE tensor ([T, G] or a batch
        [B, T, G], e.g. `apply_to_signal(..., to_host=False)`) is processed by the device kernel (micloc_envelope_track_f64: the same
        recurrence in the same order of operations, bit-identical) and a device tensor comes back; a NumPy array takes the reference's
        host loop."""
        if _is_device_tensor(sig_in):
            from . import runtime

            T, channel = sig_in.shape[-2:]
            if T < channel:
                warnings.warn("number of channels in the input signal is larger than number of samples in each channel!")
            return runtime.envelope_track(sig_in, self.win_lens[0], self.win_lens[1], want_index=False)[0]
        T, channel = sig_in.shape
        if T < channel:
            warnings.warn("number of channels in the input signal is larger than number of samples in each channel!")
        # (float64 from the start: what the reference's list of rows becomes in `np.asarray` -- an integer raster, target_xylo_localization.py:
        #  757-768, stays exact; a complex array gives its modulus)
        mag = np.abs(sig_in).astype(np.float64)
        state = np.array(mag[0], copy=True)
        out = np.empty_like(mag)
        for t in range(1, T):
            out[t - 1] = state
            rising = (mag[t] >= state).astype(int)
            inv_len = 1 / self.win_lens[rising]
            state = (1 - inv_len) * state + inv_len * mag[t] * rising
        out[T - 1] = state
        return out

    def track(self, sig_in, want_envelope=False):
        """The moving-target read-out of paper_plots/target_snn_localization.py:599-622 in one call: `np.argmax(self.evolve(sig_bf), axis=1)`
        -- the DoA index per time step -- for a device tensor [T, G] / [B, T, G] without the T x G array leaving the device (17 MB per
        0.1 s at G = 449; the script's 5 s recording: 862 MB): returns int32 indices on the device (and the envelope if asked for).  A
        NumPy array takes the host class and returns NumPy."""
        if _is_device_tensor(sig_in):
            from . import runtime

            env, idx = runtime.envelope_track(sig_in, self.win_lens[0], self.win_lens[1], want_index=True)
            return (idx, env) if want_envelope else idx
        env = self.evolve(sig_in)
        idx = np.argmax(env, axis=1)
        return (idx, env) if want_envelope else idx


def _is_device_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def find_peak_location(sig_in, win_size, periodic=True):
    """argmax of the box-car smoothed signal (full, NON-circular convolution as in the reference)."""
    sig_in = np.asarray(sig_in)
    if sig_in.ndim != 1:
        raise ValueError("input signal should be 1-dim!")
    if win_size % 2 != 1:
        raise ValueError("averaging window size should be odd to not create confusion in peak index!")
    if win_size > len(sig_in) // 2:
        raise ValueError("size of averaging window is larger than half the length of input signal!")
    smoothed = np.convolve(np.ones(win_size), sig_in, mode="full")
    index = int(np.argmax(smoothed)) - win_size // 2
    if periodic:
        index = index % len(sig_in)
    return index
