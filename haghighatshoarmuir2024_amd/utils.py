"""Host utilities with the reference's call surface (micloc/utils.py: Envelope :15-81,
find_peak_location :84-121).  Both are O(G) / O(T*C) post-processing outside the hot path."""
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
        T, channel = sig_in.shape
        if T < channel:
            warnings.warn("number of channels in the input signal is larger than number of samples in each channel!")
        mag = np.abs(sig_in)
        state = np.array(mag[0], copy=True)
        out = np.empty_like(mag)
        for t in range(1, T):
            out[t - 1] = state
            rising = (mag[t] >= state).astype(int)
            inv_len = 1 / self.win_lens[rising]
            state = (1 - inv_len) * state + inv_len * mag[t] * rising
        out[T - 1] = state
        return out


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
