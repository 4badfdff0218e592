"""Frame-by-frame SNN localisation with the structure of the reference's live demo
(micloc/localization_demo_snn.py:18-193), minus the hardware: the reference records 0.25 s packs with `sox`
(record.py) and pushes the DoA to a matplotlib process (visualizer.py); here `process_frame` is the body of that loop
for one recorded pack and `run` drives it from any iterable of packs, handing each result to a callback.

Per pack (reference :125-193): activity detection on the raw integers -> order-1 Butterworth filterbank ->
per band SNNBeamformer.apply_to_signal -> power per DoA, summed over the bands -> arg-max.  Everything after the
activity check runs on the MI355X: the filterbank through micloc_lfilter_f64, each band through the fused pipeline
(power only; the T x G product the reference materialises is never formed).
"""
import numpy as np

from .filterbank import ButterworthFilterbank
from .snn_beamformer import SNNBeamformer


class Demo:
    def __init__(self, geometry, freq_bands, doa_list, recording_duration, kernel_duration, bipolar_spikes, fs, device=None):
        freq_bands = np.asarray(freq_bands)
        if freq_bands.ndim == 1:
            freq_bands = freq_bands.reshape(1, -1)
        self.beamfs, self.bf_mats = [], []
        for freq_range in freq_bands:
            freq_mid = np.mean(freq_range)
            tau = 1 / (2 * np.pi * freq_mid)
            beamf = SNNBeamformer(geometry=geometry, kernel_duration=kernel_duration, freq_range=freq_range, tau_vec=[tau, tau],
                                  bipolar_spikes=bipolar_spikes, fs=fs, device=device)
            self.beamfs.append(beamf)
            time_temp = np.arange(0, recording_duration, step=1 / fs)
            sig_temp = np.sin(2 * np.pi * freq_mid * time_temp)
            self.bf_mats.append(beamf.design_from_template(template=(time_temp, sig_temp), doa_list=doa_list))
        self.filterbank = ButterworthFilterbank(freq_bands=freq_bands, order=1, fs=fs, device=device)
        self.doa_list = np.asarray(doa_list)
        self.recording_duration = recording_duration
        self.kernel_duration = kernel_duration
        self.fs = fs

    def power_grid(self, data):
        """float [T, num_mic] -> angular power pattern [G] summed over the frequency bands (device -> numpy)."""
        data = np.ascontiguousarray(data, dtype=np.float64)
        T = data.shape[0]
        time_vec = np.arange(0, T) / self.fs
        data_filt = self.filterbank.evolve_device(data)  # [F, T, M] on the device
        total = None
        for chan, (bf_mat, beamf) in enumerate(zip(self.bf_mats, self.beamfs)):
            out = beamf.localize_batch(bf_mat, data_filt[chan : chan + 1], time_vec=time_vec)
            total = out["power"][0] if total is None else total + out["power"][0]
        return total.cpu().numpy()

    def process_frame(self, data, rel_threshold=0.0001):
        """One recorded pack (integer samples, last channel unused as on the devkit, reference :141-150) ->
        DoA in degrees, or NaN when the pack is below the activity threshold."""
        data = np.asarray(data)
        max_value = np.iinfo(data.dtype).max if np.issubdtype(data.dtype, np.integer) else 1.0
        threshold = rel_threshold * max_value
        sig = np.asarray(data[:, :-1], dtype=np.float64)
        if np.sqrt(np.mean(sig**2)) < threshold:
            return np.nan
        return self.doa_list[int(np.argmax(self.power_grid(sig)))] * 180 / np.pi

    def run(self, source, sink=print):
        """`source`: iterable of recorded packs ([T, num_mic + 1] integer arrays); `sink(doa_deg)` replaces the visualiser."""
        for data in source:
            sink(self.process_frame(data))
