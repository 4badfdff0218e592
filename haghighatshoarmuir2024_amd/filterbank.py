"""Filterbank with the reference's call surface (micloc/filterbank.py: Filterbank :15-54,
ButterworthFilterbank :57-84).  `evolve` runs every (b, a) section on the MI355X with the same DF2T
kernel the band-pass of the beamformers uses (micloc_lfilter_f64)."""
import numpy as np

from . import runtime


class Filterbank:
    def __init__(self, ba_list, device=None):
        self.ba_list = ba_list
        self.device = device

    def evolve_device(self, sig_in):
        """[T, M] -> device tensor [F, T, M]."""
        import torch

        dev = runtime.require_gpu(self.device)
        x = sig_in if isinstance(sig_in, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(sig_in, dtype=np.float64))
        x = x.to(device=dev, dtype=torch.float64).contiguous()  # uploaded once, every band's kernel writes its own slice of the result
        out = torch.empty((len(self.ba_list),) + tuple(x.shape), dtype=torch.float64, device=dev)
        for f, (b, a) in enumerate(self.ba_list):
            runtime.lfilter(b, a, x, device=dev, out=out[f])
        return out

    def evolve(self, sig_in):
        sig_in = np.asarray(sig_in, dtype=np.float64)
        if sig_in.ndim == 1:
            sig_in = sig_in.reshape(-1, 1)
        return self.evolve_device(np.ascontiguousarray(sig_in)).cpu().numpy()

    def __call__(self, *args, **kwargs):
        return self.evolve(*args, **kwargs)

    def __len__(self):
        return len(self.ba_list)


class ButterworthFilterbank(Filterbank):
    def __init__(self, freq_bands, order, fs, device=None):
        from scipy.signal import butter

        self.order = order
        self.fs = fs
        self.freq_bands = np.asarray(freq_bands)
        if self.freq_bands.ndim == 1:
            self.freq_bands = self.freq_bands.reshape(1, -1)
        ba_list = [butter(order, band, btype="bandpass", output="ba", fs=fs) for band in freq_bands]
        super().__init__(ba_list=ba_list, device=device)
