"""Array-signal synthesis on the device: the input side of every sweep of the reference.

reference                                                              -> here
  SNNBeamformer / Beamformer.apply_to_template (noise-free part)       -> apply_to_template_batch
      micloc/snn_beamformer.py:239-267, micloc/beamformer.py:220-245      (constant or moving DoA, min-shift, clamp)
  signal_from_template   micloc/xylo_snn_localization.py:44-71         -> signal_from_template_batch   (t + delays, no clamp)
  signal_multiple_targets paper_plots/multiple_targets_snn.py:87-160   -> signal_multiple_targets      (sum of scaled targets)
  noise: sigma * np.random.randn(T, M)  micloc/snn_beamformer.py:270-275 -> add_noise_ (Philox-4x32-10 + Box-Muller kernel)

Two delay sources (csrc/synth.hip): `device_delays=False` uploads delays computed with NumPy's cos (bit-exact with the
reference's np.interp output: the parity path, checked against tests/golden/synth*.npz); `device_delays=True` computes
them inside the kernel from the DoA series (an ulp-level difference in cos; nothing of size B x T x M crosses PCIe: the
throughput path).  The reference evaluates geometry.delays once per time step in a Python loop (60 % of its per-trial
time); here it is one fused kernel per batch.
"""
import numpy as np

from . import runtime


def _resample(time_temp, sig_temp, fs):
    """apply_to_template's grid: np.arange(t.min(), t.max(), 1/fs) (drops the last sample, snn_beamformer.py:246-248)."""
    time_temp = np.asarray(time_temp, dtype=np.float64)
    time_in = np.arange(time_temp.min(), time_temp.max(), step=1 / fs)
    return time_in, np.interp(time_in, time_temp, np.asarray(sig_temp, dtype=np.float64))


def _doa_rows(doas, time_temp, time_in=None):
    """-> (array [B] or [B, T], moving flag); moving series are resampled like the template when time_in is given."""
    doas = np.asarray(doas, dtype=np.float64)
    if doas.ndim == 0:
        doas = doas.reshape(1)
    if doas.ndim == 1:
        return doas, False
    if doas.shape[1] != len(time_temp):
        raise ValueError("a moving DoA needs one value per template sample")
    if time_in is not None:
        doas = np.stack([np.interp(time_in, time_temp, d) for d in doas])  # snn_beamformer.py:250
    return np.ascontiguousarray(doas), True


def apply_to_template_batch(geometry, fs, template, doas, device=None, device_delays=False):
    """Noise-free `sig_in_vec` of apply_to_template for a batch of trials.

    template = (time_temp, sig_temp); doas: [B] constant DoAs or [B, len(time_temp)] DoA time series.
    Returns (time_in [T] numpy, x [B, T, M] device tensor)."""
    time_temp, sig_temp = template
    time_in, sig_in = _resample(time_temp, sig_temp, fs)
    tpl = runtime.Template(time_in, sig_in, fs, device=device)
    doa, moving = _doa_rows(doas, np.asarray(time_temp, dtype=np.float64), time_in)
    if device_delays:
        geo = runtime.Geometry(geometry, device=tpl.device)
        d_doa = doa[:, None, :] if moving else doa[:, None]
        d_doa = runtime._as_dev(np.ascontiguousarray(d_doa), tpl.device)
        shift = runtime.delay_min(d_doa, geo, moving=moving)
        return time_in, runtime.synth_targets(tpl, "apply_to_template", doa=d_doa, geometry=geo, moving=moving, shift=shift)
    if moving:
        delays = np.stack([geometry.delays(d, normalized=False) for d in doa])  # [B, T, M]
        delays = delays - delays.min(axis=(1, 2), keepdims=True)                # :257: one global minimum per trial
        return time_in, runtime.synth_targets(tpl, "apply_to_template", delays=delays[:, None], moving=True)
    delays = geometry.delays(doa, normalized=False)
    delays = delays - delays.min(axis=1, keepdims=True)
    return time_in, runtime.synth_delay(time_in, sig_in, delays, fs, device=tpl.device)


def signal_from_template_batch(geometry, template, doas, device=None, device_delays=False):
    """signal_from_template for a batch: template = (time_temp, sig_temp) used as is (no resampling), doas [B] or [B, T].
    Returns x [B, T, M] device tensor."""
    time_temp, sig_temp = template
    time_temp = np.asarray(time_temp, dtype=np.float64)
    fs = (len(time_temp) - 1) / (time_temp[-1] - time_temp[0])
    tpl = runtime.Template(time_temp, sig_temp, fs, device=device)
    doa, moving = _doa_rows(doas, time_temp)
    if device_delays:
        geo = runtime.Geometry(geometry, device=tpl.device)
        d_doa = runtime._as_dev(np.ascontiguousarray(doa[:, None, :] if moving else doa[:, None]), tpl.device)
        return runtime.synth_targets(tpl, "signal_from_template", doa=d_doa, geometry=geo, moving=moving)
    if moving:
        delays = np.stack([geometry.delays(d, normalized=False) for d in doa])[:, None]  # [B, 1, T, M]
    else:
        delays = geometry.delays(doa, normalized=False)[:, None]                        # [B, 1, M]
    return runtime.synth_targets(tpl, "signal_from_template", delays=delays, moving=moving)


def signal_multiple_targets(geometry, time_temp, sig_temp, doa_timeseries_targets, power_timeseries_targets, device=None,
                            device_delays=False):
    """paper_plots/multiple_targets_snn.py:87-160 on the device.  doa / power time series: [T, K] (one trial, like the
    reference) or [B, T, K].  Returns x [T, M] (one trial) or [B, T, M] as a device tensor."""
    time_temp = np.asarray(time_temp, dtype=np.float64)
    T = len(time_temp)
    if T != len(sig_temp):
        raise ValueError("time vector and input signal should have the same dimensions!")
    doa = np.asarray(doa_timeseries_targets, dtype=np.float64)
    pw = np.asarray(power_timeseries_targets, dtype=np.float64)
    single = doa.ndim <= 2
    if doa.ndim == 1:
        doa = doa.reshape(-1, 1)
    if pw.ndim == 1:
        pw = pw.reshape(-1, 1)
    if single:
        doa, pw = doa[None], pw[None]
    if doa.shape[2] != pw.shape[2]:
        raise ValueError("number of targets should be the same in doa and power vector!")
    if doa.shape[1] != pw.shape[1] or doa.shape[1] != T:
        raise ValueError("input signal, doa, and power vectors should have the same dimension!")
    B, _, K = doa.shape
    fs = (T - 1) / (time_temp[-1] - time_temp[0])
    tpl = runtime.Template(time_temp, sig_temp, fs, device=device)
    doa_bkt = np.ascontiguousarray(np.transpose(doa, (0, 2, 1)))  # [B, K, T]
    gain = np.ascontiguousarray(np.transpose(pw, (0, 2, 1)))
    if device_delays:
        geo = runtime.Geometry(geometry, device=tpl.device)
        x = runtime.synth_targets(tpl, "signal_from_template", doa=doa_bkt, geometry=geo, moving=True, gain=gain)
    else:
        delays = geometry.delays(doa_bkt.reshape(-1), normalized=False).reshape(B, K, T, -1)
        x = runtime.synth_targets(tpl, "signal_from_template", delays=delays, moving=True, gain=gain)
    return x[0] if single else x


def add_noise_(x, snr_db, seed, first_trial=0, substream=0):
    """x [B, T, M] device tensor, in place: + sqrt(mean(x_b^2)) / sqrt(10^(snr_db_b / 10)) * N(0, 1)
    (snn_beamformer.py:270-275 with the device's Philox stream instead of np.random.randn)."""
    return runtime.awgn_(x, snr_db=snr_db, seed=seed, substream=substream, first_trial=first_trial)
