"""Sweep harnesses of BASELINE configs 3 and 4 (sweep.speech_target_sweep / xylo_target_sweep) on the GPU:
the speech sweep against the reference's own results (tests/golden/speech_sweep.npz), the Xylo sweep against the
oracle chain (integer-LIF stage parity-unpinned) including one full-size run (T = 48 000, 449 neurons, 28 inputs)."""
import numpy as np
import pytest

from conftest import golden
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _beamformer():
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    tau = 1 / (2 * np.pi * 2000)
    return SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=48_000)


def test_speech_sweep_matches_reference(cfg2):
    """paper_plots/target_snn_localization.py:213-245 (no bandwidth correction), 3 SNRs x 2 trials, the reference's RNG
    order: same DoAs, same arg-max, same error, p_max to 1e-10 -- with the FLAC-decoded source."""
    from haghighatshoarmuir2024_amd.sweep import speech_source, speech_target_sweep

    z = golden("speech_sweep.npz")
    pcm = golden("speech_trial.npz")
    src = speech_source(48_000, pcm16=pcm["pcm16"], rate=int(pcm["rate"]))
    assert len(src[0]) - 1 == int(z["T"])
    res = speech_target_sweep(_beamformer(), cfg2["bf_mat"], cfg2["doa_list"], src, snr_db_vec=z["snr_db_vec"], num_sim=int(z["num_sim"]),
                              seed=int(z["seed"]), mode="parity", batch_trials=4)
    np.testing.assert_array_equal(res["doa"], z["doa"])
    np.testing.assert_array_equal(res["argmax"], z["argmax"])
    np.testing.assert_allclose(res["err"], z["err"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(res["pmax"], z["pmax"], rtol=1e-10)
    np.testing.assert_allclose(res["mae_deg"], np.mean(z["err"], axis=1) * 180 / np.pi, rtol=0, atol=1e-9)


def _oracle_xylo_trial(demo, sig, win_size):
    """The same chain on the CPU: Demo.spike_encoding restated with the oracle's stages, oracle_xylo_lif, rate, peak."""
    from haghighatshoarmuir2024_amd.utils import find_peak_location

    bf = demo.beamfs[0]
    b, a = demo.filterbank.ba_list[0]
    re, im = O.stht(sig, bf.kernel)
    pre = O.iir(b, a, np.hstack([re, im]))
    spikes = O.rzcc(pre, bf.spk_encoder.robust_width, True)
    spikes_in = np.hstack([(spikes > 0), (spikes < 0)]).astype(np.uint8)
    out, rate = O.xylo_lif(spikes_in, demo.spec["W_in"], demo.spec["w_rec"], demo.spec["dash_syn"], demo.spec["dash_mem"], demo.spec["threshold"], 31)
    power = np.mean(out, axis=0) * demo.fs
    power = power.reshape(-1, len(demo.doa_list)).mean(0)
    mx = power.max()
    power = power / mx if mx > 0 else power
    return spikes_in, rate, int(find_peak_location(sig_in=power, win_size=win_size))


@pytest.mark.parametrize("full_size", [False, True])
def test_xylo_sweep_against_oracle_chain(full_size):
    """paper_plots/target_xylo_localization.py:566-608: signal_from_template -> noise -> spike_encoding -> xylo_process ->
    rate -> find_peak_location.  full_size: the script's own shape (1 s chirp = 48 000 frames, 449 hidden neurons, 28
    input channels); otherwise a short version with more trials."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.xylo_snn_localization import Demo, signal_from_template

    from haghighatshoarmuir2024_amd.sweep import xylo_target_sweep

    geo = CenterCircularArray(4.5e-2, 7)
    G = 64 * 7 + 1 if full_size else 8 * 7 + 1
    duration = 1.0 if full_size else 0.05
    doa_list = np.linspace(-np.pi, np.pi, G)
    demo = Demo(geometry=geo, freq_bands=[[1000, 2000]], doa_list=doa_list, recording_duration=0.1 if not full_size else 0.25, bipolar_spikes=True)
    snrs = [0.0, 20.0]
    num_sim = 1 if full_size else 3
    res = xylo_target_sweep(demo, snr_db_vec=snrs, num_sim=num_sim, seed=4, mode="parity", test_duration=duration, batch_trials=2)
    assert res["win_size"] == (15 if full_size else 1)
    assert res["index"].shape == (2, num_sim)
    # replay the reference's stream on the host and run the oracle chain
    fs = 48_000
    t = np.arange(0, duration, step=1 / fs)
    period = t[-1]
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % period) / period) * 1 / fs)
    gain = (fs / 2) / 1000.0
    np.random.seed(4)
    for i, snr_db in enumerate(snrs):
        for j in range(num_sim):
            doa = np.random.rand(1)[0] * 2 * np.pi
            sig = signal_from_template(geo, (t, s, doa))
            sig = sig + np.sqrt(np.mean(sig**2) / 10 ** ((snr_db - 10 * np.log10(gain)) / 10)) * np.random.randn(*sig.shape)
            assert res["doa"][i, j] == doa
            spikes_in, rate, idx = _oracle_xylo_trial(demo, sig, res["win_size"])
            assert spikes_in.shape == (len(t), 28)
            if i == 0 and j == 0:
                np.testing.assert_array_equal(demo.spike_encoding(sig), spikes_in)
            assert res["index"][i, j] == idx, (i, j)
            assert res["err"][i, j] == np.arcsin(np.abs(np.sin(doa_list[idx] - doa)))
    # throughput mode (device synthesis + Philox noise): runs, deterministic
    r1 = xylo_target_sweep(demo, snr_db_vec=snrs, num_sim=num_sim, seed=4, mode="throughput", test_duration=duration)
    r2 = xylo_target_sweep(demo, snr_db_vec=snrs, num_sim=num_sim, seed=4, mode="throughput", test_duration=duration, batch_trials=1)
    np.testing.assert_array_equal(r1["index"], r2["index"])
