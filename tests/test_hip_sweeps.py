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
    """The same chain on the CPU: Demo.spike_encoding restated with the oracle's stages (every band: order-1 band-pass -> RZCC; bands side
    by side; bipolar: [spikes > 0 | spikes < 0], unipolar: the 0 / 1 raster itself -- micloc/xylo_snn_localization.py:329-354),
    oracle_xylo_lif, rate averaged over the bands, peak."""
    from haghighatshoarmuir2024_amd.utils import find_peak_location

    bf = demo.beamfs[0]
    re, im = O.stht(sig, bf.kernel)
    spikes = np.hstack([O.rzcc(O.iir(b, a, np.hstack([re, im])), bf.spk_encoder.robust_width, demo.bipolar_spikes) for (b, a) in demo.filterbank.ba_list])
    spikes_in = (np.hstack([(spikes > 0), (spikes < 0)]) if demo.bipolar_spikes else (spikes > 0)).astype(np.uint8)
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


def test_xylo_unipolar_sweep_against_oracle_chain():
    """paper_plots/target_xylo_localization_unipolar.py (the same harness with bipolar_spikes=False, :61 / :146 / :420): unipolar RZCC
    (peaks only), the DC-removed design (snn_beamformer.py:372-422), 14 input channels instead of 28 -- Demo.counts_batch's unipolar
    branch (the int8 raster re-read as uint8 events).  Against the oracle chain on the reference's RNG stream."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.xylo_snn_localization import Demo, signal_from_template

    from haghighatshoarmuir2024_amd.sweep import xylo_target_sweep

    geo = CenterCircularArray(4.5e-2, 7)
    G, duration = 8 * 7 + 1, 0.1
    doa_list = np.linspace(-np.pi, np.pi, G)
    demo = Demo(geometry=geo, freq_bands=[[1000, 2000]], doa_list=doa_list, recording_duration=0.1, bipolar_spikes=False)
    assert demo.spec["W_in"].shape == (14, G)
    snrs = [5.0, 20.0]
    num_sim = 3
    res = xylo_target_sweep(demo, snr_db_vec=snrs, num_sim=num_sim, seed=11, mode="parity", test_duration=duration, batch_trials=4)
    dev = xylo_target_sweep(demo, snr_db_vec=snrs, num_sim=num_sim, seed=11, mode="parity", test_duration=duration, batch_trials=4, peak="device")
    fs = 48_000
    t = np.arange(0, duration, step=1 / fs)
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % t[-1]) / t[-1]) * 1 / fs)
    gain = (fs / 2) / 1000.0
    np.random.seed(11)
    total = 0
    for i, snr_db in enumerate(snrs):
        for j in range(num_sim):
            doa = np.random.rand(1)[0] * 2 * np.pi
            sig = signal_from_template(geo, (t, s, doa))
            sig = sig + np.sqrt(np.mean(sig**2) / 10 ** ((snr_db - 10 * np.log10(gain)) / 10)) * np.random.randn(*sig.shape)
            spikes_in, rate, idx = _oracle_xylo_trial(demo, sig, res["win_size"])
            assert spikes_in.shape == (len(t), 14) and spikes_in.max() == 1
            if j == 0:
                np.testing.assert_array_equal(demo.spike_encoding(sig), spikes_in)
                np.testing.assert_array_equal(demo.counts_batch(sig[None])[0].cpu().numpy(), rate)
                np.testing.assert_array_equal(demo.xylo_process(spikes_in).sum(axis=0), rate)
            assert res["index"][i, j] == idx and dev["index"][i, j] == idx, (i, j)
            assert res["err"][i, j] == np.arcsin(np.abs(np.sin(doa_list[idx] - doa)))
            total += int(rate.sum())
    assert total > 0


def test_xylo_two_band_demo_against_oracle_chain():
    """A multi-band Demo (micloc/xylo_snn_localization.py:150-153, 196-212: F bands -> block-diagonal weights, F * G hidden neurons,
    2 * F * 2M = 56 input channels; :379-398: the rate averaged over the bands before find_peak_location): spike_encoding, the counts,
    the band-averaged rate and the peak on the device against the oracle chain, plus the single-trial host methods."""
    import torch

    from micloc.array_geometry import CenterCircularArray
    from micloc.utils import find_peak_location
    from micloc.xylo_snn_localization import Demo, signal_from_template

    from haghighatshoarmuir2024_amd import runtime

    geo = CenterCircularArray(4.5e-2, 7)
    G = 8 * 7 + 1
    doa_list = np.linspace(-np.pi, np.pi, G)
    bands = [[1000, 1500], [1500, 2000]]
    demo = Demo(geometry=geo, freq_bands=bands, doa_list=doa_list, recording_duration=0.1, bipolar_spikes=True)
    W = demo.spec["W_in"]
    assert W.shape == (56, 2 * G) and len(demo.filterbank.ba_list) == 2
    # block-diagonal: band f's channels (+ block rows f*14 .., - block rows 28 + f*14 ..) only reach neurons f*G .. (f+1)*G
    assert not W[:14, G:].any() and not W[14:28, :G].any() and not W[28:42, G:].any() and not W[42:, :G].any()
    np.testing.assert_array_equal(W[28:], -W[:28])
    fs = 48_000
    t = np.arange(0, 0.1, step=1 / fs)
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % t[-1]) / t[-1]) / fs)
    rng = np.random.RandomState(3)
    doas = [0.7, -2.1, 2.9]
    sigs = []
    for doa in doas:
        sig = signal_from_template(geo, (t, s, doa))
        sigs.append(sig + np.sqrt(np.mean(sig**2) / 30) * rng.randn(*sig.shape))
    x = np.stack(sigs)
    counts = demo.counts_batch(x).cpu().numpy()
    rate_b = demo.rate_batch(x).cpu().numpy()
    win = 3
    idx_dev = demo.peak_batch(x, win).cpu().numpy()
    ev = demo.spike_encoding_device(x).cpu().numpy()
    assert counts.shape == (3, 2 * G) and rate_b.shape == (3, G) and ev.shape == (3, len(t), 56)
    # the queued and the one-workgroup-per-trial LIF launches agree on the two-k-step (56-channel) network too
    raster = demo.raster_device(x)
    assert raster.shape == (3, len(t), 28) and raster.dtype == torch.int8
    assert torch.equal(demo.network().run(raster, ternary=True, queued=True)[1], demo.network().run(raster, ternary=True, queued=False)[1])
    for i, sig in enumerate(sigs):
        spikes_in, rate, idx = _oracle_xylo_trial(demo, sig, win)
        np.testing.assert_array_equal(ev[i], spikes_in)
        np.testing.assert_array_equal(demo.spike_encoding(sig), spikes_in)
        np.testing.assert_array_equal(counts[i], rate)
        out = demo.xylo_process(spikes_in)
        assert out.shape == (len(t), 2 * G)
        np.testing.assert_array_equal(out.sum(axis=0), rate)
        r_host = demo.extract_rate(out)  # (mean over time) * fs, then the mean over the two bands
        np.testing.assert_allclose(rate_b[i], r_host, rtol=1e-13)
        np.testing.assert_allclose(r_host, (rate[:G] + rate[G:]) / 2 / len(t) * fs, rtol=1e-13)
        assert idx_dev[i] == idx == find_peak_location(r_host / r_host.max(), win_size=win)
    with pytest.raises(ValueError):
        runtime.peak_location(torch.zeros((1, 2 * G + 1), dtype=torch.int32, device="cuda"), G, win)


def test_sweeps_persist_and_resume_on_the_device(cfg2, tmp_path):
    """`out_dir=` on the device paths (sweep.ShardStore): the pipelined throughput sweep (several batches in flight: a batch is written
    when the event behind its device -> host copies has fired), the parity sweep and the Xylo sweep each leave one file per batch;
    with some of the files removed ("the job died there") a rerun recomputes exactly those batches and ends on the bits of the
    uninterrupted run.  ref:paper_plots/target_snn_localization.py:447-467, :525."""
    import os

    from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep, xylo_target_sweep

    bf = _beamformer()
    kw = dict(snr_db_vec=[-5.0, 5.0, 15.0], num_sim=20, seed=3, batch_trials=8)
    for mode in ("throughput", "parity"):
        ref = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], mode=mode, **kw)
        d = tmp_path / mode
        one = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], mode=mode, out_dir=d, **kw)
        assert one["persistence"]["files_written"] == 8 and one["persistence"]["trials_loaded"] == 0  # 60 trials in batches of 8
        sub = one["persistence"]["dir"]
        files = sorted(f for f in os.listdir(sub) if f.startswith("trials_"))
        assert len(files) == 8
        for f in (files[1], files[5], files[7]):
            os.remove(os.path.join(sub, f))
        two = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], mode=mode, out_dir=d, **kw)
        assert two["persistence"]["files_written"] == 3 and two["persistence"]["trials_loaded"] == 60 - 8 - 8 - 4
        three = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], mode=mode, out_dir=d, **kw)
        assert three["persistence"]["files_written"] == 0 and three["persistence"]["trials_loaded"] == 60
        for res in (one, two, three):
            for key in ("doa", "argmax", "pmax", "err", "mae_deg"):
                np.testing.assert_array_equal(res[key], ref[key], err_msg=f"{mode} {key}")
    # the Xylo sweep (integer-LIF stage parity-unpinned; what is checked here is the resume, against its own uninterrupted run)
    from micloc.array_geometry import CenterCircularArray
    from micloc.xylo_snn_localization import Demo

    demo = Demo(geometry=CenterCircularArray(4.5e-2, 7), freq_bands=[[1000.0, 2000.0]], doa_list=np.linspace(-np.pi, np.pi, 65), recording_duration=0.25,
                bipolar_spikes=True, fs=48_000)
    xkw = dict(snr_db_vec=[0.0, 20.0], num_sim=6, seed=4, mode="throughput", test_duration=50e-3, batch_trials=4)
    xref = xylo_target_sweep(demo, **xkw)
    x1 = xylo_target_sweep(demo, out_dir=tmp_path / "xylo", **xkw)
    files = sorted(f for f in os.listdir(x1["persistence"]["dir"]) if f.startswith("trials_"))
    assert len(files) == 3
    os.remove(os.path.join(x1["persistence"]["dir"], files[1]))
    x2 = xylo_target_sweep(demo, out_dir=tmp_path / "xylo", **xkw)
    assert x2["persistence"]["files_written"] == 1 and x2["persistence"]["trials_loaded"] == 8
    for res in (x1, x2):
        np.testing.assert_array_equal(res["index"], xref["index"])
        np.testing.assert_array_equal(res["doa"], xref["doa"])
