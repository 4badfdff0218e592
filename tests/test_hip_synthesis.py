"""Device synthesis (csrc/synth.hip) and device noise (csrc/rng.hip) against the reference's golden signals
(tests/golden/synth.npz, synth_xylo.npz: produced by the reference's own apply_to_template / signal_from_template /
signal_multiple_targets) and against the oracle's restatement of the Philox stream."""
import numpy as np
import pytest

from conftest import golden
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def geo():
    from micloc.array_geometry import CenterCircularArray

    return CenterCircularArray(4.5e-2, 7)


def test_apply_to_template_moving_doa_bit_exact(geo):
    """Array-valued DoA (snn_beamformer.py:239-267 with doa_temp a time series): host-delay path == reference bit for bit,
    device-delay path to an ulp of cos."""
    from haghighatshoarmuir2024_amd import synthesis

    z = golden("synth.npz")
    doas = np.stack([z["moving_doa"], z["moving_doa"][::-1].copy()])
    t, x = synthesis.apply_to_template_batch(geo, 48_000, (z["time_test"], z["sig_test"]), doas)
    np.testing.assert_array_equal(t, z["moving_time"])
    # (the golden signal carries the reference's noise at "3000 dB": 1e-150)
    np.testing.assert_allclose(x[0].cpu().numpy(), z["moving_sig"], rtol=0, atol=1e-100)
    np.testing.assert_array_equal(x[0].cpu().numpy(), O.synth_template(geo.r_vec, geo.theta_vec, z["time_test"], z["sig_test"], doas[0], 48_000)[1])
    want1 = O.synth_template(geo.r_vec, geo.theta_vec, z["time_test"], z["sig_test"], doas[1], 48_000)[1]
    np.testing.assert_array_equal(x[1].cpu().numpy(), want1)
    _, xd = synthesis.apply_to_template_batch(geo, 48_000, (z["time_test"], z["sig_test"]), doas, device_delays=True)
    np.testing.assert_allclose(xd.cpu().numpy(), x.cpu().numpy(), rtol=0, atol=1e-12)
    # constant DoA through the general kernel (device delays) against the bit-exact fast path
    _, xc = synthesis.apply_to_template_batch(geo, 48_000, (z["time_test"], z["sig_test"]), np.array([1.2345, 4.0]))
    np.testing.assert_allclose(xc[0].cpu().numpy(), z["fixed_sig"], rtol=0, atol=1e-100)
    _, xcd = synthesis.apply_to_template_batch(geo, 48_000, (z["time_test"], z["sig_test"]), np.array([1.2345, 4.0]), device_delays=True)
    np.testing.assert_allclose(xcd.cpu().numpy(), xc.cpu().numpy(), rtol=0, atol=1e-12)


def test_signal_from_template_and_multiple_targets_bit_exact(geo):
    from haghighatshoarmuir2024_amd import synthesis

    z = golden("synth_xylo.npz")
    t, s = z["time"], z["sig"]
    x = synthesis.signal_from_template_batch(geo, (t, s), np.array([float(z["fixed_doa"]), 0.1]))
    np.testing.assert_array_equal(x[0].cpu().numpy(), z["fixed_sig"])
    np.testing.assert_array_equal(x[1].cpu().numpy(), O.signal_from_template(geo.r_vec, geo.theta_vec, t, s, 0.1))
    xm = synthesis.signal_from_template_batch(geo, (t, s), z["moving_doa"][None])
    np.testing.assert_array_equal(xm[0].cpu().numpy(), z["moving_sig"])
    xmd = synthesis.signal_from_template_batch(geo, (t, s), z["moving_doa"][None], device_delays=True)
    np.testing.assert_allclose(xmd.cpu().numpy(), xm.cpu().numpy(), rtol=0, atol=1e-12)
    # three targets with time-varying DoA and power (paper_plots/multiple_targets_snn.py:87-160)
    y = synthesis.signal_multiple_targets(geo, t, s, z["multi_doa"], z["multi_power"])
    np.testing.assert_array_equal(y.cpu().numpy(), z["multi_sig"])
    yb = synthesis.signal_multiple_targets(geo, t, s, np.stack([z["multi_doa"], z["multi_doa"][:, ::-1]]), np.stack([z["multi_power"]] * 2))
    np.testing.assert_array_equal(yb[0].cpu().numpy(), z["multi_sig"])
    np.testing.assert_array_equal(yb[1].cpu().numpy(), O.signal_multiple_targets(geo.r_vec, geo.theta_vec, t, s, z["multi_doa"][:, ::-1], z["multi_power"]))
    yd = synthesis.signal_multiple_targets(geo, t, s, z["multi_doa"], z["multi_power"], device_delays=True)
    np.testing.assert_allclose(yd.cpu().numpy(), z["multi_sig"], rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        synthesis.signal_multiple_targets(geo, t, s, z["multi_doa"], z["multi_power"][:, :2])
    with pytest.raises(ValueError):
        synthesis.signal_multiple_targets(geo, t, s[:-1], z["multi_doa"], z["multi_power"])


def test_uniform_bit_exact_and_awgn_against_oracle():
    import torch

    from haghighatshoarmuir2024_amd import runtime

    seed = (0x1234 << 32) | 0xBEEF
    for n in (1, 2, 7, 1000, 100_001):
        u = runtime.uniform(n, seed, substream=2, lo=0.0, hi=2 * np.pi).cpu().numpy()
        np.testing.assert_array_equal(u, O.uniform(n, seed, 2, 0.0, 2 * np.pi))
    # the epoch word (a device counter a captured graph advances): its own counter word, equal to the oracle's
    ep = torch.full((1,), 5, dtype=torch.int32, device="cuda")
    u5 = runtime.uniform(1000, seed, substream=2, lo=0.0, hi=1.0, epoch=ep).cpu().numpy()
    np.testing.assert_array_equal(u5, O.uniform(1000, seed, 2, 0.0, 1.0, epoch=5))
    assert not np.array_equal(u5, O.uniform(1000, seed, 7, 0.0, 1.0))
    rng = np.random.RandomState(1)
    for shape in ((3, 4799, 7), (2, 33, 3), (1, 1, 1), (2, 8192 // 7 + 5, 7)):
        x = rng.randn(*shape) * np.linspace(0.5, 3.0, shape[0])[:, None, None]
        snr = np.linspace(-10, 20, shape[0])
        want, sigma = O.awgn(x, snr, seed=seed, substream=1, first_trial=40)
        xd = torch.from_numpy(x).cuda()
        runtime.awgn_(xd, snr_db=snr, seed=seed, substream=1, first_trial=40)
        got = xd.cpu().numpy()
        # same integers, same mapping; libm vs device log / sin / cos and the order of the power sum differ in the last bits
        np.testing.assert_allclose(got - x, want - x, rtol=0, atol=1e-12 * sigma.max())
        # numbered by global trial: a later slice of the batch draws the same noise
        if shape[0] > 1:
            xs = torch.from_numpy(x[1:]).cuda()
            runtime.awgn_(xs, snr_db=snr[1:], seed=seed, substream=1, first_trial=41)
            np.testing.assert_array_equal(xs.cpu().numpy(), got[1:])
    x1 = rng.randn(2, 501, 3)
    want1, sg1 = O.awgn(x1, [3.0, 4.0], seed=seed, substream=1, first_trial=9, epoch=6)
    xd1 = torch.from_numpy(x1).cuda()
    runtime.awgn_(xd1, snr_db=np.array([3.0, 4.0]), seed=seed, substream=1, first_trial=9, epoch=torch.full((1,), 6, dtype=torch.int32, device="cuda"))
    np.testing.assert_allclose(xd1.cpu().numpy() - x1, want1 - x1, rtol=0, atol=1e-12 * sg1.max())
    # explicit sigma, statistics of a large draw
    xz = torch.zeros((4, 100_000, 7), dtype=torch.float64, device="cuda")
    runtime.awgn_(xz, sigma=np.array([1.0, 2.0, 0.5, 1.0]), seed=7)
    z = xz.cpu().numpy()
    for b, sg in enumerate([1.0, 2.0, 0.5, 1.0]):
        assert abs(z[b].mean()) < 0.01 * sg and abs(z[b].std() / sg - 1) < 0.01
    assert abs(np.corrcoef(z[0].ravel(), z[3].ravel())[0, 1]) < 0.01


def test_throughput_sweep_is_sharding_invariant(cfg2, geo):
    """Throughput mode (device synthesis + Philox noise numbered by global trial): identical results whatever the
    sharding, and the MAE curve falls with the SNR like the reference's."""
    from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer
    from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep

    tau = 1 / (2 * np.pi * 2000)
    bf = SNNBeamformer(geo, 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=48_000)
    kw = dict(snr_db_vec=[-10.0, 5.0, 20.0], num_sim=20, seed=3, mode="throughput")
    full = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], **kw)
    res = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], **kw, batch_trials=7)
    np.testing.assert_array_equal(res["argmax"], full["argmax"])  # batch split 7 vs 1100: same draws, same results
    np.testing.assert_array_equal(res["pmax"], full["pmax"])
    assert full["mae_deg"][0] > full["mae_deg"][2] and full["mae_deg"][2] < 3.0
    # the default localizer keeps several batches in flight (sweep._throughput_pipelined); streams=0 is one batch at a time through
    # synthesize_batch / add_noise_ / localize_batch: the same draws, the same bits
    serial = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], **kw, batch_trials=7, streams=0)
    np.testing.assert_array_equal(serial["argmax"], full["argmax"])
    np.testing.assert_array_equal(serial["pmax"], full["pmax"])
    piped = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], **kw, batch_trials=9, streams=3)  # 60 trials: 6 batches of 9 and one of 6
    np.testing.assert_array_equal(piped["argmax"], full["argmax"])
    np.testing.assert_array_equal(piped["pmax"], full["pmax"])


def test_long_recording_sweep_pipelined_on_the_scan_lane(cfg2, geo):
    """speech_target_sweep in throughput mode on a long source (the encoder is time-chunked): batches in flight on streams restricted
    to compute units [4, 32) of every XCD, their serial scans on the lane -- identical to one batch at a time."""
    from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer
    from haghighatshoarmuir2024_amd.sweep import speech_target_sweep

    fs = 48_000
    tau = 1 / (2 * np.pi * 2000)
    bf = SNNBeamformer(geo, 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
    T = 70_000
    t = np.arange(T) / fs
    rng = np.random.RandomState(5)
    src = np.sin(2 * np.pi * 1500 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.05 * rng.randn(T)
    kw = dict(snr_db_vec=[0.0, 20.0], num_sim=11, seed=9, mode="throughput", batch_trials=5)
    assert bf.plan().encoder_chunks(5, len(np.arange(t.min(), t.max(), step=1 / fs))) > 1
    serial = speech_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], (t, src), **kw, streams=0)
    piped = speech_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], (t, src), **kw, streams=3)  # 22 trials: 4 batches of 5 and one of 2
    np.testing.assert_array_equal(piped["argmax"], serial["argmax"])
    np.testing.assert_array_equal(piped["pmax"], serial["pmax"])
    assert piped["mae_deg"][1] < 5.0


def test_fused_synthesis_and_noise_equals_the_two_calls(geo):
    """micloc_synth_awgn_f64 (two passes that never store the noise-free signal) == micloc_synth_targets_f64 followed by
    micloc_awgn_f64, bit for bit: constant and moving DoAs, several targets with gains, both delay sources and conventions,
    ragged block ends, an epoch word."""
    import torch

    from haghighatshoarmuir2024_amd import runtime

    fs = 48_000
    rng = np.random.RandomState(11)
    for T, B in ((4799, 5), (1171, 3), (8192 // 7 + 1, 2), (333, 1)):
        t = np.arange(T) / fs
        tpl = runtime.Template(t, np.sin(2 * np.pi * 1500 * t) + 0.1 * rng.randn(T), fs)
        g = runtime.Geometry(geo)
        snr = np.linspace(-5, 15, B)
        ep = torch.full((1,), 3, dtype=torch.int32, device="cuda")
        cases = []
        doa = rng.rand(B, 1) * 2 * np.pi
        cases.append(dict(mode="apply_to_template", doa=doa, geometry=g, shift=runtime.delay_min(torch.from_numpy(doa).cuda(), g)))
        cases.append(dict(mode="signal_from_template", doa=doa, geometry=g))
        cases.append(dict(mode="apply_to_template", delays=geo.delays(doa[:, 0], normalized=False)[:, None, :], shift=np.zeros(B)))
        mv = rng.rand(B, 2, T) * 2 * np.pi
        cases.append(dict(mode="signal_from_template", doa=mv, geometry=g, moving=True, gain=rng.rand(B, 2, T)))
        for kw in cases:
            want = runtime.synth_targets(tpl, **kw)
            runtime.awgn_(want, snr_db=snr, seed=99, substream=4, first_trial=17, epoch=ep)
            got = runtime.synth_awgn(tpl, snr_db=snr, seed=99, substream=4, first_trial=17, epoch=ep, **kw)
            assert torch.equal(got, want), (T, B, kw["mode"], sorted(kw))


def test_fused_synthesis_other_shapes():
    """The fused kernels' wave-per-microphone order (csrc/rng.hip TimeMap) on shapes the sweep does not have: fewer microphones
    than waves, more than a wave, more than a workgroup; recordings shorter than one 64-row chunk; an array so wide that the
    template window does not fit LDS; a jittered (non-uniform) time grid, where the constant row offset does not hold and every
    sample is searched; two constant-DoA targets with gains.  Always the bits of synth_targets + awgn_."""
    import torch

    from haghighatshoarmuir2024_amd import runtime
    from micloc.array_geometry import CenterCircularArray

    fs = 48_000
    rng = np.random.RandomState(5)
    ep = torch.full((1,), 9, dtype=torch.int32, device="cuda")
    shapes = [(4.5e-2, 1, 901, 2), (4.5e-2, 2, 2500, 3), (4.5e-2, 3, 4799, 2), (4.5e-2, 64, 700, 2), (4.5e-2, 300, 300, 2), (4.5e-2, 7, 5, 3), (4.5e-2, 7, 63, 2),
              (3.0, 7, 3000, 2), (0.5, 16, 2100, 2)]
    for radius, M, T, B in shapes:
        geo = CenterCircularArray(radius, M)
        g = runtime.Geometry(geo)
        snr = np.linspace(0, 10, B)
        doa = rng.rand(B, 1) * 2 * np.pi
        for jitter in (0.0, 0.3):
            t = (np.arange(T) + jitter * (rng.rand(T) - 0.5)) / fs
            tpl = runtime.Template(t, np.sin(2 * np.pi * 900 * t) + 0.1 * rng.randn(T), fs)
            cases = [dict(mode="apply_to_template", doa=doa, geometry=g, shift=runtime.delay_min(torch.from_numpy(doa).cuda(), g)),
                     dict(mode="signal_from_template", doa=doa, geometry=g)]
            if M <= 64:
                doa2 = rng.rand(B, 2) * 2 * np.pi
                cases.append(dict(mode="signal_from_template", doa=doa2, geometry=g, gain=rng.rand(B, 2, T)))
            for kw in cases:
                want = runtime.synth_targets(tpl, **kw)
                runtime.awgn_(want, snr_db=snr, seed=5, substream=2, first_trial=3, epoch=ep)
                got = runtime.synth_awgn(tpl, snr_db=snr, seed=5, substream=2, first_trial=3, epoch=ep, **kw)
                assert torch.equal(got, want), (radius, M, T, B, jitter, kw["mode"], sorted(kw))
