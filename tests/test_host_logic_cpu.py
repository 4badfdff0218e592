"""CPU-only tests of the host-side logic that mirrors the reference's class surface (no kernels involved)."""
import numpy as np
import pytest

from conftest import golden


def test_geometries_match_reference():
    from micloc.array_geometry import ArrayGeometry, CenterCircularArray, CircularArray, LinearArray, Random2DArray

    k = golden("kat_init.npz")
    g = CenterCircularArray(4.5e-2, 7)
    np.testing.assert_array_equal(g.r_vec, k["ccirc_r"])
    np.testing.assert_array_equal(g.theta_vec, k["ccirc_theta"])
    assert len(g) == 7 and g.speed == 340
    g2 = CircularArray(4.5e-2, 7)
    np.testing.assert_array_equal(g2.r_vec, k["circ_r"])
    np.testing.assert_array_equal(g2.theta_vec, k["circ_theta"])
    lin = LinearArray(spacing=0.02, num_mic=8, radius=0.07)
    np.testing.assert_array_equal(lin.r_vec, k["lin_r"])
    np.testing.assert_array_equal(lin.theta_vec, k["lin_theta"])
    assert lin.radius == 0.07
    np.random.seed(1)
    rnd = Random2DArray(radius=0.2, num_mic=16)
    np.testing.assert_array_equal(rnd.r_vec, k["rand_r"])
    np.testing.assert_array_equal(rnd.theta_vec, k["rand_theta"])
    # BASELINE config 5's array: 64 microphones after np.random.seed(1) (stress_case.npz, from the reference), and the host
    # synthesis at that geometry against the reference's noise-free rows
    zs = golden("stress_case.npz")
    np.random.seed(int(zs["geometry_seed"]))
    rnd64 = Random2DArray(radius=0.2, num_mic=64)
    np.testing.assert_array_equal(rnd64.r_vec, zs["r_vec"])
    np.testing.assert_array_equal(rnd64.theta_vec, zs["theta_vec"])
    from haghighatshoarmuir2024_amd.snn_beamformer import synthesize_array_signal

    fs = int(zs["fs"])
    tt = np.arange(0, 100e-3, step=1 / fs)
    t64, x64 = synthesize_array_signal(rnd64, fs, tt, np.sin(2 * np.pi * 2000 * tt), float(zs["doa"]))
    np.testing.assert_array_equal(t64, zs["time_vec"])
    np.testing.assert_allclose(x64[zs["clean_idx"]], zs["clean_rows"], rtol=0, atol=1e-100)
    for i, th in enumerate(k["thetas"]):
        np.testing.assert_array_equal(g.delays(th, normalized=True), k["ccirc_delays_norm"][i])
        np.testing.assert_array_equal(g.delays(th, normalized=False), k["ccirc_delays_raw"][i])
        np.testing.assert_array_equal(lin.delays(th), k["lin_delays_norm"][i])
    # vectorised call == row-wise scalar calls (this is what replaces the reference's T-call list comprehension)
    np.testing.assert_array_equal(g.delays(k["thetas"], normalized=False), k["ccirc_delays_raw"])
    np.testing.assert_array_equal(g.delays(k["thetas"], normalized=True), k["ccirc_delays_norm"])
    with pytest.raises(ValueError):
        ArrayGeometry(np.array([1.0, -1.0]), np.zeros(2))


def test_utils_match_reference():
    from micloc.utils import Envelope, find_peak_location

    k = golden("kat_init.npz")
    p = k["fpl_in"]
    assert [find_peak_location(p, 15), find_peak_location(p, 1), find_peak_location(p, 15, periodic=False)] == list(k["fpl_out"])
    with pytest.raises(ValueError):
        find_peak_location(p.reshape(1, -1), 15)
    with pytest.raises(ValueError):
        find_peak_location(p, 14)
    with pytest.raises(ValueError):
        find_peak_location(p[:20], 15)
    env = Envelope(rise_time=1e-3, fall_time=20e-3, fs=48_000)
    np.testing.assert_allclose(env.evolve(k["env_in"]), k["env_out"], rtol=1e-15, atol=0)
    with pytest.raises(ValueError):
        Envelope(rise_time=1.0, fall_time=0.5, fs=48_000)


def test_beamformer_constructors_and_errors():
    from micloc.array_geometry import CenterCircularArray
    from micloc.beamformer import Beamformer
    from micloc.snn_beamformer import Fs, SNNBeamformer
    from micloc.spike_encoder import SpikeEncoder, ZeroCrossingSpikeEncoder

    assert Fs == 48_000
    k = golden("kat_init.npz")
    geo = CenterCircularArray(4.5e-2, 7)
    tau = 1 / (2 * np.pi * 2000)
    for tag, fs, fr in [("48k", 48_000, [1000.0, 2000.0]), ("96k", 96_000, [1000.0, 2000.0]), ("48k_4k", 48_000, [2000.0, 4000.0])]:
        bf = SNNBeamformer(geo, 10e-3, fr, np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
        np.testing.assert_array_equal(bf.kernel, k[f"kernel_{tag}"])
        np.testing.assert_array_equal(bf.bandpass_filter[0], k[f"b_{tag}"])
        np.testing.assert_array_equal(bf.bandpass_filter[1], k[f"a_{tag}"])
        assert bf.kernel_length == int(k[f"kernel_length_{tag}"])
        assert bf.spk_encoder.robust_width == int(k[f"robust_width_{tag}"]) and bf.spk_encoder.bipolar and bf.spk_encoder.fs == fs
        assert isinstance(bf.spk_encoder, ZeroCrossingSpikeEncoder) and bf.bipolar_spikes and bf.geometry is geo
        b2 = Beamformer(geo, 10e-3, fr, fs=fs)
        np.testing.assert_array_equal(b2.kernel, k[f"kernel_{tag}"])
        np.testing.assert_array_equal(b2.bandpass_filter[1], k[f"a_{tag}"])
    with pytest.raises(ValueError):
        SNNBeamformer(geo, 10e-3, [2000, 1000], [tau, tau])
    with pytest.raises(ValueError):
        SNNBeamformer(geo, 10e-3, 1500.0, [tau, tau])
    with pytest.raises(ValueError):
        Beamformer(geo, 10e-3, [2000, 1000])
    bf = SNNBeamformer(geo, 10e-3, [1000, 2000], [tau, tau])
    with pytest.raises(ValueError):
        bf.apply_to_signal(np.zeros((14, 5)), (np.arange(50) / 48000, np.zeros((50, 6))))
    with pytest.raises(ValueError):
        bf.apply_to_template(np.zeros((14, 5)), (np.arange(10), np.arange(10)), 3.0)
    with pytest.raises(ValueError):
        bf.design_from_template((np.arange(10),), np.zeros(3))
    with pytest.raises(NotImplementedError):
        SpikeEncoder().evolve(np.zeros((4, 2)))


def test_neuron_kernel_and_synthesis():
    from haghighatshoarmuir2024_amd.snn_beamformer import neuron_impulse_response, synthesize_array_signal
    from micloc.array_geometry import CenterCircularArray

    k = golden("kat_init.npz")
    for tag, fs, f_hi in [("48k", 48_000, 2000.0), ("96k", 96_000, 2000.0), ("48k_4k", 48_000, 4000.0)]:
        tau = 1 / (2 * np.pi * f_hi)
        T = int(k[f"nir_T_{tag}"])
        np.testing.assert_array_equal(neuron_impulse_response(np.arange(T) / fs, [tau, tau]), k[f"nir_{tag}"])
    with pytest.raises(ValueError):
        neuron_impulse_response(np.arange(10) / 48000, [1e-4, 2e-4])
    z = golden("synth.npz")
    geo = CenterCircularArray(4.5e-2, 7)
    for name in ("fixed", "moving"):
        doa = z[f"{name}_doa"]
        doa = float(doa) if doa.ndim == 0 else doa
        t, sig = synthesize_array_signal(geo, 48_000, z["time_test"], z["sig_test"], doa)
        np.testing.assert_array_equal(t, z[f"{name}_time"])
        np.testing.assert_allclose(sig, z[f"{name}_sig"], rtol=0, atol=1e-100)


def test_iaf_encoders_and_peak_encoder_shapes():
    from micloc.spike_encoder import IAFSpikeEncoder, IAFZeroCrossingSpikeEncoder, PeakSpikeEncoder

    rng = np.random.RandomState(0)
    x = rng.randn(200, 3)
    s = IAFSpikeEncoder(target_spike_rate=2000, fs=48_000).evolve(x)
    assert s.shape == (199, 3) and s.min() >= 0
    s2 = IAFZeroCrossingSpikeEncoder(target_spike_rate=2000, fs=48_000).evolve(x)
    assert s2.shape == (199, 3)
    s3 = PeakSpikeEncoder(48_000).evolve(x, robust_width=3)
    assert s3.shape == x.shape and set(np.unique(s3)) <= {0.0, 1.0}


def test_shard_range_and_doa_error():
    from haghighatshoarmuir2024_amd.sweep import doa_error, shard_range

    for total in (0, 1, 7, 1100, 1001):
        for world in (1, 2, 3, 8):
            got = [shard_range(total, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1
    assert abs(doa_error(0.1 + np.pi, 0.1)) < 1e-12  # pi-periodic metric
    assert abs(doa_error(0.3, 0.1) - 0.2) < 1e-12
