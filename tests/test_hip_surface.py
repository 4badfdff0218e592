"""GPU tests of the reference call surface (micloc.*) and of the Monte-Carlo sweep against golden data."""
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def make_beamformer(bipolar=True):
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    geo = CenterCircularArray(radius=4.5e-2, num_mic=7)
    tau = 1.0 / (2 * np.pi * 2000)
    return SNNBeamformer(geometry=geo, kernel_duration=10.0e-3, tau_vec=np.asarray([tau, tau]), freq_range=[1000.0, 2000.0], fs=48_000,
                         bipolar_spikes=bipolar)


def test_constructor_matches_reference_constants():
    k = golden("kat_init.npz")
    bf = make_beamformer()
    np.testing.assert_array_equal(bf.kernel, k["kernel_48k"])
    np.testing.assert_array_equal(bf.bandpass_filter[0], k["b_48k"])
    np.testing.assert_array_equal(bf.bandpass_filter[1], k["a_48k"])
    assert bf.kernel_length == 480 and bf.spk_encoder.robust_width == 12 and bf.spk_encoder.bipolar


def test_apply_to_signal_like_the_scripts(cfg2):
    z = golden("trials_cfg2.npz")
    bf = make_beamformer()
    y = bf.apply_to_signal(bf_mat=cfg2["bf_mat"], sig_in_vec=(z["time0"], z["sig_in"][0]))
    assert y.shape == (4799, 449) and y.dtype == np.float64
    np.testing.assert_allclose(y[z["row_idx"]], z["y_rows"][0], rtol=0, atol=1e-12)
    power = np.mean(np.abs(y) ** 2, axis=0)
    np.testing.assert_allclose(power, z["power"][0], rtol=1e-10)
    assert int(np.argmax(power)) == 83


def test_apply_to_template_reference_rng_order(cfg2):
    """np.random.seed(1234); doa = rand(1)*2pi; apply_to_template(...) -> the reference's trials."""
    z = golden("trials_cfg2.npz")
    bf = make_beamformer()
    fs = 48_000
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * 2000 * time_test)
    np.random.seed(1234)
    for i in range(3):
        doa = np.random.rand(1)[0] * 2 * np.pi
        assert doa == z["doa"][i]
        y = bf.apply_to_template(bf_mat=cfg2["bf_mat"], template=(time_test, sig_test, doa), snr_db=float(z["snr_db"]))
        power = np.mean(np.abs(y) ** 2, axis=0)
        assert int(np.argmax(power)) == int(z["argmax"][i])
        np.testing.assert_allclose(power, z["power"][i], rtol=1e-10)


def test_sweep_seed0_matches_reference(cfg2):
    """300 trials of the paper's sweep (3 SNRs x 100): identical arg-max and MAE to the reference."""
    from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep

    z = golden("sweep_seed0.npz")
    bf = make_beamformer()
    res = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], snr_db_vec=z["snr_db_vec"], num_sim=100, seed=int(z["seed"]), mode="parity")
    np.testing.assert_array_equal(res["doa"], z["doa"])
    np.testing.assert_array_equal(res["argmax"], z["argmax"])
    np.testing.assert_allclose(res["err"], z["err"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(res["mae_deg"], z["mae_deg"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(res["pmax"], z["pmax"], rtol=1e-10)


def test_synthesis_matches_reference(cfg2):
    from haghighatshoarmuir2024_amd.snn_beamformer import synthesize_array_signal

    z = golden("synth.npz")
    bf = make_beamformer()
    for name in ("fixed", "moving"):
        doa = z[f"{name}_doa"]
        doa = float(doa) if doa.ndim == 0 else doa
        t, sig = synthesize_array_signal(bf.geometry, 48_000, z["time_test"], z["sig_test"], doa)
        np.testing.assert_array_equal(t, z[f"{name}_time"])
        np.testing.assert_allclose(sig, z[f"{name}_sig"], rtol=0, atol=1e-100)


def test_design_from_template_bipolar_subset(cfg2):
    """Device chain + host SVD reproduces the reference's bf_mat columns (same LAPACK, inputs equal to ~1e-15)."""
    z = golden("bf_mat_chirp449_bipolar.npz")
    bf = make_beamformer()
    fs = 48_000
    time_temp = np.arange(0, 1.0, step=1 / fs)
    period = time_temp[-1]
    freq_inst = 1000 + 1000 * (time_temp % period) / period
    sig_temp = np.sin(2 * np.pi * np.cumsum(freq_inst) / fs)
    idx = z["cov_idx"][:6]
    W = bf.design_from_template((time_temp, sig_temp), z["doa_list"][idx], doa_batch=3)
    assert W.shape == (14, len(idx))
    ref = z["bf_mat"][:, idx]
    # columns are defined up to a unit complex phase of U[:,0]; compare after aligning the phase
    Wc, Rc = W[:7] + 1j * W[7:], ref[:7] + 1j * ref[7:]
    phase = np.sum(np.conj(Wc) * Rc, axis=0)
    phase /= np.abs(phase)
    np.testing.assert_allclose(Wc * phase, Rc, rtol=0, atol=1e-9)
    np.testing.assert_allclose(W, ref, rtol=0, atol=1e-8)  # and in practice LAPACK picks the same phase


def test_design_from_template_unipolar_subset():
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    z = golden("bf_mat_sin225_unipolar.npz")
    f, fs = 2000, 48_000
    tau = 1 / (2 * np.pi * f)
    bf = SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [0.5 * f, 2 * f], np.asarray([tau, tau]), bipolar_spikes=False, fs=fs)
    time_temp = np.arange(0, 0.4, step=1 / fs)
    idx = z["cov_idx"][:4]
    W = bf.design_from_template((time_temp, np.sin(2 * np.pi * f * time_temp)), z["doa_list"][idx])
    np.testing.assert_allclose(W, z["bf_mat_f2000"][:, idx], rtol=0, atol=1e-7)


def test_beamformer_class_surface(cfg2):
    from micloc.array_geometry import CenterCircularArray
    from micloc.beamformer import Beamformer

    z = golden("beamformer_c128.npz")
    bf = Beamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], fs=48_000)
    y = bf.apply_to_signal(z["bf_mat"], z["sig_in"])
    assert y.dtype == np.complex128 and y.shape == (4799, 57)
    np.testing.assert_allclose(y[z["row_idx"]], z["y_rows"], rtol=0, atol=1e-11)
    np.random.seed(99)
    doa = np.random.rand(1)[0] * 2 * np.pi
    fs = 48_000
    time_test = np.arange(0, 100e-3, step=1 / fs)
    y2 = bf.apply_to_template(z["bf_mat"], (time_test, np.sin(2 * np.pi * 2000 * time_test), doa), snr_db=3.0)
    np.testing.assert_allclose(np.mean(np.abs(y2) ** 2, axis=0), z["power"], rtol=1e-10)
    # design (no interference removal): covariances to 1e-12, bf_mat columns up to phase
    t = np.arange(0, 1.0, step=1 / fs)
    period = t[-1]
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % period) / period) / fs)
    W, covs = bf.design_from_template((t, s), z["doa_list"][:5])
    np.testing.assert_allclose(np.asarray(covs), z["cov_list"][:5], rtol=0, atol=1e-11)
    phase = np.sum(np.conj(W) * z["bf_mat"][:, :5], axis=0)
    np.testing.assert_allclose(W * (phase / np.abs(phase)), z["bf_mat"][:, :5], rtol=0, atol=1e-8)
    # the same design with the decompositions on the device (Jacobi on the real embedding of the Hermitian covariance): the
    # reference's columns up to their unit phase; the kernel's convention is "first component real and negative"
    Wd, covs_d = bf.design_from_template((t, s), z["doa_list"][:5], svd="device")
    np.testing.assert_allclose(np.asarray(covs_d), z["cov_list"][:5], rtol=0, atol=1e-11)
    phase = np.sum(np.conj(Wd) * z["bf_mat"][:, :5], axis=0)
    np.testing.assert_allclose(np.abs(phase), 1.0, rtol=0, atol=1e-9)
    np.testing.assert_allclose(Wd * (phase / np.abs(phase)), z["bf_mat"][:, :5], rtol=0, atol=1e-8)
    assert np.all(Wd[0].real < 0) and np.max(np.abs(Wd[0].imag)) < 1e-12
    with pytest.raises(ValueError):
        bf.design_from_template((t, s), z["doa_list"][:2], svd="device", interference_removal=True)


def test_beamformer_design_interference_removal():
    """Beamformer.design_from_template(interference_removal=True) (reference beamformer.py:152-190: generalised eigenvectors of
    (cov_g, sum_g' cov_g' + offset - cov_g)) against the reference's own matrix: STHT + complex covariance on the device
    (micloc_planar_gram_f64), scipy.linalg.eigh on the host like the reference.  A generalised eigenvector is defined up to a unit
    phase: columns are compared after aligning it, and through the beam pattern |W^H W|."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.beamformer import Beamformer

    z = golden("beamformer_c128.npz")
    fs = 48_000
    bf = Beamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], fs=fs)
    t = np.arange(0, 1.0, step=1 / fs)
    period = t[-1]
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % period) / period) / fs)
    doas = z["doa_list"][::4]
    W, covs = bf.design_from_template((t[:9600], s[:9600]), doas, interference_removal=True)
    ref = z["bf_mat_ir"]
    assert W.shape == ref.shape == (7, len(doas)) and W.dtype == np.complex128
    np.testing.assert_allclose(np.linalg.norm(W, axis=0), 1.0, rtol=0, atol=1e-12)
    phase = np.sum(np.conj(W) * ref, axis=0)
    assert np.all(np.abs(phase) > 1 - 1e-7)  # the same one-dimensional subspace
    np.testing.assert_allclose(W * (phase / np.abs(phase)), ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.abs(W.conj().T @ W), np.abs(ref.conj().T @ ref), rtol=0, atol=1e-6)
    # covariances are Hermitian PSD and equal to the NumPy form of the same STHT output
    for c in covs:
        np.testing.assert_allclose(c, c.conj().T, rtol=0, atol=1e-15)
        assert np.linalg.eigvalsh(c).min() > -1e-12


def test_device_synthesis_bit_exact(cfg2):
    """synthesize_batch (device) == synthesize_array_signal (host NumPy, itself == the reference's golden output)."""
    from haghighatshoarmuir2024_amd.snn_beamformer import synthesize_array_signal

    z = golden("synth.npz")
    bf = make_beamformer()
    rng = np.random.RandomState(3)
    doas = np.concatenate([[float(z["fixed_doa"])], rng.rand(40) * 2 * np.pi, [0.0, np.pi, -np.pi / 2]])
    t, x = bf.synthesize_batch((z["time_test"], z["sig_test"]), doas)
    x = x.cpu().numpy()
    np.testing.assert_array_equal(t, z["fixed_time"])
    np.testing.assert_allclose(x[0], z["fixed_sig"], rtol=0, atol=1e-100)
    for i, doa in enumerate(doas):
        _, ref = synthesize_array_signal(bf.geometry, 48_000, z["time_test"], z["sig_test"], float(doa))
        np.testing.assert_array_equal(x[i], ref)
    # a template that is NOT on the fs grid (resampled first) and the 0.1 s test tone of the sweep
    fs = 48_000
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * 2000 * time_test)
    t2, x2 = bf.synthesize_batch((time_test, sig_test), doas[:8])
    for i in range(8):
        _, ref = synthesize_array_signal(bf.geometry, fs, time_test, sig_test, float(doas[i]))
        np.testing.assert_array_equal(x2[i].cpu().numpy(), ref)


def test_throughput_sweep_statistics(cfg2):
    """Throughput mode (device synthesis + device noise): MAE-vs-SNR must look like the reference's curve
    (13.3 / 1.23 / 1.17 deg at -10 / +5 / +20 dB, grid step 0.8 deg)."""
    from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep

    bf = make_beamformer()
    res = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], snr_db_vec=[-10.0, 5.0, 20.0], num_sim=200, seed=11, mode="throughput")
    mae = res["mae_deg"]
    assert 6.0 < mae[0] < 25.0 and 0.5 < mae[1] < 2.5 and 0.5 < mae[2] < 2.0 and mae[0] > 3 * mae[1]


def test_full_sweep_1100_trials_matches_reference(cfg2):
    """BASELINE config 2 end to end: 11 SNRs x 100 trials on the reference's RNG stream -> identical arg-max for all
    1100 trials and the reference's MAE curve to the printed digits."""
    from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep

    z = golden("sweep_full_seed0.npz")
    bf = make_beamformer()
    res = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], num_sim=100, seed=0, mode="parity")
    np.testing.assert_array_equal(res["snr_db_vec"], z["snr_db_vec"])
    np.testing.assert_array_equal(res["doa"], z["doa"])
    np.testing.assert_array_equal(res["argmax"], z["argmax"])
    np.testing.assert_allclose(res["pmax"], z["pmax"], rtol=1e-10)
    np.testing.assert_allclose(res["mae_deg"], z["mae_deg"], rtol=0, atol=1e-9)
    # covariance-form variant: same decisions
    from haghighatshoarmuir2024_amd.sweep import device_localizer

    def cov_localizer(sig_batch, time_vec):
        out = bf.localize_batch(cfg2["bf_mat"], sig_batch, time_vec=time_vec, power_mode="covariance")
        a = out["argmax"].cpu().numpy().astype(np.int64)
        return a, out["power"].cpu().numpy()[np.arange(len(a)), a]

    res2 = noisy_target_sweep(bf, cfg2["bf_mat"], cfg2["doa_list"], snr_db_vec=z["snr_db_vec"][:3], num_sim=100, seed=0, mode="parity",
                              localizer=cov_localizer)
    np.testing.assert_array_equal(res2["argmax"], z["argmax"][:3])


def test_full_design_bipolar_449_and_config1_unipolar_225(cfg2):
    """design_from_template for the complete DoA grids of config 2 (449, bipolar, 1 s chirp) and config 1
    (array_resolution_snn.py: 225, unipolar, 0.4 s sine) against the reference's bf_mat, plus config 1's beam pattern."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    z = golden("bf_mat_chirp449_bipolar.npz")
    bf = make_beamformer()
    fs = 48_000
    time_temp = np.arange(0, 1.0, step=1 / fs)
    period = time_temp[-1]
    sig_temp = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (time_temp % period) / period) / fs)
    W = bf.design_from_template((time_temp, sig_temp), z["doa_list"])
    assert W.shape == (14, 449)
    Wc, Rc = W[:7] + 1j * W[7:], z["bf_mat"][:7] + 1j * z["bf_mat"][7:]
    phase = np.sum(np.conj(Wc) * Rc, axis=0)
    np.testing.assert_allclose(Wc * (phase / np.abs(phase)), Rc, rtol=0, atol=1e-8)  # up to the phase of U[:, 0]
    np.testing.assert_allclose(np.abs(W.T @ W), np.abs(z["bf_mat"].T @ z["bf_mat"]), rtol=0, atol=1e-7)

    zu = golden("bf_mat_sin225_unipolar.npz")
    for f in (1000, 2000, 4000):
        tau = 1 / (2 * np.pi * f)
        bfu = SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [0.5 * f, 2 * f], np.asarray([tau, tau]), bipolar_spikes=False, fs=fs)
        t = np.arange(0, 0.4, step=1 / fs)
        Wu = bfu.design_from_template((t, np.sin(2 * np.pi * f * t)), zu["doa_list"])
        ref = zu[f"bf_mat_f{f}"]
        np.testing.assert_allclose(Wu, ref, rtol=0, atol=2e-7)
        # the quantity config 1 plots (array_resolution_snn.py:157-160)
        np.testing.assert_allclose(np.abs(Wu.T @ Wu), np.abs(ref.T @ ref), rtol=0, atol=1e-6)


def test_full_design_on_device_jacobi(cfg2):
    """design_from_template(svd="device"): delayed templates, chain, covariance AND the decompositions on the device (batched
    Jacobi kernel + secular bisection, micloc_design_vectors_f64) for the complete grids of configs 2 and 1.  Bipolar columns
    agree with the reference's up to the (arbitrary) unit phase of U[:, 0] and in the beam pattern; unipolar columns directly."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    z = golden("bf_mat_chirp449_bipolar.npz")
    bf = make_beamformer()
    fs = 48_000
    time_temp = np.arange(0, 1.0, step=1 / fs)
    period = time_temp[-1]
    sig_temp = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (time_temp % period) / period) / fs)
    W = bf.design_from_template((time_temp, sig_temp), z["doa_list"], svd="device")
    assert W.shape == (14, 449)
    np.testing.assert_allclose(np.linalg.norm(W, axis=0), 1.0, rtol=0, atol=1e-12)
    Wc, Rc = W[:7] + 1j * W[7:], z["bf_mat"][:7] + 1j * z["bf_mat"][7:]
    phase = np.sum(np.conj(Wc) * Rc, axis=0)
    np.testing.assert_allclose(np.abs(phase), 1.0, rtol=0, atol=1e-8)
    np.testing.assert_allclose(Wc * (phase / np.abs(phase)), Rc, rtol=0, atol=1e-8)
    # phase convention of the kernel: the first component of every column is real and negative -- what LAPACK leaves on these
    # matrices to ~2e-4 of the modulus, so that the columns agree with the reference's DIRECTLY to that level (the
    # real-projected spectrum is not invariant to the phase; a convention that jumps along the grid moved half the arg-maxima)
    assert np.all(Wc[0].real < 0) and np.max(np.abs(Wc[0].imag)) < 1e-12
    assert np.max(np.abs(Wc - Rc)) < 2e-3
    # what the scripts plot does not depend on the phase: complex beam pattern |W^H W|
    np.testing.assert_allclose(np.abs(Wc.conj().T @ Wc), np.abs(Rc.conj().T @ Rc), rtol=0, atol=1e-7)
    # host-SVD and device-SVD designs localise alike: the golden trials identically, the reference's accuracy sweep (its RNG
    # stream, 1100 trials) with >= 93 % equal arg-maxima, p_max within 1e-3 and the MAE curve within 0.1 deg of the reference's
    zt = golden("trials_cfg2.npz")
    out_h = bf.localize_batch(z["bf_mat"], zt["sig_in"])
    out_d = bf.localize_batch(W, zt["sig_in"])
    ph, pd = out_h["power"].cpu().numpy(), out_d["power"].cpu().numpy()
    assert np.array_equal(np.argmax(ph, axis=1), np.argmax(pd, axis=1))
    np.testing.assert_allclose(pd, ph, rtol=2e-3)
    from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep

    zs = golden("sweep_full_seed0.npz")
    res = noisy_target_sweep(bf, W, z["doa_list"], num_sim=100, seed=0, mode="parity")
    assert np.mean(res["argmax"] == zs["argmax"]) >= 0.93
    np.testing.assert_allclose(res["pmax"], zs["pmax"], rtol=1e-3)
    assert np.max(np.abs(res["mae_deg"] - zs["mae_deg"])) < 0.1

    zu = golden("bf_mat_sin225_unipolar.npz")
    for f in (1000, 2000, 4000):
        tau = 1 / (2 * np.pi * f)
        bfu = SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [0.5 * f, 2 * f], np.asarray([tau, tau]), bipolar_spikes=False, fs=fs)
        t = np.arange(0, 0.4, step=1 / fs)
        Wu = bfu.design_from_template((t, np.sin(2 * np.pi * f * t)), zu["doa_list"], svd="device")
        ref = zu[f"bf_mat_f{f}"]
        sgn = np.sign(np.sum(Wu * ref, axis=0))
        assert np.all(sgn > 0)  # the conditional singular vector has no sign freedom
        np.testing.assert_allclose(Wu, ref, rtol=0, atol=2e-7)
        np.testing.assert_allclose(np.abs(Wu.T @ Wu), np.abs(ref.T @ ref), rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        bf.design_from_template((time_temp, sig_temp), z["doa_list"][:2], svd="gpu")


def test_design_vectors_wide_one_sided_jacobi():
    """micloc_design_vectors_f64 for 32 < C <= 128 (up to 64 microphones; one-sided Jacobi, csrc/design.hip) against LAPACK on
    random covariances: the DC-removed conditional singular vector (snn_beamformer.py:372-422) directly, the leading left singular
    vector of the complex fold (:191-203) up to its unit phase, whose convention (first component real and negative) is checked.
    Then a 32-microphone bipolar design end to end: svd="device" against svd="host"."""
    import torch

    from haghighatshoarmuir2024_amd import runtime
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    rng = np.random.RandomState(3)
    helper = make_beamformer()
    for C in (34, 64, 100, 128):
        n_doa = 5
        A = rng.randn(n_doa, C, 3 * C)
        cov = A @ A.transpose(0, 2, 1) / (3 * C) + 0.5 * np.exp(rng.randn(n_doa, 1, 1)) * (np.ones((C, C)) + np.eye(C))
        cov_d = torch.from_numpy(cov).cuda()
        for bipolar in (False, True):
            out = torch.zeros((C, n_doa + 2), dtype=torch.float64, device="cuda")
            runtime.design_vectors(cov_d, bipolar, out, 1, rel_prec=1e-8)
            W = out.cpu().numpy()
            assert np.all(W[:, 0] == 0) and np.all(W[:, -1] == 0)  # only columns g0 .. g0 + n are written
            W = W[:, 1:-1]
            np.testing.assert_allclose(np.linalg.norm(W, axis=0), 1.0, rtol=0, atol=1e-12)
            for i in range(n_doa):
                if not bipolar:
                    ref = helper._find_dc_removed_sing_vec(cov[i], rel_prec=1e-8)
                    np.testing.assert_allclose(W[:, i], ref, rtol=0, atol=1e-7)
                    assert abs(np.sum(W[:, i])) < 1e-4  # orthogonal to the all-one vector (to the bisection's rel_prec)
                else:
                    d = C // 2
                    Cc = (cov[i][:d, :d] + cov[i][d:, d:]) / 2 + 1j * ((cov[i][:d, d:] + cov[i][d:, :d].T) / 2)
                    U, S, _ = np.linalg.svd(Cc)
                    w = W[:d, i] + 1j * W[d:, i]
                    assert abs(abs(np.vdot(U[:, 0], w)) - 1.0) < 1e-9, (C, i, S[:3])
                    assert w[0].real < 0 and abs(w[0].imag) < 1e-12
    fs = 48_000
    geo = CenterCircularArray(8e-2, 32)
    tau = 1 / (2 * np.pi * 1500)
    bf = SNNBeamformer(geo, 10e-3, [1000, 2000], np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
    t = np.arange(0, 0.2, step=1 / fs)
    sig = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * t / t[-1]) / fs)
    doas = np.linspace(-np.pi, np.pi, 9)
    Wh = bf.design_from_template((t, sig), doas, svd="host", device_synthesis=True)
    Wd = bf.design_from_template((t, sig), doas, svd="device")
    assert Wd.shape == (64, 9)
    Wh, Wd = Wh[:32] + 1j * Wh[32:], Wd[:32] + 1j * Wd[32:]
    np.testing.assert_allclose(np.abs(np.sum(np.conj(Wh) * Wd, axis=0)), 1.0, rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.abs(Wd.conj().T @ Wd), np.abs(Wh.conj().T @ Wh), rtol=0, atol=1e-7)


def test_live_demo_frame_processing(cfg2):
    """localization_demo_snn.Demo.process_frame == filterbank -> apply_to_signal per band -> summed power -> arg-max,
    checked against the oracle composition; weak packs give NaN (reference :153-159)."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.localization_demo_snn import Demo
    from micloc.xylo_snn_localization import signal_from_template
    from oracle import oracle as O

    fs = 48_000
    geo = CenterCircularArray(4.5e-2, 7)
    doa_list = np.linspace(-np.pi, np.pi, 57)
    bands = [[1600, 2000], [2000, 2300]]
    demo = Demo(geometry=geo, freq_bands=bands, doa_list=doa_list, recording_duration=0.1, kernel_duration=10e-3, bipolar_spikes=True, fs=fs)
    t = np.arange(0, 0.1, 1 / fs)
    rng = np.random.RandomState(2)
    doa = 0.9
    sig = signal_from_template(geo, (t, np.sin(2 * np.pi * 1900 * t) + 0.7 * np.sin(2 * np.pi * 2150 * t), doa))
    sig = sig + 0.05 * rng.randn(*sig.shape)
    pack = np.zeros((len(t), 8), dtype=np.int32)
    pack[:, :7] = np.round(sig * 2**28).astype(np.int32)
    got = demo.process_frame(pack)
    # oracle composition
    data = pack[:, :-1].astype(np.float64)
    power = 0
    for (b1, a1), bf_mat, beamf in zip(demo.filterbank.ba_list, demo.bf_mats, demo.beamfs):
        filt = O.iir(b1, a1, data)
        bb, aa = beamf.bandpass_filter
        tau = beamf.tau_vec[0]
        nir = O.neuron_kernel(np.arange(len(t)) / fs, [tau, tau])
        out = O.snn_chain(filt, beamf.kernel, bb, aa, beamf.spk_encoder.robust_width, True, nir, bf_mat, want=("power",))
        power = power + out["power"]
    np.testing.assert_allclose(demo.power_grid(data), power, rtol=1e-12)
    assert got == doa_list[int(np.argmax(power))] * 180 / np.pi
    assert abs(np.degrees(np.arcsin(abs(np.sin(np.radians(got) - doa))))) < 12
    assert np.isnan(demo.process_frame(np.ones((4800, 8), dtype=np.int32)))
    seen = []
    demo.run([pack, np.ones((4800, 8), dtype=np.int32)], sink=seen.append)
    assert seen[0] == got and np.isnan(seen[1])


def test_pipeline_in_stages_equals_fused_call(cfg2):
    """micloc_snn_pipeline_stages_f64: the three kernel groups enqueued one call at a time (same plan, same outputs)
    give exactly what the fused call gives; invalid stage masks are refused."""
    import torch

    from conftest import golden
    from haghighatshoarmuir2024_amd import _lib
    from haghighatshoarmuir2024_amd.runtime import Plan

    z = golden("trials_cfg2.npz")
    p = Plan(7, cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True)
    p.set_neuron_kernel(cfg2["nir"])
    p.set_bf_mat(cfg2["bf_mat"])
    x = p.to_device(z["sig_in"])
    fused = p.snn_pipeline(x, want_spikes=True, want_power=True)
    fused = {k: (v.clone() if v is not None else None) for k, v in fused.items()}
    out = p.snn_pipeline(x, want_spikes=True, want_power=True, stages=1)
    out["spikes"].fill_(9)
    p.snn_pipeline(x, stages=2, out=out)
    assert torch.equal(out["spikes"], fused["spikes"])
    out["power"].fill_(-1.0)
    p.snn_pipeline(x, stages=4, out=out)
    assert torch.equal(out["power"], fused["power"]) and torch.equal(out["argmax"], fused["argmax"])
    np.testing.assert_array_equal(out["argmax"].cpu().numpy(), z["argmax"])
    for bad in (0, 32, -1):  # (8 and 16 are the two halves of stage 2: test_hip_chunked.py::test_scan_and_rest_in_two_calls)
        with pytest.raises(_lib.MiclocError):
            p.snn_pipeline(x, stages=bad, out=out)


def test_integration_md_ctypes_stub_runs(cfg2):
    """The ctypes binding INTEGRATION.md shows a reference maintainer (section B) is executed verbatim -- only the library
    path is made absolute -- and must return what the drop-in class returns."""
    import re

    from haghighatshoarmuir2024_amd import _lib

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "micloc/_hip.py" in b]
    assert len(stub) == 1
    code = stub[0].replace('ctypes.CDLL("libmicloc_hip.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    bf = make_beamformer()
    z = golden("trials_cfg2.npz")
    sig = z["sig_in"][0]
    plan = ns["make_plan"](bf, 7)
    y = ns["apply_to_signal"](plan, sig, cfg2["nir"], cfg2["bf_mat"])
    want = bf.apply_to_signal(cfg2["bf_mat"], (np.arange(sig.shape[0]) / 48_000, sig))
    assert y.shape == want.shape == (4799, 449)
    np.testing.assert_array_equal(y, want)
    with pytest.raises(ValueError):
        ns["apply_to_signal"](plan, sig, cfg2["nir"], cfg2["bf_mat"][:10])


def test_captured_graph_goes_stale_when_a_table_changes_shape(cfg2):
    """A captured hipGraph holds the tables' dimensions by value.  A bf_mat with fewer DoAs (or a shorter neuron kernel)
    fits the old allocation and is written in place: the plan's generation must change all the same and replay() must
    refuse, instead of reading the new table with the old stride.  Same-shape replacements keep the graph."""
    import torch

    from haghighatshoarmuir2024_amd import _lib, runtime

    z = golden("trials_cfg2.npz")
    plan = runtime.Plan(7, cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True)
    plan.set_neuron_kernel(cfg2["nir"])
    W = cfg2["bf_mat"]
    plan.set_bf_mat(W)
    x = plan.to_device(z["sig_in"])
    pipe = runtime.StreamPipeline([plan])
    replay = pipe.capture(lambda p: p.snn_pipeline(x, want_power=True))
    out = replay()
    pipe.synchronize()  # (the outputs belong to the pipeline's stream)
    ref = {k: v.clone() for k, v in out.items() if v is not None}
    # same shape, other values: in place, the graph stays valid and sees the new table
    gen0 = plan.generation
    plan.set_bf_mat(2.0 * W)
    assert plan.generation == gen0
    out = replay()
    pipe.synchronize()
    torch.testing.assert_close(out["power"], 4.0 * ref["power"], rtol=1e-12, atol=0)
    # fewer DoAs: fits the allocation, changes the layout
    plan.set_bf_mat(W[:, :200])
    assert plan.generation != gen0
    with pytest.raises(_lib.MiclocError, match="stale HIP graph"):
        replay()
    # a shorter neuron kernel likewise
    plan.set_bf_mat(W)
    replay2 = pipe.capture(lambda p: p.snn_pipeline(x, want_power=True))
    replay2()
    pipe.synchronize()
    gen1 = plan.generation
    plan.set_neuron_kernel(cfg2["nir"][:20])
    assert plan.generation != gen1
    with pytest.raises(_lib.MiclocError, match="stale HIP graph"):
        replay2()
