"""GPU twins of the round-5 reference fixtures: BASELINE config 5 at its real shape (64 random
microphones, 96 kHz, 1440 DoAs), the complex Beamformer at G = 449, and config 1's designs at 3.6 kHz and 8 kHz.  Everything is
compared with what the REAL reference produced (tests/golden/make_golden.py), through the C-ABI (Plan) and through the drop-in
class surface (micloc.*).  Bars: spikes and arg-max identical, power 1e-10 relative, y rows 1e-11 / 1e-12 absolute, design
columns 2e-7 (unipolar: no sign freedom) or up to the singular vector's unit phase (complex)."""
import numpy as np
import pytest

from conftest import golden
from oracle import oracle as O
from test_oracle_golden import stress_case_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def test_stress_case_plan_vs_reference(torch):
    """micloc_snn_pipeline_f64 on the reference's own config-5 trial: the 960-tap walking STHT, robust width 24, the 71-tap neuron
    kernel and beamform_gen_kernel at 128 channels x 1440 DoAs (power only, and with y stored)."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    s = stress_case_inputs()
    z, x = s["z"], s["x"]
    p = Plan(64, s["ker"], s["b"], s["a"], s["w"], True)
    p.set_neuron_kernel(s["nir"])
    p.set_bf_mat(s["W"])
    out = p.snn_pipeline(p.to_device(x[None]), want_spikes=True, want_power=True)
    np.testing.assert_array_equal(out["spikes"][0].cpu().numpy(), z["spikes"])
    power = out["power"][0].cpu().numpy()
    np.testing.assert_allclose(power, z["power"], rtol=1e-10, atol=0)
    assert int(out["argmax"][0]) == int(z["argmax"]) == int(np.argmax(power))
    out = p.snn_pipeline(p.to_device(x[None]), want_spikes=True, want_y=True, want_power=True)
    y = out["y"][0].cpu().numpy()
    assert y.shape == (9599, 1440)
    np.testing.assert_allclose(y[z["row_idx"]], z["y_rows"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["power"][0].cpu().numpy(), z["power"], rtol=1e-10, atol=0)
    np.testing.assert_allclose(np.mean(y * y, axis=0), z["power"], rtol=1e-10, atol=0)
    # pre-encoder rows (stage API: STHT + band-pass) against the reference's, the stage's spikes against the fused call's
    T = x.shape[0]
    pre, spikes = p.bandpass_rzcc(p.stht(p.to_device(x[None])), T)
    rows = np.ascontiguousarray(pre[:, :, :T].cpu().numpy().transpose(0, 2, 1))[0]
    np.testing.assert_allclose(rows[z["pre_idx"]], z["pre_enc_rows"], rtol=0, atol=1e-11)
    np.testing.assert_array_equal(spikes[0].cpu().numpy(), z["spikes"])


def test_stress_case_class_surface(torch):
    """The same trial the way a script would run it: Random2DArray after np.random.seed(1) (the reference's draw order,
    ref:micloc/array_geometry.py:126-127), SNNBeamformer at 96 kHz, apply_to_signal -> T x G; and the noise-free synthesis for that
    geometry on the host surface and on the device."""
    from micloc.array_geometry import Random2DArray
    from micloc.snn_beamformer import SNNBeamformer

    s = stress_case_inputs()
    z, x = s["z"], s["x"]
    np.random.seed(int(z["geometry_seed"]))
    geo = Random2DArray(radius=0.2, num_mic=64)
    np.testing.assert_array_equal(geo.r_vec, z["r_vec"])
    np.testing.assert_array_equal(geo.theta_vec, z["theta_vec"])
    fs = s["fs"]
    tau = 1 / (2 * np.pi * 2000.0)
    bf = SNNBeamformer(geo, 10e-3, [1000.0, 2000.0], [tau, tau], bipolar_spikes=True, fs=fs)
    assert bf.kernel_length == 960 and bf.spk_encoder.robust_width == 24
    y = bf.apply_to_signal(s["W"], (z["time_vec"], x))
    assert y.shape == (9599, 1440) and y.dtype == np.float64
    np.testing.assert_allclose(y[z["row_idx"]], z["y_rows"], rtol=0, atol=1e-12)
    power = np.mean(np.abs(y) ** 2, axis=0)
    np.testing.assert_allclose(power, z["power"], rtol=1e-10, atol=0)
    assert int(np.argmax(power)) == int(z["argmax"])
    out = bf.localize_batch(s["W"], x[None], time_vec=z["time_vec"], return_spikes=True)
    np.testing.assert_array_equal(out["spikes"][0].cpu().numpy(), z["spikes"])
    assert int(out["argmax"][0]) == int(z["argmax"])
    # synthesis: host surface and device kernel against the reference's noise-free rows
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * 2000 * time_test)
    t, xs = bf.synthesize_batch((time_test, sig_test), np.array([float(z["doa"])]))
    np.testing.assert_array_equal(t, z["time_vec"])
    np.testing.assert_allclose(xs[0].cpu().numpy()[z["clean_idx"]], z["clean_rows"], rtol=0, atol=1e-100)
    # apply_to_template with the reference's noise draw: the quantised recording is what the fixture holds
    cap = {}
    orig = bf.apply_to_signal

    def spy(bf_mat, sig_in_vec):
        cap["sig"] = np.array(sig_in_vec[1], copy=True)
        return np.zeros((1, 1))

    bf.apply_to_signal = spy
    try:
        np.random.seed(int(z["noise_seed"]))
        bf.apply_to_template(s["W"], (time_test, sig_test, float(z["doa"])), snr_db=float(z["snr_db"]))
    finally:
        bf.apply_to_signal = orig
    np.testing.assert_array_equal(np.rint(cap["sig"] * 4096.0).astype(np.int16), z["sig_q"])


def test_beamformer_c128_g449_plan_and_surface(cfg2, torch):
    """Complex Beamformer at G = 449 (58 tiles of 16 complex columns: three passes of beamform_wsc_kernel; with y stored: pairs of
    tiles): one reference trial, then the class surface incl. the design of a grid subset (covariances 1e-11, columns up to the
    unit phase of U[:, 0], host LAPACK and device Jacobi)."""
    from haghighatshoarmuir2024_amd.runtime import Plan
    from micloc.array_geometry import CenterCircularArray
    from micloc.beamformer import Beamformer

    z = golden("beamformer_c128_g449.npz")
    p = Plan(7, cfg2["kernel"], cfg2["b"], cfg2["a"], 1, False)
    p.set_bf_mat(z["bf_mat"])
    x = z["sig_in"]
    out = p.beamformer_pipeline(p.to_device(x[None]), want_y=True)
    ref = O.beamformer_chain(x, cfg2["kernel"], cfg2["b"], cfg2["a"], z["bf_mat"])
    y = out["y"][0].cpu().numpy()
    assert y.shape == (4799, 449) and y.dtype == np.complex128
    np.testing.assert_allclose(y, ref["y"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(y[z["row_idx"]], z["y_rows"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out["power"][0].cpu().numpy(), z["power"], rtol=1e-10)
    assert int(out["argmax"][0]) == int(z["argmax"])
    out2 = p.beamformer_pipeline(p.to_device(x[None]), want_y=False)
    np.testing.assert_allclose(out2["power"][0].cpu().numpy(), z["power"], rtol=1e-10)
    assert int(out2["argmax"][0]) == int(z["argmax"])

    fs = 48_000
    bf = Beamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], fs=fs)
    ys = bf.apply_to_signal(z["bf_mat"], x)
    np.testing.assert_allclose(ys[z["row_idx"]], z["y_rows"], rtol=0, atol=1e-11)
    np.random.seed(int(z["seed"]))
    doa = np.random.rand(1)[0] * 2 * np.pi
    assert doa == float(z["doa"])
    time_test = np.arange(0, 100e-3, step=1 / fs)
    y2 = bf.apply_to_template(z["bf_mat"], (time_test, np.sin(2 * np.pi * 2000 * time_test), doa), snr_db=float(z["snr_db"]))
    np.testing.assert_allclose(np.mean(np.abs(y2) ** 2, axis=0), z["power"], rtol=1e-10)
    t = np.arange(0, 1.0, step=1 / fs)
    period = t[-1]
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % period) / period) / fs)
    idx = z["cov_idx"]
    for svd in ("host", "device"):
        W, covs = bf.design_from_template((t, s), z["doa_list"][idx], svd=svd)
        np.testing.assert_allclose(np.asarray(covs), z["cov_sel"], rtol=0, atol=1e-11)
        phase = np.sum(np.conj(W) * z["bf_mat"][:, idx], axis=0)
        np.testing.assert_allclose(np.abs(phase), 1.0, rtol=0, atol=1e-9)
        np.testing.assert_allclose(W * (phase / np.abs(phase)), z["bf_mat"][:, idx], rtol=0, atol=1e-8)


@pytest.mark.parametrize("f", [3600, 8000])
def test_config1_designs_high_frequencies(f, torch):
    """array_resolution_snn.py's designs at 3.6 kHz and 8 kHz, complete 225-DoA grids, host LAPACK and device Jacobi, against the
    reference's bf_mat and the beam pattern the script plots (ref:paper_plots/array_resolution_snn.py:118-160).  At 8 kHz the
    encoder's robust width is 1 (no candidate is ever suppressed) and the band-pass reaches 16 kHz."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    z = golden("bf_mat_sin225_unipolar_hf.npz")
    fs = 48_000
    tau = 1 / (2 * np.pi * f)
    bf = SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [f / 2, 2 * f], [tau, tau], bipolar_spikes=False, fs=fs)
    assert bf.spk_encoder.robust_width == int(z[f"robust_width_f{f}"])
    t = np.arange(0, 0.4, step=1 / fs)
    ref = z[f"bf_mat_f{f}"]
    refc = ref[:7] + 1j * ref[7:]
    for svd in ("host", "device"):
        W = bf.design_from_template((t, np.sin(2 * np.pi * f * t)), z["doa_list"], svd=svd)
        assert W.shape == (14, 225)
        np.testing.assert_allclose(W, ref, rtol=0, atol=2e-7)
        Wc = W[:7] + 1j * W[7:]
        np.testing.assert_allclose(np.abs(Wc.conj().T @ Wc), np.abs(refc.conj().T @ refc), rtol=0, atol=1e-6)
    # the covariances the reference decomposed, from the device chain
    idx = z["cov_idx"]
    delays = bf.geometry.delays(z["doa_list"][idx], normalized=True)
    delays = delays - delays.min(axis=1, keepdims=True)
    tt = np.arange(t.min(), t.max(), step=1 / fs)
    ss = np.interp(tt, t, np.sin(2 * np.pi * f * t))
    td = np.maximum(tt.reshape(1, 1, -1) - delays[:, :, None], tt.min())
    sig = np.ascontiguousarray(np.transpose(np.interp(td.ravel(), tt, ss).reshape(td.shape), (0, 2, 1)))
    cov = bf.membrane_covariance_batch(sig, time_vec=tt, t_start=sig.shape[1] // 4).cpu().numpy()
    np.testing.assert_allclose(cov, z[f"cov_sel_f{f}"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("which", ["rand", "lin"])
def test_designs_on_other_geometries(which, torch):
    """The complete designs of array_resolution_random_snn.py (13 random microphones after np.random.seed(1): 26 channels -- two channel
    tiles of the covariance kernel, the 26 x 26 two-sided Jacobi -- 833 DoAs shifted by pi) and array_resolution_linear_snn.py
    (LinearArray, 449 DoAs in [0, pi], jittered template), host LAPACK and device Jacobi, against the reference's bf_mat and the beam
    pattern the scripts plot (ref:paper_plots/array_resolution_random_snn.py:100-170, array_resolution_linear_snn.py:120-190)."""
    from micloc.array_geometry import LinearArray, Random2DArray
    from micloc.snn_beamformer import SNNBeamformer

    z = golden("design_other_geometries.npz")
    fs, f = 48_000, int(z["freq_design"])
    tau = 1 / (2 * np.pi * f)
    t = np.arange(0, 0.6, step=1 / fs)
    if which == "rand":
        np.random.seed(1)
        geo = Random2DArray(radius=4.5e-2, num_mic=13)
        s = np.sin(2 * np.pi * f * t)
    else:
        geo = LinearArray(spacing=2 * 4.5e-2 / 7, num_mic=7, radius=4.5e-2)
        s = z["lin_template_f32"].astype(np.float64)
    np.testing.assert_array_equal(geo.r_vec, z[f"{which}_r"])
    np.testing.assert_array_equal(geo.theta_vec, z[f"{which}_theta"])
    M = len(geo)
    bf = SNNBeamformer(geo, 10e-3, [f / 2, 2 * f], [tau, tau], bipolar_spikes=False, fs=fs)
    ref = z[f"{which}_bf_mat"]
    refc = ref[:M] + 1j * ref[M:]
    for svd in ("host", "device"):
        W = bf.design_from_template((t, s), z[f"{which}_doa_list"], svd=svd)
        assert W.shape == ref.shape
        np.testing.assert_allclose(W, ref, rtol=0, atol=2e-7)
        Wc = W[:M] + 1j * W[M:]
        np.testing.assert_allclose(np.abs(Wc.conj().T @ Wc), np.abs(refc.conj().T @ refc), rtol=0, atol=1e-6)
    # the covariances the reference decomposed, from the device chain
    idx = z[f"{which}_cov_idx"]
    delays = geo.delays(z[f"{which}_doa_list"][idx], normalized=True)
    delays = delays - delays.min(axis=1, keepdims=True)
    tt = np.arange(t.min(), t.max(), step=1 / fs)
    ss = np.interp(tt, t, s)
    td = np.maximum(tt.reshape(1, 1, -1) - delays[:, :, None], tt.min())
    sig = np.ascontiguousarray(np.transpose(np.interp(td.ravel(), tt, ss).reshape(td.shape), (0, 2, 1)))
    cov = bf.membrane_covariance_batch(sig, time_vec=tt, t_start=sig.shape[1] // 4).cpu().numpy()
    np.testing.assert_allclose(cov, z[f"{which}_cov_sel"], rtol=0, atol=1e-12)


def test_beamformer_sweep_matches_reference(cfg2, torch):
    """The complex Beamformer through the sweep harness (parity mode: the reference's MT19937 stream replayed on the host) against the
    reference's own sweep output (ref:paper_plots/target_localization.py:400-440): every arg-max, p_max 1e-10, the MAE curve."""
    from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep
    from micloc.array_geometry import CenterCircularArray
    from micloc.beamformer import Beamformer

    z = golden("beamformer_sweep_seed0.npz")
    W = golden("beamformer_c128.npz")
    bf = Beamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], fs=48_000)
    res = noisy_target_sweep(bf, W["bf_mat"], W["doa_list"], snr_db_vec=z["snr_db_vec"], num_sim=int(z["num_sim"]), seed=int(z["seed"]), mode="parity")
    np.testing.assert_array_equal(res["doa"], z["doa"])
    np.testing.assert_array_equal(res["argmax"], z["argmax"])
    np.testing.assert_allclose(res["pmax"], z["pmax"], rtol=1e-10)
    np.testing.assert_allclose(res["err"], z["err"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(res["mae_deg"], z["mae_deg"], rtol=0, atol=1e-9)


def test_live_demo_frame_matches_reference(torch):
    """localization_demo_snn.Demo (the reference's live loop without the sound card) on the reference-generated pack: the designed
    bf_mat, the power pattern and the DoA the reference's own components produced (ref:micloc/localization_demo_snn.py:125-193)."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.localization_demo_snn import Demo

    z = golden("live_demo_frame.npz")
    demo = Demo(geometry=CenterCircularArray(4.5e-2, 7), freq_bands=z["freq_bands"], doa_list=z["doa_list"], recording_duration=0.25,
                kernel_duration=10e-3, bipolar_spikes=True, fs=48_000)
    W, R = demo.bf_mats[0], z["bf_mat0"]
    Wc, Rc = W[:7] + 1j * W[7:], R[:7] + 1j * R[7:]
    phase = np.sum(np.conj(Wc) * Rc, axis=0)
    np.testing.assert_allclose(Wc * (phase / np.abs(phase)), Rc, rtol=0, atol=1e-8)  # (up to the unit phase of U[:, 0])
    demo.bf_mats[0] = R  # the reference's own matrix for the frame itself
    data = z["pack"][:, :-1].astype(np.float64)
    np.testing.assert_allclose(demo.power_grid(data), z["power_grid"], rtol=1e-10, atol=0)
    assert demo.process_frame(z["pack"]) == float(z["doa_deg"])
    # pi-periodic error against the direction the pack was synthesised from
    assert abs(np.degrees(np.arcsin(abs(np.sin(np.radians(float(z["doa_deg"])) - float(z["true_doa"])))))) < 5
