"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle and the golden vectors.

Bars: bit-exact for the STHT output, the band-passed pre-encoder signal, the spikes and the arg-max;
the beamformed signal y and the power within 1e-12 relative of the oracle (both fp64; they differ only
in the order of the final time reduction) and within 1e-10 of the reference's golden vectors.
"""
import os

import numpy as np
import pytest

from conftest import campaign_seeds, golden
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def plan2(cfg2, torch):
    from haghighatshoarmuir2024_amd.runtime import Plan

    p = Plan(7, cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True)
    p.set_neuron_kernel(cfg2["nir"])
    p.set_bf_mat(cfg2["bf_mat"])
    return p


def planar_to_rows(h, T):
    """device planar [B, C, Ts] -> numpy [B, T, C]"""
    return np.ascontiguousarray(h[:, :, :T].cpu().numpy().transpose(0, 2, 1))


@pytest.mark.parametrize("T", [1, 7, 8, 100, 479, 480, 481, 513, 1500, 4799])
def test_stht_bit_exact(plan2, cfg2, T):
    rng = np.random.RandomState(T)
    x = rng.randn(3, T, 7)
    h = planar_to_rows(plan2.stht(plan2.to_device(x)), T)
    for b in range(3):
        re, im = O.stht(x[b], cfg2["kernel"])
        np.testing.assert_array_equal(h[b][:, :7], re)
        np.testing.assert_array_equal(h[b][:, 7:], im)


def test_stht_dense_and_odd_kernels(torch):
    """kstep = 1 path (a kernel with no zero taps), odd length, leading zeros, and an all-zero kernel."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    rng = np.random.RandomState(3)
    # (the stride-2 kernels -- odd taps only, even taps only, leading zeros, 960 taps: one tile per wave and the looped staging --
    #  run on the matrix cores, csrc/stht.hip stht_mfma_kernel; the dense ones on the vector ALU)
    for L, maker in [(37, lambda L: rng.randn(L)), (64, lambda L: np.r_[np.zeros(5), rng.randn(L - 5)]), (16, lambda L: np.zeros(L)),
                     (50, lambda L: np.where(np.arange(L) % 2 == 1, rng.randn(L), 0.0)),
                     (51, lambda L: np.where(np.arange(L) % 2 == 0, rng.randn(L), 0.0)),
                     (70, lambda L: np.where((np.arange(L) % 2 == 0) & (np.arange(L) >= 6), rng.randn(L), 0.0)),
                     (960, lambda L: np.where(np.arange(L) % 2 == 1, rng.randn(L), 0.0)),
                     (9, lambda L: np.where(np.arange(L) == 3, 1.5, 0.0))]:
        ker = maker(L)
        p = Plan(5, ker, [1.0], [1.0], 3, True)
        x = rng.randn(2, 700, 5)
        h = planar_to_rows(p.stht(p.to_device(x)), 700)
        for b in range(2):
            re, im = O.stht(x[b], ker)
            np.testing.assert_array_equal(h[b][:, :5], re)
            np.testing.assert_array_equal(h[b][:, 5:], im)


def test_bandpass_and_spikes_bit_exact(plan2, cfg2):
    z = golden("trials_cfg2.npz")
    x = z["sig_in"]
    h = plan2.stht(plan2.to_device(x))
    pre, spikes = plan2.bandpass_rzcc(h, x.shape[1])
    pre = planar_to_rows(pre, x.shape[1])
    spikes = spikes.cpu().numpy()
    for b in range(3):
        out = O.snn_chain(x[b], cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True, cfg2["nir"], cfg2["bf_mat"])
        np.testing.assert_array_equal(pre[b], out["pre_enc"])
        np.testing.assert_array_equal(spikes[b], out["spikes"])
        np.testing.assert_array_equal(spikes[b], z["spikes"][b])  # the reference's own spikes


def test_pipeline_vs_golden_and_oracle(plan2, cfg2):
    z = golden("trials_cfg2.npz")
    x = z["sig_in"]
    out = plan2.snn_pipeline(plan2.to_device(x), want_spikes=True, want_y=True, want_power=True)
    spikes = out["spikes"].cpu().numpy()
    y = out["y"].cpu().numpy()
    power = out["power"].cpu().numpy()
    argmax = out["argmax"].cpu().numpy()
    np.testing.assert_array_equal(spikes, z["spikes"])
    np.testing.assert_array_equal(argmax, z["argmax"])
    np.testing.assert_allclose(power, z["power"], rtol=1e-10, atol=0)
    for b in range(3):
        ref = O.snn_chain(x[b], cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True, cfg2["nir"], cfg2["bf_mat"])
        np.testing.assert_allclose(y[b], ref["y"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(power[b], ref["power"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(y[b][z["row_idx"]], z["y_rows"][b], rtol=0, atol=1e-12)
        assert argmax[b] == ref["argmax"]
    # power-only call (no T x G store; bf_mat-stationary kernel, different order of the sum over time): same numbers
    out2 = plan2.snn_pipeline(plan2.to_device(x), want_power=True)
    np.testing.assert_allclose(out2["power"].cpu().numpy(), power, rtol=1e-13, atol=0)
    np.testing.assert_array_equal(out2["argmax"].cpu().numpy(), argmax)


def test_membrane_and_y_bit_exact(plan2, cfg2):
    """bf_mat = I returns the membrane signal itself: the MFMA chains reproduce the oracle's fma order exactly."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    z = golden("trials_cfg2.npz")
    x = z["sig_in"][:1]
    p = Plan(7, cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True)
    p.set_neuron_kernel(cfg2["nir"])
    p.set_bf_mat(np.eye(14))
    v = p.snn_pipeline(p.to_device(x), want_y=True, want_power=False)["y"][0].cpu().numpy()
    ref = O.snn_chain(x[0], cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True, cfg2["nir"], cfg2["bf_mat"])
    np.testing.assert_array_equal(v, ref["vmem"])
    y = plan2.snn_pipeline(plan2.to_device(x), want_y=True, want_power=False)["y"][0].cpu().numpy()
    np.testing.assert_array_equal(y, ref["y"])


def _edge_cases():
    z = golden("rzcc_edge.npz")
    return z, sorted({n.split("__")[0] for n in z.files})


def test_rzcc_edge_cases(torch):
    from haghighatshoarmuir2024_amd.spike_encoder import ZeroCrossingSpikeEncoder

    z, names = _edge_cases()
    for n in names:
        x, w, bip = z[f"{n}__in"], int(z[f"{n}__w"]), int(z[f"{n}__bip"])
        enc = ZeroCrossingSpikeEncoder(fs=48_000, robust_width=w, bipolar=bool(bip))
        got = enc.evolve(x)
        assert got.dtype == x.dtype and got.shape == x.shape
        np.testing.assert_array_equal(got.astype(np.int8), O.rzcc(x, w, bip), err_msg=n)
        if not n.startswith("int_ties"):  # tie order is unspecified in the reference (see test_oracle_golden)
            np.testing.assert_array_equal(got.astype(np.int8), z[f"{n}__out"], err_msg=n)


def test_rzcc_random_batches_vs_oracle(torch):
    from haghighatshoarmuir2024_amd import runtime

    rng = np.random.RandomState(0)
    for (B, T, C, w, bip, kind) in [(5, 1000, 14, 12, 1, "walk"), (3, 777, 3, 1, 1, "noise"), (2, 2048, 130, 24, 0, "noise"),
                                    (4, 500, 7, 5, 1, "int"), (1, 3000, 2, 400, 1, "noise"), (70, 64, 1, 3, 1, "int"),
                                    # cluster sizes around the register-resident widths (4 / 8 / 16) and beyond (list walk)
                                    (3, 1500, 14, 4, 1, "noise"), (3, 1500, 14, 8, 1, "noise"), (3, 1500, 14, 16, 1, "noise"),
                                    (3, 1500, 14, 40, 1, "noise"), (2, 1500, 5, 9, 0, "noise"), (2, 900, 6, 7, 1, "int")]:
        if kind == "int":
            x = rng.randint(-2, 3, size=(B, T, C)).astype(np.float64)
        elif kind == "walk":
            x = np.sin(np.arange(T)[None, :, None] * 0.2 + rng.rand(B, 1, C) * 6) + 0.5 * rng.randn(B, T, C)
        else:
            x = rng.randn(B, T, C)
        got = runtime.rzcc_encode(x, w, bip).cpu().numpy()
        for b in range(B):
            np.testing.assert_array_equal(got[b], O.rzcc(x[b], w, bip), err_msg=f"{(B, T, C, w, bip, kind)} b={b}")


def test_unipolar_trial(torch):
    from haghighatshoarmuir2024_amd.runtime import Plan

    z = golden("unipolar_trial.npz")
    W = golden("bf_mat_sin225_unipolar.npz")["bf_mat_f2000"]
    fs, f = 48_000, 2000
    b, a = O.bandpass(fs, [0.5 * f, 2 * f])
    x = z["sig_in"]
    tau = 1 / (2 * np.pi * f)
    p = Plan(7, O.stht_kernel(fs, 10e-3), b, a, O.robust_width(fs, 2 * f), False)
    p.set_neuron_kernel(O.neuron_kernel(np.arange(x.shape[0]) / fs, [tau, tau]))
    p.set_bf_mat(W)
    out = p.snn_pipeline(p.to_device(x[None]), want_spikes=True)
    np.testing.assert_array_equal(out["spikes"][0].cpu().numpy(), z["spikes"])
    np.testing.assert_allclose(out["power"][0].cpu().numpy(), z["power"], rtol=1e-10)
    assert int(out["argmax"][0]) == int(z["argmax"])


def test_wide_case_generic_shapes(torch):
    """16 mics (C = 32 -> two channel tiles), 96 kHz (L = 960, w = 24, 71-tap neuron kernel), G = 90."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    z = golden("wide_case.npz")
    fs = int(z["fs"])
    b, a = O.bandpass(fs, [1000.0, 2000.0])
    tau = 1 / (2 * np.pi * 2000.0)
    nir = O.neuron_kernel(z["time_vec"], [tau, tau])
    ker = O.stht_kernel(fs, 10e-3)
    p = Plan(16, ker, b, a, O.robust_width(fs, 2000.0), True)
    p.set_neuron_kernel(nir)
    p.set_bf_mat(z["bf_mat"])
    x = z["sig_in"]
    out = p.snn_pipeline(p.to_device(x[None]), want_spikes=True, want_y=True)
    ref = O.snn_chain(x, ker, b, a, O.robust_width(fs, 2000.0), True, nir, z["bf_mat"])
    np.testing.assert_array_equal(out["spikes"][0].cpu().numpy(), z["spikes"])
    np.testing.assert_array_equal(out["spikes"][0].cpu().numpy(), ref["spikes"])
    np.testing.assert_allclose(out["y"][0].cpu().numpy(), ref["y"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(out["power"][0].cpu().numpy(), z["power"], rtol=1e-10)
    assert int(out["argmax"][0]) == int(z["argmax"])


def test_beamformer_c128(cfg2, torch):
    from haghighatshoarmuir2024_amd.runtime import Plan

    z = golden("beamformer_c128.npz")
    p = Plan(7, cfg2["kernel"], cfg2["b"], cfg2["a"], 1, False)
    p.set_bf_mat(z["bf_mat"])
    x = z["sig_in"]
    out = p.beamformer_pipeline(p.to_device(x[None]), want_y=True)
    ref = O.beamformer_chain(x, cfg2["kernel"], cfg2["b"], cfg2["a"], z["bf_mat"])
    y = out["y"][0].cpu().numpy()
    np.testing.assert_allclose(y, ref["y"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(y[z["row_idx"]], z["y_rows"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out["power"][0].cpu().numpy(), z["power"], rtol=1e-10)
    np.testing.assert_allclose(out["power"][0].cpu().numpy(), ref["power"], rtol=1e-12)
    assert int(out["argmax"][0]) == int(z["argmax"])


def test_lfilter_filterbank(cfg2, torch):
    from haghighatshoarmuir2024_amd.filterbank import ButterworthFilterbank
    from haghighatshoarmuir2024_amd import runtime

    z = golden("filterbank.npz")
    re, im = O.stht(z["sig_in"], cfg2["kernel"])
    sig_real = np.hstack([re, im])
    fb = ButterworthFilterbank(freq_bands=[[1000, 2000]], order=1, fs=48_000)
    filt = fb.evolve(sig_real)
    assert filt.shape == (1,) + sig_real.shape
    np.testing.assert_array_equal(filt[0], O.iir(z["b"], z["a"], sig_real))
    np.testing.assert_allclose(filt[0][:800], z["filt_head"], rtol=0, atol=1e-11)
    s = runtime.rzcc_encode(filt[0], cfg2["robust_width"], True).cpu().numpy()
    np.testing.assert_array_equal(np.hstack([(s > 0), (s < 0)]).astype(np.int8), z["spikes_in"])


def test_error_conventions(cfg2, torch):
    from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer
    from haghighatshoarmuir2024_amd.beamformer import Beamformer
    from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray

    geo = CenterCircularArray(4.5e-2, 7)
    tau = 1 / (2 * np.pi * 2000)
    bf = SNNBeamformer(geo, 10e-3, [1000, 2000], [tau, tau], bipolar_spikes=True)
    with pytest.raises(ValueError):
        bf.apply_to_signal(cfg2["bf_mat"], (np.arange(100) / 48000, np.zeros((100, 6))))
    with pytest.raises(ValueError):
        bf.apply_to_template(cfg2["bf_mat"], (np.arange(10), np.arange(10)), 3.0)
    with pytest.raises(ValueError):
        SNNBeamformer(geo, 10e-3, [2000, 1000], [tau, tau])
    with pytest.raises(ValueError):
        Beamformer(geo, 10e-3, [1000, 2000]).apply_to_signal(np.zeros((7, 5), complex), np.zeros((50, 3)))


@pytest.mark.parametrize("M,G,T", [(64, 50, 700), (24, 33, 530), (40, 16, 100)])
def test_many_mic_shapes_vs_oracle(torch, M, G, T):
    """Stress-shape plumbing (BASELINE config 5 uses 64 mics): C = 128 / 48 / 80 channels, 96 kHz kernel."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    fs = 96_000
    rng = np.random.RandomState(M)
    ker = O.stht_kernel(fs, 10e-3)
    b, a = O.bandpass(fs, [1000.0, 2000.0])
    w = O.robust_width(fs, 2000.0)
    tau = 1 / (2 * np.pi * 2000.0)
    nir = O.neuron_kernel(np.arange(T) / fs, [tau, tau])
    W = rng.randn(2 * M, G)
    x = np.sin(2 * np.pi * 1500 * np.arange(T)[None, :, None] / fs + rng.rand(2, 1, M) * 6) + 0.3 * rng.randn(2, T, M)
    p = Plan(M, ker, b, a, w, True)
    p.set_neuron_kernel(nir)
    p.set_bf_mat(W)
    out = p.snn_pipeline(p.to_device(x), want_spikes=True, want_y=True)
    for i in range(2):
        ref = O.snn_chain(x[i], ker, b, a, w, True, nir, W)
        np.testing.assert_array_equal(out["spikes"][i].cpu().numpy(), ref["spikes"])
        np.testing.assert_array_equal(out["y"][i].cpu().numpy(), ref["y"])
        np.testing.assert_allclose(out["power"][i].cpu().numpy(), ref["power"], rtol=1e-12)
        assert int(out["argmax"][i]) == ref["argmax"]


def test_covariance_form_power_and_membrane_covariance(plan2, cfg2):
    """w^T (V^T V / T) w == mean_t (V w)^2; the Gram kernel also returns V^T V / T' for the design path."""
    z = golden("trials_cfg2.npz")
    x = plan2.to_device(z["sig_in"])
    direct = plan2.snn_pipeline(x, want_power=True)
    cov = plan2.snn_pipeline_cov(x, want_cov=True, want_power=True, want_spikes=True)
    np.testing.assert_array_equal(cov["spikes"].cpu().numpy(), z["spikes"])
    np.testing.assert_allclose(cov["power"].cpu().numpy(), direct["power"].cpu().numpy(), rtol=1e-12, atol=0)
    np.testing.assert_allclose(cov["power"].cpu().numpy(), z["power"], rtol=1e-10, atol=0)
    np.testing.assert_array_equal(cov["argmax"].cpu().numpy(), z["argmax"])
    T = x.shape[1]
    stable = T // 4
    part = plan2.snn_pipeline_cov(x, t_start=stable, want_cov=True, want_power=False)["cov"].cpu().numpy()
    for b in range(3):
        ref = O.snn_chain(z["sig_in"][b], cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True, cfg2["nir"], cfg2["bf_mat"], want=("vmem",))
        v = ref["vmem"]
        np.testing.assert_allclose(cov["cov"][b].cpu().numpy(), v.T @ v / T, rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(part[b], v[stable:].T @ v[stable:] / (T - stable), rtol=1e-12, atol=1e-18)


def test_covariance_form_wide(torch):
    """C = 32 and C = 48 channels (two / three channel tiles -> 3 / 6 Gram tiles, register form); C = 56, 64, 80 and 128 channels
    (four to eight channel tiles: the LDS-shared kernel of BASELINE config 5, up to 36 Gram tiles dealt over the waves), incl. a
    ragged T and t_start."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    for M, G, T in [(16, 40, 900), (24, 20, 530), (28, 25, 650), (32, 30, 800), (40, 33, 700), (64, 70, 1111)]:
        fs = 96_000
        rng = np.random.RandomState(M)
        ker = O.stht_kernel(fs, 10e-3)
        b, a = O.bandpass(fs, [1000.0, 2000.0])
        tau = 1 / (2 * np.pi * 2000.0)
        nir = O.neuron_kernel(np.arange(T) / fs, [tau, tau])
        W = rng.randn(2 * M, G)
        x = np.sin(2 * np.pi * 1500 * np.arange(T)[None, :, None] / fs + rng.rand(2, 1, M) * 6) + 0.3 * rng.randn(2, T, M)
        p = Plan(M, ker, b, a, O.robust_width(fs, 2000.0), True)
        p.set_neuron_kernel(nir)
        p.set_bf_mat(W)
        xd = p.to_device(x)
        d = p.snn_pipeline(xd, want_power=True)
        c = p.snn_pipeline_cov(xd, want_cov=True, want_power=True)
        np.testing.assert_allclose(c["power"].cpu().numpy(), d["power"].cpu().numpy(), rtol=1e-11)
        ref = O.snn_chain(x[1], ker, b, a, O.robust_width(fs, 2000.0), True, nir, W, want=("vmem",))
        np.testing.assert_allclose(c["cov"][1].cpu().numpy(), ref["vmem"].T @ ref["vmem"] / T, rtol=1e-11, atol=1e-18)
        assert np.array_equal(c["argmax"].cpu().numpy(), d["argmax"].cpu().numpy())
        ts = T // 4
        part = p.snn_pipeline_cov(xd, t_start=ts, want_cov=True, want_power=False)["cov"][1].cpu().numpy()
        np.testing.assert_allclose(part, ref["vmem"][ts:].T @ ref["vmem"][ts:] / (T - ts), rtol=1e-11, atol=1e-18)


def test_many_mic_complex_beamformer(torch):
    """Complex Beamformer with 40 mics (80 stacked channels -> the slab kernel with the planar source)."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    fs, M, G, T = 48_000, 40, 21, 600
    rng = np.random.RandomState(8)
    ker = O.stht_kernel(fs, 10e-3)
    b, a = O.bandpass(fs, [1000.0, 2000.0])
    W = rng.randn(M, G) + 1j * rng.randn(M, G)
    x = rng.randn(2, T, M)
    p = Plan(M, ker, b, a, 1, False)
    p.set_bf_mat(W)
    out = p.beamformer_pipeline(p.to_device(x), want_y=True)
    for i in range(2):
        ref = O.beamformer_chain(x[i], ker, b, a, W)
        np.testing.assert_allclose(out["y"][i].cpu().numpy(), ref["y"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(out["power"][i].cpu().numpy(), ref["power"], rtol=1e-12)
        assert int(out["argmax"][i]) == ref["argmax"]


@pytest.mark.parametrize("M,G,T", [(1, 5, 40), (2, 300, 530), (3, 17, 257), (4, 513, 300), (5, 64, 1000), (6, 100, 33), (7, 449, 700),
                                   (7, 57, 4799), (8, 360, 512), (7, 256, 255), (7, 257, 17)])
def test_complex_beamformer_bf_stationary_shapes(torch, M, G, T):
    """beamform_wsc_kernel (up to 8 microphones): every k-step / vector-tail split (2M = 2 ... 16 channels), one to several DoA passes,
    ragged recordings, y stored (flat blocks for G <= 256, row segments above) and power only -- vs the oracle's
    Beamformer.apply_to_signal (micloc/beamformer.py:260-292)."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    fs = 48_000
    rng = np.random.RandomState(100 * M + G)
    ker = O.stht_kernel(fs, 10e-3)
    b, a = O.bandpass(fs, [1000.0, 2000.0])
    W = rng.randn(M, G) + 1j * rng.randn(M, G)
    x = rng.randn(3, T, M)
    p = Plan(M, ker, b, a, 1, False)
    p.set_bf_mat(W)
    xd = p.to_device(x)
    out = p.beamformer_pipeline(xd, want_y=True)
    only = p.beamformer_pipeline(xd, want_y=False)
    assert torch.equal(out["power"], only["power"]) and torch.equal(out["argmax"], only["argmax"])
    for i in range(3):
        ref = O.beamformer_chain(x[i], ker, b, a, W)
        np.testing.assert_allclose(out["y"][i].cpu().numpy(), ref["y"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(out["power"][i].cpu().numpy(), ref["power"], rtol=1e-12)
        assert int(out["argmax"][i]) == ref["argmax"]


@pytest.mark.parametrize("M,G", [(8, 449), (8, 512), (6, 400), (8, 513)])
def test_eight_mics_many_doas_vs_oracle(torch, M, G):
    """16 (and 12) channels with four DoA tiles per wave (385-512 DoAs: beamform_ws_kernel's two-pass form without a vector tail), and
    one DoA more than the bf_mat-stationary kernel serves (513: the general kernel with one channel tile) -- vs the oracle."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    fs, T = 48_000, 700
    rng = np.random.RandomState(M * 1000 + G)
    ker = O.stht_kernel(fs, 10e-3)
    b, a = O.bandpass(fs, [1000.0, 2000.0])
    w = O.robust_width(fs, 2000.0)
    tau = 1 / (2 * np.pi * 2000.0)
    nir = O.neuron_kernel(np.arange(T) / fs, [tau, tau])
    W = rng.randn(2 * M, G)
    x = rng.randn(2, T, M)
    p = Plan(M, ker, b, a, w, True)
    p.set_neuron_kernel(nir)
    p.set_bf_mat(W)
    xd = p.to_device(x)
    out = p.snn_pipeline(xd, want_spikes=True, want_y=True, want_power=True)
    only = p.snn_pipeline(xd, want_power=True)
    for i in range(2):
        ref = O.snn_chain(x[i], ker, b, a, w, True, nir, W)
        np.testing.assert_array_equal(out["spikes"][i].cpu().numpy(), ref["spikes"])
        np.testing.assert_array_equal(out["y"][i].cpu().numpy(), ref["y"])
        np.testing.assert_allclose(out["power"][i].cpu().numpy(), ref["power"], rtol=1e-12)
        np.testing.assert_allclose(only["power"][i].cpu().numpy(), ref["power"], rtol=1e-12)
        assert int(out["argmax"][i]) == ref["argmax"] == int(only["argmax"][i])


@pytest.mark.parametrize("T", [2, 3, 16, 17, 31, 100, 511, 512, 513, 1025])
def test_pipeline_short_and_boundary_lengths(plan2, cfg2, T):
    """Ragged lengths around every tile size in the kernels (16-step RZCC tiles, 512-frame beamforming chunks)."""
    rng = np.random.RandomState(T)
    x = rng.randn(2, T, 7)
    out = plan2.snn_pipeline(plan2.to_device(x), want_spikes=True, want_y=True, want_power=True)
    cov = plan2.snn_pipeline_cov(plan2.to_device(x), want_power=True)
    for i in range(2):
        ref = O.snn_chain(x[i], cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True, cfg2["nir"], cfg2["bf_mat"])
        np.testing.assert_array_equal(out["spikes"][i].cpu().numpy(), ref["spikes"])
        np.testing.assert_array_equal(out["y"][i].cpu().numpy(), ref["y"])
        np.testing.assert_allclose(out["power"][i].cpu().numpy(), ref["power"], rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(cov["power"][i].cpu().numpy(), ref["power"], rtol=1e-11, atol=1e-300)
        assert int(out["argmax"][i]) == ref["argmax"]


def test_empty_and_invalid_inputs(plan2, torch):
    from haghighatshoarmuir2024_amd import _lib, runtime

    assert runtime.rzcc_encode(np.zeros((0, 3)), 3, True).shape == (0, 3)
    assert runtime.rzcc_encode(np.zeros((5, 0)), 3, True).shape == (5, 0)
    with pytest.raises(ValueError):
        plan2.snn_pipeline(plan2.to_device(np.zeros((1, 50, 6))))
    with pytest.raises((_lib.MiclocError, ValueError)):
        plan2.set_bf_mat(np.zeros((12, 5)))


def test_pipeline_is_graph_capturable(plan2, cfg2, torch):
    """include/micloc_hip.h promises that the stage calls neither allocate nor synchronise: capture the fused
    pipeline in a HIP graph, replay it on new input data, compare with the eager result."""
    z = golden("trials_cfg2.npz")
    x = plan2.to_device(z["sig_in"])
    eager = plan2.snn_pipeline(x, want_spikes=True, want_power=True)
    static_x = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan2.snn_pipeline(static_x, want_spikes=True, want_power=True)  # warm-up: workspace is allocated here
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = plan2.snn_pipeline(static_x, want_spikes=True, want_power=True)
    static_x.copy_(x.flip(0))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["spikes"], eager["spikes"].flip(0))
    assert torch.equal(out["power"], eager["power"].flip(0))
    assert torch.equal(out["argmax"], eager["argmax"].flip(0))


def test_f32_mfma_beamform_variant(plan2, cfg2):
    """fp32-MFMA tail (variant): spikes untouched (fp64 front end), power within 1e-5 of the fp64 path and of the
    reference (BASELINE north star: 'within 1e-5 float32'), arg-max equal on the golden trials."""
    z = golden("trials_cfg2.npz")
    x = plan2.to_device(z["sig_in"])
    d = plan2.snn_pipeline(x, want_power=True)
    f = plan2.snn_pipeline_f32bf(x, want_spikes=True)
    np.testing.assert_array_equal(f["spikes"].cpu().numpy(), z["spikes"])
    np.testing.assert_allclose(f["power"].cpu().numpy(), d["power"].cpu().numpy(), rtol=1e-5, atol=0)
    np.testing.assert_allclose(f["power"].cpu().numpy(), z["power"], rtol=1e-5, atol=0)
    rel = np.abs(f["power"].cpu().numpy() / d["power"].cpu().numpy() - 1).max()
    assert rel < 5e-6, rel
    np.testing.assert_array_equal(f["argmax"].cpu().numpy(), z["argmax"])
    # ragged length + two channel tiles
    from haghighatshoarmuir2024_amd.runtime import Plan

    w = golden("wide_case.npz")
    fs = int(w["fs"])
    b, a = O.bandpass(fs, [1000.0, 2000.0])
    tau = 1 / (2 * np.pi * 2000.0)
    p = Plan(16, O.stht_kernel(fs, 10e-3), b, a, O.robust_width(fs, 2000.0), True)
    p.set_neuron_kernel(O.neuron_kernel(w["time_vec"], [tau, tau]))
    p.set_bf_mat(w["bf_mat"])
    f2 = p.snn_pipeline_f32bf(p.to_device(w["sig_in"][None]))
    np.testing.assert_allclose(f2["power"][0].cpu().numpy(), w["power"], rtol=1e-5)
    assert int(f2["argmax"][0]) == int(w["argmax"])


@pytest.mark.parametrize(
    "C,G,T,n_nir",
    [
        (14, 5, 1, 35),       # one frame, one DoA tile (NGW = 1)
        (14, 128, 15, 35),    # shorter than a 16-frame tile
        (14, 200, 16, 1),     # single-tap neuron kernel, NGW = 2
        (14, 360, 511, 35),   # one frame short of a chunk
        (14, 360, 512, 4),    # exactly one chunk
        (14, 360, 513, 50),   # one frame into the second chunk, 50 taps (NK = 17: unrolled part + remainder)
        (14, 449, 1037, 35),  # NGW = 4 (script-exact DoA grid)
        (14, 449, 257, 35),   # one frame into the second 256-frame workgroup of the bf_mat-stationary kernels
        (14, 1, 300, 35),     # a single DoA column
        (14, 17, 256, 35),    # one column into the second DoA tile; exactly one workgroup
        (14, 512, 700, 71),   # 32 DoA tiles (largest bf_mat-stationary shape), long kernel
        (2, 24, 1100, 35),    # single microphone (one k-step)
        (6, 225, 600, 35),    # one k-step + two channels on the vector ALU
        (8, 130, 300, 35),    # two full k-steps
        (10, 64, 300, 35),    # two k-steps + two channels on the vector ALU
        (12, 300, 520, 35),   # three full k-steps
        (16, 100, 600, 35),   # C = 16: no channel padding
        (14, 520, 600, 35),   # 33 DoA tiles: falls back to the time-stationary kernel
    ],
)
def test_lif_beamform_stage_shapes_vs_oracle(torch, C, G, T, n_nir):
    """Stage API `micloc_lif_beamform_f64`, power-only and with y stored (bf_mat-stationary kernels where eligible, the
    y rows leaving through an LDS block; time-stationary kernel otherwise), against the oracle's LIF FIR + beamforming +
    power on random ternary spike trains.  The y-only call writes into the middle of a sentinel-filled buffer: nothing
    outside [B][T][G] may change."""
    from haghighatshoarmuir2024_amd import _lib, runtime
    from haghighatshoarmuir2024_amd.runtime import Plan

    rng = np.random.default_rng(C * 1000 + G + T)
    bipolar = True
    M = C // 2
    B = 3
    nir = rng.standard_normal(n_nir)
    W = rng.standard_normal((C, G))
    p = Plan(M, np.array([0.0, 1.0, 0.0, -1.0]), np.array([1.0]), np.array([1.0]), 2, bipolar)
    p.set_neuron_kernel(nir)
    p.set_bf_mat(W)
    spikes = (rng.integers(-1, 2, size=(B, T, C)) * (rng.random((B, T, C)) < 0.3)).astype(np.int8)
    sd = torch.from_numpy(spikes).cuda()
    out_p = p.lif_beamform(sd, want_y=False, want_power=True)
    out_y = p.lif_beamform(sd, want_y=True, want_power=True)
    # y only (no workspace, no partial sums), guarded on both sides
    pad = 4096
    buf = torch.full((pad + B * T * G + pad,), 12345.5, dtype=torch.float64, device="cuda")
    y_only = buf[pad:pad + B * T * G]
    _lib.check(p.lib.micloc_lif_beamform_f64(p.handle, runtime._ptr(sd), B, T, runtime._ptr(y_only), None, None, None, 0,
                                             runtime._stream(p.device)), "lif_beamform")
    assert bool((buf[:pad] == 12345.5).all()) and bool((buf[-pad:] == 12345.5).all())
    assert torch.equal(y_only.view(B, T, G), out_y["y"])
    for b in range(B):
        v = O.lif_fir(spikes[b], nir)
        y = O.beamform(v, W)
        power = (y * y).sum(axis=0) / T
        np.testing.assert_array_equal(out_y["y"][b].cpu().numpy(), y)
        for out in (out_p, out_y):
            np.testing.assert_allclose(out["power"][b].cpu().numpy(), power, rtol=1e-12, atol=1e-300)
            assert int(out["argmax"][b]) == int(np.argmax(out["power"][b].cpu().numpy()))
    # the two kernels agree on the arg-max unless the two best powers tie to rounding
    pw = out_y["power"].cpu().numpy()
    if G < 2:
        return
    top2 = np.sort(pw, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-9 * top2[:, 1]
    np.testing.assert_array_equal(out_p["argmax"].cpu().numpy()[clear], out_y["argmax"].cpu().numpy()[clear])


def _draw_fused_configuration(seed):
    """One random configuration of the fused pipeline.  The channel counts cover all three beamforming forms (C <= 16: bf_mat-stationary;
    C = 24 / 26: the general kernel with two channel tiles; C = 128: BASELINE config 5's shape), robust widths 1 ... 40, band-pass orders
    1 ... 3 (order 1 = config 4's filter: dense candidate trains), recordings from one frame to 50 000."""
    from scipy.signal import butter

    rng = np.random.default_rng(1000 + seed)
    M = int(rng.choice([1, 2, 3, 5, 7, 8, 12, 13, 64], p=[0.07, 0.07, 0.07, 0.07, 0.32, 0.08, 0.10, 0.12, 0.10]))
    L = int(rng.choice([8, 30, 64, 96, 200, 480] + ([960] if M == 64 else [])))
    kernel = rng.standard_normal(L)
    if rng.random() < 0.6:
        kernel[::2] = 0.0  # Hilbert-like stride-2 pattern (matrix-core STHT)
    order = int(rng.choice([1, 2, 2, 3]))
    b, a = butter(order, [0.05, 0.2 + 0.1 * rng.random()], btype="bandpass")
    w = int(rng.integers(1, 41))
    bipolar = bool(rng.random() < 0.7)
    n_nir = int(rng.choice([1, 7, 35, 60, 71]))
    nir = np.abs(rng.standard_normal(n_nir)) + 0.01
    G = int(rng.choice([3, 16, 17, 100, 200, 361, 449] + ([1440] if M == 64 else [])))
    if M == 64:
        T = int(rng.choice([17, 257, 1500, 4799]))
        B = int(rng.choice([1, 2]))
    else:
        T = int(rng.choice([1, 2, 17, 255, 256, 257, 700, 1500, 4799, 12000, 50000], p=[0.04, 0.04, 0.08, 0.1, 0.1, 0.1, 0.2, 0.2, 0.08, 0.04, 0.02]))
        B = 1 if T > 5000 else int(rng.choice([1, 2, 5]))
    W = rng.standard_normal((2 * M, G))
    kind = seed % 4
    t = np.arange(T)[None, :, None]
    if kind == 0:
        x = rng.standard_normal((B, T, M))
    elif kind == 1:  # tone + weak noise: long monotone stretches of the cumulative sum
        x = np.sin(0.3 * t + rng.random((B, 1, M)) * 6.28) + 0.01 * rng.standard_normal((B, T, M))
    elif kind == 2:  # quantised: exact ties and plateaus
        x = np.round(3 * rng.standard_normal((B, T, M)))
    else:  # silent start, then signal: streams without a direction for a while
        x = rng.standard_normal((B, T, M)) * (t >= T // 2)
    return rng, dict(M=M, L=L, kernel=kernel, order=order, b=b, a=a, w=w, bipolar=bipolar, nir=nir, G=G, T=T, B=B, W=W, x=x)


@pytest.mark.parametrize("seed", campaign_seeds("fused", 600))
def test_fused_pipeline_random_configurations_vs_oracle(torch, seed):
    """Randomised configurations (microphones 1 ... 64, STHT length and tap pattern, filter order, robust width 1 ... 40, polarity,
    neuron-kernel length, DoA count, trial length up to 50 000, signal character) through the fused pipeline against the oracle:
    spikes bit-exact, power 1e-12, same arg-max; then the same batch with a random time chunking of the encoder: identical bits.
    600 seeds in the driver's run (the id says so); the builder's campaigns set MICLOC_RANDOM_SEEDS."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    rng, c = _draw_fused_configuration(seed)
    M, w, bipolar, T, B, G, x = c["M"], c["w"], c["bipolar"], c["T"], c["B"], c["G"], c["x"]
    tag = f"seed={seed} M={M} L={c['L']} order={c['order']} w={w} bip={bipolar} n_nir={len(c['nir'])} G={G} T={T} B={B}"
    p = Plan(M, c["kernel"], c["b"], c["a"], w, bipolar)
    p.set_neuron_kernel(c["nir"])
    p.set_bf_mat(c["W"])
    xd = p.to_device(x)
    out = p.snn_pipeline(xd, want_spikes=True, want_power=True)
    spikes, power, argmax = out["spikes"].cpu().numpy(), out["power"].cpu().numpy(), out["argmax"].cpu().numpy()
    for i in range(B):
        ref = O.snn_chain(x[i], c["kernel"], c["b"], c["a"], w, bipolar, c["nir"], c["W"], want=("spikes", "power"))
        np.testing.assert_array_equal(spikes[i], ref["spikes"], err_msg=tag)
        np.testing.assert_allclose(power[i], ref["power"], rtol=1e-12, atol=1e-300, err_msg=tag)
        top = np.sort(power[i])[-2:] if G > 1 else np.array([0.0, power[i][0]])
        if top[1] - top[0] > 1e-9 * abs(top[1]):
            assert int(argmax[i]) == ref["argmax"], tag
    # the same batch with the time axis cut into chunks (scan checkpoints + one workgroup per chunk): identical spikes
    if T >= 64:
        lo = 16 * (-(-w // 16) + 1)
        chunk = int(rng.integers(lo, max(lo + 1, min(T, 6000))))
        p.set_encoder_chunk(chunk)
        out2 = p.snn_pipeline(xd, want_spikes=True, want_power=True)
        np.testing.assert_array_equal(out2["spikes"].cpu().numpy(), spikes, err_msg=f"chunk={chunk} " + tag)
        np.testing.assert_array_equal(out2["power"].cpu().numpy(), power, err_msg=f"chunk={chunk} " + tag)


def test_planar_gram_against_numpy(torch):
    """micloc_planar_gram_f64 (complex covariance of Beamformer.design_from_template, beamformer.py:142-150): the real Gram
    matrix of a planar signal over frames >= t_start, every tile-pair shape incl. channel padding and ragged chunk ends;
    fp64 tolerance 1e-14 relative to the largest entry (NumPy sums in another order)."""
    from haghighatshoarmuir2024_amd import runtime

    rng = np.random.RandomState(0)
    for B, C, T, t0 in ((3, 14, 4799, 480), (2, 80, 5000, 17), (1, 128, 9599, 2400), (2, 5, 100, 0), (1, 16, 2048, 0), (1, 17, 2049, 1)):
        Ts = (T + 7) // 8 * 8
        x = rng.randn(B, C, Ts)
        xd = torch.from_numpy(x).cuda()
        for norm in (True, False):
            g = runtime.planar_gram(xd, T, t_start=t0, normalise=norm).cpu().numpy()
            ref = np.einsum("bct,bdt->bcd", x[:, :, t0:T], x[:, :, t0:T]) / ((T - t0) if norm else 1.0)
            np.testing.assert_allclose(g, ref, rtol=0, atol=1e-14 * np.abs(ref).max() * (T - t0) ** 0.5)
            np.testing.assert_array_equal(g, np.transpose(g, (0, 2, 1)))  # exactly symmetric: one tile serves both halves
    with pytest.raises(ValueError):
        runtime.planar_gram(xd, 2049, t_start=2049)


def test_stht_vector_form_still_exact():
    """The vector-ALU STHT (stht_kernel) serves dense and very long kernels in the product; for the stride-2 Hilbert kernels it is the A/B
    partner of the matrix-core form.  The `stht_valu` variant library (tools/dev/make_variant.py: the same sources with
    VARIANT_STHT_VECTOR_FORM flipped, built by __graft_entry__.build()) sends them through it: a fresh child process loads that
    library and checks it against the oracle like test_stht_bit_exact does.  (The shipped library has no run-time switch.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(root, "tools", "_variants", "libmicloc_hip_stht_valu.so")
    # built by __graft_entry__.build(); (re)built here if it is missing or older than the objects it links (a no-op otherwise)
    mk = subprocess.run([sys.executable, os.path.join(root, "tools", "dev", "make_variant.py"), "stht_valu"], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=900)
    assert mk.returncode == 0 and os.path.exists(variant), mk.stderr.decode()[-2000:]
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from haghighatshoarmuir2024_amd import _lib
_lib.LIB_PATH = %r
from oracle import oracle as O
from haghighatshoarmuir2024_amd.runtime import Plan
fs = 48000
ker = O.stht_kernel(fs, 10e-3)
b, a = O.bandpass(fs, [1000.0, 2000.0])
p = Plan(7, ker, b, a, O.robust_width(fs, 2000.0), True)
rng = np.random.RandomState(1)
for T in (33, 700, 1500):
    x = rng.randn(2, T, 7)
    h = p.stht(p.to_device(x))[:, :, :T].cpu().numpy().transpose(0, 2, 1)
    for i in range(2):
        re, im = O.stht(x[i], ker)
        assert np.array_equal(h[i][:, :7], re) and np.array_equal(h[i][:, 7:], im), T
print("ok")
""" % (root, os.path.dirname(os.path.abspath(__file__)), variant)
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr.decode()[-2000:]


def test_event_driven_lif_variant_is_exact():
    """The `ws_sparse_lif` variant library (beamform_ws_kernel's LIF stage walked spike by spike instead of multiplied densely: round 5's
    experiment, as fast as the product's form, not faster -- DESIGN.md 4.3) gives the oracle's membrane sums bit for bit: power equal
    to the product's to the last bit on the reference's trials, on odd channel counts, ragged lengths, a neuron kernel at the walk's
    limit (65 taps), a DENSE raster and a raster that is not ternary (the general path broadcasts the value from its lane)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(root, "tools", "_variants", "libmicloc_hip_ws_sparse_lif.so")
    mk = subprocess.run([sys.executable, os.path.join(root, "tools", "dev", "make_variant.py"), "ws_sparse_lif"], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=900)
    assert mk.returncode == 0 and os.path.exists(variant), mk.stderr.decode()[-2000:]
    code = r"""
import numpy as np, sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r)
from haghighatshoarmuir2024_amd import _lib
if sys.argv[1] == "variant":
    _lib.LIB_PATH = %r
import torch
from oracle import oracle as O
from haghighatshoarmuir2024_amd.runtime import Plan
fs = 48000
ker = O.stht_kernel(fs, 10e-3)
b, a = O.bandpass(fs, [1000.0, 2000.0])
rng = np.random.RandomState(3)
out = {}
cases = [(7, 360, 4799, 35, "enc"), (7, 449, 700, 35, "enc"), (3, 40, 1000, 20, "enc"), (5, 100, 257, 65, "enc"), (8, 33, 513, 7, "enc"),
         (7, 64, 900, 35, "dense"), (6, 50, 600, 35, "int8"), (1, 17, 300, 3, "dense")]
for (M, G, T, n, kind) in cases:
    p = Plan(M, ker, b, a, 12, True)
    nir = np.abs(rng.randn(n)) / n
    p.set_neuron_kernel(nir)
    W = rng.randn(2 * M, G)
    p.set_bf_mat(W)
    B = 3
    if kind == "enc":
        x = np.sin(2 * np.pi * 1700 * np.arange(T) / fs)[None, :, None] + 0.8 * rng.randn(B, T, M)
        spikes = p.snn_pipeline(p.to_device(x), want_spikes=True, want_power=False)["spikes"]
    elif kind == "dense":
        spikes = torch.from_numpy(rng.randint(-1, 2, size=(B, T, 2 * M)).astype(np.int8)).cuda()
    else:
        spikes = torch.from_numpy(rng.randint(-5, 6, size=(B, T, 2 * M)).astype(np.int8)).cuda()
    r = p.lif_beamform(spikes, want_power=True)
    power = r["power"].cpu().numpy()
    s = spikes.cpu().numpy()
    for i in range(B):
        v = O.lif_fir(s[i], nir)
        y = O.beamform(v, W)
        ref = np.mean(y * y, axis=0)
        assert np.allclose(power[i], ref, rtol=1e-12, atol=0), (M, G, T, n, kind)
    out[str((M, G, T, n, kind))] = power
np.savez(sys.argv[2], **out)
print("ok")
""" % (root, os.path.dirname(os.path.abspath(__file__)), variant)
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        got = {}
        for which in ("product", "variant"):
            path = os.path.join(d, which + ".npz")
            r = subprocess.run([sys.executable, "-c", code, which, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert r.returncode == 0 and b"ok" in r.stdout, (which, r.stderr.decode()[-2000:])
            got[which] = dict(np.load(path))
        assert set(got["product"]) == set(got["variant"]) and len(got["product"]) == 8
        for k in got["product"]:
            np.testing.assert_array_equal(got["product"][k], got["variant"][k], err_msg=k)  # bit for bit
