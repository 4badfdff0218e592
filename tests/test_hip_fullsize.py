"""GPU tests at BASELINE.json's full config-2 size (1100 trials x 4799 frames x 7 mics, G = 449) through
size-independent properties, plus spot checks of individual trials against the oracle."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

B, T, M = 1100, 4799, 7


@pytest.fixture(scope="module")
def big(cfg2):
    import torch

    from haghighatshoarmuir2024_amd.runtime import Plan

    p = Plan(M, cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True)
    p.set_neuron_kernel(cfg2["nir"])
    p.set_bf_mat(cfg2["bf_mat"])
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    t = torch.arange(T, device="cuda", dtype=torch.float64) / 48_000
    phase = torch.rand((B, 1, M), generator=g, device="cuda", dtype=torch.float64) * 6.28
    amp = 10 ** (torch.rand((B, 1, 1), generator=g, device="cuda", dtype=torch.float64) * 2 - 1)
    x = torch.sin(2 * np.pi * 2000 * t[None, :, None] + phase) + amp * torch.randn((B, T, M), generator=g, device="cuda", dtype=torch.float64)
    out = p.snn_pipeline(x, want_spikes=True, want_power=True)
    torch.cuda.synchronize()
    return p, x, out


def test_spot_trials_against_oracle(big, cfg2):
    p, x, out = big
    for i in (0, 1, 547, 1099):
        ref = O.snn_chain(x[i].cpu().numpy(), cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True, cfg2["nir"], cfg2["bf_mat"],
                          want=("spikes", "power"))
        np.testing.assert_array_equal(out["spikes"][i].cpu().numpy(), ref["spikes"])
        np.testing.assert_allclose(out["power"][i].cpu().numpy(), ref["power"], rtol=1e-12, atol=0)
        assert int(out["argmax"][i]) == ref["argmax"]


def test_deterministic_and_batch_independent(big):
    import torch

    p, x, out = big
    again = p.snn_pipeline(x, want_spikes=True, want_power=True)
    assert torch.equal(again["spikes"], out["spikes"]) and torch.equal(again["power"], out["power"]) and torch.equal(again["argmax"], out["argmax"])
    sub = p.snn_pipeline(x[500:503].contiguous(), want_spikes=True, want_power=True)
    assert torch.equal(sub["spikes"], out["spikes"][500:503]) and torch.equal(sub["power"], out["power"][500:503])


def test_power_of_two_scaling_invariance(big):
    """Every stage before the encoder is linear and a scale by 2^k is exact in binary64, so the spikes (hence the
    power and arg-max) must not change at all."""
    import torch

    p, x, out = big
    for k in (-7, 5):
        sc = p.snn_pipeline(x * (2.0**k), want_spikes=True, want_power=True)
        assert torch.equal(sc["spikes"], out["spikes"])
        assert torch.equal(sc["power"], out["power"]) and torch.equal(sc["argmax"], out["argmax"])


def test_min_distance_property_and_counts(big, cfg2):
    """No two same-polarity spikes of a channel are closer than robust_width; argmax is the arg-max of power."""
    import torch

    p, x, out = big
    w = cfg2["robust_width"]
    s = out["spikes"]
    for mark in (1, -1):
        m = (s == mark).to(torch.int32)
        c = torch.cumsum(m, dim=1)
        win = c[:, w - 1 :, :] - torch.nn.functional.pad(c, (0, 0, 1, 0))[:, : c.shape[1] - w + 1, :]
        assert int(win.max()) <= 1
    assert torch.equal(out["power"].argmax(dim=1).to(torch.int32), out["argmax"])
    rate = (s != 0).double().mean().item()
    assert 0.04 < rate < 0.12  # ~3.5 k spikes / channel / s at 48 kHz (density 6.8 %)


def test_negation_swaps_polarity(big):
    """x -> -x negates the cumulative sum exactly: maxima and minima swap, so spikes flip sign bit for bit, except
    where a priority tie is broken by index order (none with continuous data)."""
    import torch

    p, x, out = big
    neg = p.snn_pipeline(-x[:64].contiguous(), want_spikes=True, want_power=True)
    assert torch.equal(neg["spikes"], -out["spikes"][:64])
    torch.testing.assert_close(neg["power"], out["power"][:64], rtol=1e-12, atol=0)


def test_graph_replays_full_size(big):
    """Regression: a captured hipMemsetAsync node was not ordered before the kernels that follow it (stale
    flagged-stream counter -> wild scatter / GPU memory fault on the third replay at this size).  The zero fill is a
    kernel now; replay the stage-API composition (shared workspace head) and the fused pipeline several times."""
    import torch

    p, x, out = big
    variants = {"f32": lambda: p.snn_pipeline_f32bf(x), "direct": lambda: p.snn_pipeline(x, want_power=True),
                "cov": lambda: p.snn_pipeline_cov(x, want_power=True)}
    ref = {k: fn() for k, fn in variants.items()}
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    for name, fn in variants.items():
        with torch.cuda.stream(s):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            o = fn()
        for _ in range(5):
            with torch.cuda.stream(s):
                g.replay()
            torch.cuda.synchronize()
            assert torch.equal(o["argmax"], ref[name]["argmax"]), name
            assert torch.equal(o["power"], ref[name]["power"]), name
    assert torch.equal(ref["direct"]["argmax"], out["argmax"])


def test_config5_shape_vs_oracle():
    """BASELINE config 5 (stress shape) at its real parameters: 64-mic random planar array, 96 kHz (960-tap STHT,
    w = 24, 71-tap neuron kernel), 1440 DoAs, T = 9599 frames; two noisy trials through the fused pipeline against the
    oracle (spikes bit-exact, power 1e-12, same arg-max).  Throughput of this shape: tools/stress_config5.py."""
    import torch

    from haghighatshoarmuir2024_amd.array_geometry import Random2DArray
    from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer, neuron_impulse_response

    fs, M5, G5 = 96_000, 64, 1440
    np.random.seed(1)
    geometry = Random2DArray(radius=0.2, num_mic=M5)
    tau = 1 / (2 * np.pi * 2000.0)
    beamf = SNNBeamformer(geometry, 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
    rng = np.random.RandomState(5)
    W = rng.randn(2 * M5, G5)
    W /= np.linalg.norm(W, axis=0, keepdims=True)
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * 2000 * time_test)
    doa = rng.rand(2) * 2 * np.pi
    time_in, clean = beamf.synthesize_batch((time_test, sig_test), doa)
    gen = torch.Generator(device=clean.device)
    gen.manual_seed(7)
    x = (clean + 0.5 * torch.randn(clean.shape, generator=gen, device=clean.device, dtype=torch.float64)).contiguous()
    assert x.shape == (2, 9599, M5)
    plan = beamf.plan()
    nir = neuron_impulse_response(time_in, beamf.tau_vec)
    assert len(nir) == 71 and beamf.spk_encoder.robust_width == 24 and len(beamf.kernel) == 960
    plan.set_neuron_kernel(nir)
    plan.set_bf_mat(W)
    full = plan.snn_pipeline(x, want_spikes=True, want_power=True)
    b, a = beamf.bandpass_filter
    for i in range(2):
        ref = O.snn_chain(x[i].cpu().numpy(), beamf.kernel, b, a, 24, True, nir, W, want=("spikes", "power"))
        np.testing.assert_array_equal(full["spikes"][i].cpu().numpy(), ref["spikes"])
        np.testing.assert_allclose(full["power"][i].cpu().numpy(), ref["power"], rtol=1e-12, atol=0)
        assert int(full["argmax"][i]) == ref["argmax"]


def test_config5_designed_bf_mat_end_to_end():
    """BASELINE config 5 with a DESIGNED beamforming matrix (micloc/snn_beamformer.py:183-203 for 64 microphones): the whole
    1440-DoA design on the device (chain, 128 x 128 membrane covariances, ONE launch of the one-sided Jacobi kernel) against
    the host decomposition (LAPACK, the reference's route) on a handful of DoAs -- columns equal up to the singular vector's
    unit phase, same beam pattern |W^H W| -- and the fused pipeline with that matrix against the oracle."""
    import torch

    from haghighatshoarmuir2024_amd.array_geometry import Random2DArray
    from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer, neuron_impulse_response

    fs, M5, G5 = 96_000, 64, 1440
    np.random.seed(1)
    geometry = Random2DArray(radius=0.2, num_mic=M5)
    tau = 1 / (2 * np.pi * 2000.0)
    beamf = SNNBeamformer(geometry, 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
    t = np.arange(0, 1.0, step=1 / fs)
    chirp = np.sin(2 * np.pi * np.cumsum(1000.0 + 1000.0 * (t % t[-1]) / t[-1]) / fs)  # target_snn_localization.py:345-356
    doa_list = np.linspace(-np.pi, np.pi, G5)
    Wd = beamf.design_from_template((t, chirp), doa_list, svd="device", doa_batch=48)
    assert Wd.shape == (2 * M5, G5) and np.all(np.isfinite(Wd))
    np.testing.assert_allclose(np.linalg.norm(Wd, axis=0), 1.0, rtol=0, atol=1e-12)
    pick = np.arange(0, G5, 131)  # 11 DoAs across the grid
    Wh = beamf.design_from_template((t, chirp), doa_list[pick], svd="host", device_synthesis=True, doa_batch=11)
    wd = Wd[:M5, pick] + 1j * Wd[M5:, pick]
    wh = Wh[:M5] + 1j * Wh[M5:]
    np.testing.assert_allclose(np.abs(np.sum(np.conj(wh) * wd, axis=0)), 1.0, rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.abs(wd.conj().T @ wd), np.abs(wh.conj().T @ wh), rtol=0, atol=1e-7)
    assert np.all(wd[0].real < 0) and np.max(np.abs(wd[0].imag)) < 1e-12  # LAPACK's phase convention

    rng = np.random.RandomState(6)
    time_test = np.arange(0, 100e-3, step=1 / fs)
    doa = rng.rand(2) * 2 * np.pi
    time_in, clean = beamf.synthesize_batch((time_test, np.sin(2 * np.pi * 2000 * time_test)), doa)
    gen = torch.Generator(device=clean.device)
    gen.manual_seed(8)
    x = (clean + 0.5 * torch.randn(clean.shape, generator=gen, device=clean.device, dtype=torch.float64)).contiguous()
    plan = beamf.plan()
    nir = neuron_impulse_response(time_in, beamf.tau_vec)
    plan.set_neuron_kernel(nir)
    plan.set_bf_mat(Wd)
    full = plan.snn_pipeline(x, want_spikes=True, want_power=True)
    b, a = beamf.bandpass_filter
    err = []
    for i in range(2):
        ref = O.snn_chain(x[i].cpu().numpy(), beamf.kernel, b, a, 24, True, nir, Wd, want=("spikes", "power"))
        np.testing.assert_array_equal(full["spikes"][i].cpu().numpy(), ref["spikes"])
        np.testing.assert_allclose(full["power"][i].cpu().numpy(), ref["power"], rtol=1e-12, atol=0)
        assert int(full["argmax"][i]) == ref["argmax"]
        err.append(abs(np.arcsin(abs(np.sin(doa_list[ref["argmax"]] - doa[i])))))
    assert max(err) < np.deg2rad(3.0)  # the designed matrix localises (the pi-periodic error of the script, :466)
