"""BASELINE config 4 (Xylo): spike encoding is pinned to the reference's own classes; the integer LIF is a
restatement of the published Xylo-A2 rule (PARITY UNPINNED: rockpool/xylosim is absent), so the tests check
HIP == oracle bit for bit, hand-computed known answers of that rule, and end-to-end plausibility."""
import numpy as np
import pytest

from conftest import golden
from oracle import oracle as O


def test_oracle_known_answers():
    # one neuron, weight 100, dash 2 / 2, threshold 150 (worked by hand in the commit that introduced this test)
    s = np.zeros((6, 1), np.uint8)
    s[0, 0] = s[1, 0] = 1
    out, rate = O.xylo_lif(s, np.array([[100]], np.int8), 0, 2, 2, 150)
    assert list(out.ravel()) == [0, 1, 1, 0, 1, 0] and int(rate[0]) == 3
    # "at least one LSB" decay: 3 -> 2 -> 1 -> 0 with dash 4, and the same towards zero for negative values
    s = np.zeros((5, 1), np.uint8)
    s[0, 0] = 1
    for wgt in (3, -3):
        out, rate = O.xylo_lif(s, np.array([[wgt]], np.int8), 0, 4, 15, 1000)
        assert int(rate[0]) == 0
    # saturation at +32767 and the per-step spike cap
    s = np.ones((400, 1), np.uint8) * 15
    out, rate = O.xylo_lif(s, np.array([[127]], np.int8), 0, 15, 15, 10, max_spikes=31)
    assert out.max() == 31 and out[-1, 0] == 31
    # shared recurrent weight: inhibition from the previous step's total spike count
    s = np.ones((50, 2), np.uint8)
    W = np.array([[60, 50, 40], [60, 50, 40]], np.int8)
    free, _ = O.xylo_lif(s, W, 0, 3, 3, 100)
    inhib, _ = O.xylo_lif(s, W, -20, 3, 3, 100)
    assert inhib.sum() < free.sum() and np.array_equal(inhib[0], free[0])


def test_quantised_specification(cfg2):
    from haghighatshoarmuir2024_amd.xylo_snn_localization import xylo_specification

    tau = 1 / (2 * np.pi * 1500.0)
    spec = xylo_specification([cfg2["bf_mat"]], [[tau, tau]], fs=48_000, target_dt=1e-3, bipolar_spikes=True)
    assert spec["W_in"].shape == (28, 449) and spec["W_in"].dtype == np.int8
    assert np.abs(spec["W_in"]).max() == 127
    np.testing.assert_array_equal(spec["W_in"][14:], -spec["W_in"][:14])
    # tau * fs/1000 / dt = 48 / (2 pi 1500) = 5.09 -> dash = round(log2(5.09)) = 2
    assert set(spec["dash_syn"]) == {2} and set(spec["dash_mem"]) == {2}
    # -0.1/449 vanishes under the global 8-bit scale; threshold 1.0 maps to round(scaling)
    assert spec["w_rec"] == 0
    assert spec["threshold"][0] == int(round(spec["scaling"])) and spec["threshold"].dtype == np.int16


def test_signal_from_template_formula():
    from micloc.array_geometry import CenterCircularArray
    from micloc.xylo_snn_localization import signal_from_template

    geo = CenterCircularArray(4.5e-2, 7)
    t = np.arange(0, 5e-3, 1 / 48_000)
    s = np.sin(2 * np.pi * 1500 * t)
    doa = np.linspace(0.3, 0.9, len(t))
    got = signal_from_template(geo, (t, s, doa))
    exp = np.stack([np.interp(t[i] + geo.delays(doa[i], normalized=False), t, s) for i in range(len(t))])
    np.testing.assert_array_equal(got, exp)
    got2 = signal_from_template(geo, (t, s, 0.5))
    np.testing.assert_array_equal(got2[10], np.interp(t[10] + geo.delays(0.5, normalized=False), t, s))


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,w_rec,B,T", [(449, 28, 0, 3, 700), (100, 28, -3, 2, 300), (1024, 5, 2, 1, 257), (70, 64, 0, 2, 513)])
def test_hip_xylo_lif_equals_oracle(N, Cin, w_rec, B, T):
    from haghighatshoarmuir2024_amd.xylo_snn_localization import xylo_lif

    rng = np.random.RandomState(N + Cin)
    spec = dict(W_in=rng.randint(-127, 128, size=(Cin, N)).astype(np.int8), w_rec=w_rec,
                dash_syn=rng.randint(0, 6, size=N).astype(np.uint8), dash_mem=rng.randint(0, 6, size=N).astype(np.uint8),
                threshold=rng.randint(50, 4000, size=N).astype(np.int16))
    spikes = (rng.rand(B, T, Cin) < 0.1).astype(np.uint8) * rng.randint(1, 4, size=(B, T, Cin)).astype(np.uint8)
    out, rate = xylo_lif(spikes, spec, max_spikes=31)
    out, rate = out.cpu().numpy(), rate.cpu().numpy()
    _, rate_only = xylo_lif(spikes, spec, max_spikes=31, want_spikes=False)
    for b in range(B):
        eo, er = O.xylo_lif(spikes[b], spec["W_in"], w_rec, spec["dash_syn"], spec["dash_mem"], spec["threshold"], 31)
        np.testing.assert_array_equal(out[b], eo)
        np.testing.assert_array_equal(rate[b], er)
    np.testing.assert_array_equal(rate_only.cpu().numpy(), rate)
    assert rate.sum() > 0


@pytest.mark.gpu
def test_demo_spike_encoding_pinned_and_end_to_end(cfg2):
    """spike_encoding == the reference's own STHT / ButterworthFilterbank / RZCC classes (golden); the whole Demo
    localises a clean chirp (plausibility only: the LIF stage is unpinned)."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.utils import find_peak_location
    from micloc.xylo_snn_localization import Demo, signal_from_template

    z = golden("filterbank.npz")
    geo = CenterCircularArray(4.5e-2, 7)
    doa_list = np.linspace(-np.pi, np.pi, 8 * 7 + 1)
    demo = Demo(geometry=geo, freq_bands=[[1000, 2000]], doa_list=doa_list, recording_duration=0.1, bipolar_spikes=True)
    spikes_in = demo.spike_encoding(z["sig_in"])
    assert spikes_in.dtype == np.int64 and spikes_in.shape == (3000, 28)
    np.testing.assert_array_equal(spikes_in.astype(np.int8), z["spikes_in"])
    # end to end on a 0.25 s chirp from a known direction, 20 dB SNR
    fs = 48_000
    t = np.arange(0, 0.25, 1 / fs)
    period = t[-1]
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % period) / period) / fs)
    rng = np.random.RandomState(0)
    errs = []
    for doa in (0.4, 2.0, -1.3):
        sig = signal_from_template(geo, (t, s, doa))
        sig = sig + np.sqrt(np.mean(sig**2) / 100) * rng.randn(*sig.shape)
        spk = demo.spike_encoding(sig)
        out = demo.xylo_process(spk)
        assert out.shape == (len(t), len(doa_list))
        rate = demo.extract_rate(out)
        rate_b = demo.rate_batch(sig[None])[0].cpu().numpy()
        np.testing.assert_allclose(rate_b, rate, rtol=1e-12)
        idx = find_peak_location(rate / rate.max(), win_size=3)
        errs.append(np.degrees(np.arcsin(abs(np.sin(doa_list[idx] - doa)))))
        assert demo.estimate_doa_from_rate(rate, "peak") == doa_list[np.argmax(rate)]
    assert max(errs) < 15.0, errs
    with pytest.raises(ValueError):
        demo.estimate_doa_from_rate(rate, "median")


@pytest.mark.gpu
@pytest.mark.parametrize(
    "C,N,T,thr_lo,thr_hi,max_spikes,dash_hi",
    [
        (14, 449, 700, 50, 4000, 31, 6),    # the sweep's network size; first spikes only
        (14, 449, 1037, 5, 60, 31, 3),      # thresholds far below the input current: many spikes per step (exact 32-bit path)
        (14, 449, 300, 5, 60, 1, 3),        # ... with the per-step cap at 1
        (14, 449, 300, 5, 60, 3, 3),        # ... and at 3
        (7, 17, 257, 100, 900, 31, 15),     # one neuron into the second column tile; decay shifts up to 15
        (1, 1, 16, 1, 2, 31, 2),            # one channel, one neuron, exactly one 16-step tile
        (20, 513, 255, 200, 3000, 31, 15),  # 40 input channels (two 32-channel k-steps), a second workgroup
        (32, 1100, 15, 30000, 32767, 31, 4),  # 64 channels, thresholds at the top of the range, shorter than a tile
        (14, 130, 1, 10, 50, 31, 2),        # a single step
    ],
)
def test_packed_xylo_kernel_equals_oracle(C, N, T, thr_lo, thr_hi, max_spikes, dash_hi):
    """The sweep's form of the integer LIF (ternary raster in, no recurrence: two neurons per lane in 16-bit halves, input
    currents on the int8 matrix cores) against the oracle, spike raster and counts, bit for bit."""
    import torch

    from haghighatshoarmuir2024_amd.xylo_snn_localization import XyloNetwork

    rng = np.random.RandomState(C * 1000 + N + T)
    Cin = 2 * C
    spec = dict(W_in=rng.randint(-127, 128, size=(Cin, N)).astype(np.int8), w_rec=0,
                dash_syn=rng.randint(0, dash_hi + 1, size=N).astype(np.uint8), dash_mem=rng.randint(0, dash_hi + 1, size=N).astype(np.uint8),
                threshold=rng.randint(thr_lo, thr_hi + 1, size=N).astype(np.int16))
    B = 3
    raster = rng.choice([-1, 0, 1], size=(B, T, C), p=[0.15, 0.7, 0.15]).astype(np.int8)
    raster[1] = rng.choice([-1, 1], size=(T, C))  # every channel fires at every step: the largest input currents
    events = np.concatenate([raster > 0, raster < 0], axis=2).astype(np.uint8)
    net = XyloNetwork(spec)
    out, rate = net.run(torch.from_numpy(raster).cuda(), ternary=True, want_spikes=True, max_spikes=max_spikes)
    _, rate_only = net.run(torch.from_numpy(raster).cuda(), ternary=True, want_spikes=False, max_spikes=max_spikes)
    assert torch.equal(rate, rate_only)
    for b in range(B):
        eo, er = O.xylo_lif(events[b], spec["W_in"], 0, spec["dash_syn"], spec["dash_mem"], spec["threshold"], max_spikes)
        np.testing.assert_array_equal(out[b].cpu().numpy(), eo)
        np.testing.assert_array_equal(rate[b].cpu().numpy(), er)
    assert int(rate.sum()) > 0 or thr_lo >= 30000 or N < 100


@pytest.mark.gpu
def test_resident_network_ternary_input_and_peak_location(cfg2):
    """The two-phase form (constants resident, int8 raster in, +/- split inside the kernel) equals the one-shot form and the
    oracle; the on-device find_peak_location equals the host function wherever the integer window sums have no tie."""
    import torch

    from haghighatshoarmuir2024_amd import runtime
    from haghighatshoarmuir2024_amd.utils import find_peak_location
    from haghighatshoarmuir2024_amd.xylo_snn_localization import XyloNetwork, xylo_lif, xylo_specification

    tau = 1 / (2 * np.pi * 1500)
    spec = xylo_specification([cfg2["bf_mat"]], [[tau, tau]], fs=48_000, target_dt=1e-3, bipolar_spikes=True)
    rng = np.random.RandomState(4)
    B, T, C = 3, 700, 14
    raster = rng.choice([-1, 0, 1], size=(B, T, C), p=[0.05, 0.9, 0.05]).astype(np.int8)
    events = np.concatenate([raster > 0, raster < 0], axis=2).astype(np.uint8)
    net = XyloNetwork(spec)
    out_t, rate_t = net.run(torch.from_numpy(raster).cuda(), ternary=True, want_spikes=True)
    out_e, rate_e = net.run(torch.from_numpy(events).cuda(), ternary=False, want_spikes=True)
    out_1, rate_1 = xylo_lif(events, spec)
    assert torch.equal(out_t, out_e) and torch.equal(rate_t, rate_e) and torch.equal(out_t, out_1) and torch.equal(rate_t, rate_1)
    for b in range(B):
        eo, er = O.xylo_lif(events[b], spec["W_in"], spec["w_rec"], spec["dash_syn"], spec["dash_mem"], spec["threshold"], 31)
        np.testing.assert_array_equal(out_t[b].cpu().numpy(), eo)
        np.testing.assert_array_equal(rate_t[b].cpu().numpy(), er)
    assert int(rate_t.sum()) > 0
    # a recurrent weight and other channel counts go through the other instantiations
    for cin, n, w_rec in ((3, 70, -3), (28, 449, -1), (40, 300, 0), (64, 1000, 2)):
        sp = dict(W_in=rng.randint(-127, 128, size=(cin, n)).astype(np.int8), w_rec=w_rec, dash_syn=np.full(n, 2, np.uint8),
                  dash_mem=np.full(n, 3, np.uint8), threshold=np.full(n, 300, np.int16))
        ev = (rng.rand(2, 150, cin) < 0.2).astype(np.uint8)
        o, r = XyloNetwork(sp).run(torch.from_numpy(ev).cuda(), want_spikes=True)
        for b in range(2):
            eo, er = O.xylo_lif(ev[b], sp["W_in"], w_rec, sp["dash_syn"], sp["dash_mem"], sp["threshold"], 31)
            np.testing.assert_array_equal(o[b].cpu().numpy(), eo)
            np.testing.assert_array_equal(r[b].cpu().numpy(), er)
    # peak location: random counts (ties are rare), a flat profile (all ties -> first maximum) and a peak at the wrap-around
    G = 449
    counts = rng.poisson(40, size=(6, G)).astype(np.int32)
    counts[1, 5:9] += 400
    counts[2, -3:] += 400
    counts[3, :] = 7
    counts[4, :] = 0
    idx = runtime.peak_location(torch.from_numpy(counts).cuda(), G, 15).cpu().numpy()
    for b in range(6):
        p = counts[b].astype(np.float64)
        want = find_peak_location(p, 15)  # exact in fp64 for these integers (window sums < 2^53, no normalisation)
        assert idx[b] == want, (b, idx[b], want)
    two = np.concatenate([counts, counts[:, ::-1]], axis=1)  # two bands: the counts are summed per DoA first
    idx2 = runtime.peak_location(torch.from_numpy(np.ascontiguousarray(two)).cuda(), G, 15).cpu().numpy()
    for b in range(6):
        assert idx2[b] == find_peak_location((two[b, :G] + two[b, G:]).astype(np.float64), 15)
    with pytest.raises(ValueError):
        runtime.peak_location(torch.from_numpy(counts).cuda(), G, 14)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C,N", [(5, 5000, 14, 449), (1300, 4500, 14, 360), (40, 2048, 7, 100), (3, 2049, 20, 512), (2, 100, 14, 449)])
def test_queued_sweep_equals_one_workgroup_per_trial(B, T, C, N):
    """micloc_xylo_lif_sweep_i16: persistent workgroups on the (trial, time chunk) ticket queue -- a trial's integer state handed
    from workgroup to workgroup at every 2048-step boundary, more trials than workers (1300 > 4 x 256), a chunk boundary one step
    before the end, two 32-channel k-steps -- gives the counts of the one-workgroup-per-trial kernel bit for bit (which
    test_packed_xylo_kernel_equals_oracle pins to the oracle), and the oracle's on the first trials."""
    import torch

    from haghighatshoarmuir2024_amd.xylo_snn_localization import XyloNetwork

    rng = np.random.RandomState(B + T + N)
    Cin = 2 * C
    spec = dict(W_in=rng.randint(-127, 128, size=(Cin, N)).astype(np.int8), w_rec=0,
                dash_syn=rng.randint(0, 5, size=N).astype(np.uint8), dash_mem=rng.randint(0, 5, size=N).astype(np.uint8),
                threshold=rng.randint(40, 3000, size=N).astype(np.int16))
    raster = torch.from_numpy(rng.choice([-1, 0, 1], size=(B, T, C), p=[0.1, 0.8, 0.1]).astype(np.int8)).cuda()
    net = XyloNetwork(spec)
    _, rate_q = net.run(raster, ternary=True, want_spikes=False)          # the queue
    Bs = min(B, 64)
    _, rate_w = net.run(raster[:Bs].contiguous(), ternary=True, want_spikes=True)  # one workgroup per trial (also stores the raster)
    assert torch.equal(rate_q[:Bs], rate_w)
    st = net.queue_status()
    assert st["gave_up"] == 0 and st["tickets"] >= B * -(-T // 2048)
    for _ in range(2):  # the scratch of a call is reset by the call itself: again, same result
        assert torch.equal(net.run(raster, ternary=True, want_spikes=False)[1], rate_q)
    assert net.queue_status()["gave_up"] == 0
    r = raster[:2].cpu().numpy()
    events = np.concatenate([r > 0, r < 0], axis=2).astype(np.uint8)
    for b in range(2):
        _, er = O.xylo_lif(events[b], spec["W_in"], 0, spec["dash_syn"], spec["dash_mem"], spec["threshold"], 31)
        np.testing.assert_array_equal(rate_q[b].cpu().numpy(), er)
    if B > 64:
        # every trial, against the general kernel (lane = neuron, 32-bit arithmetic)
        ev = torch.cat([(raster > 0), (raster < 0)], dim=2).to(torch.uint8).contiguous()
        _, rate_g = net.run(ev, ternary=False, want_spikes=False)
        assert torch.equal(rate_q, rate_g)
