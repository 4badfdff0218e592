#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by running the *real* reference.

Run in the build container only (the reference lives at /root/reference, which does not exist on the
GPU box):

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py rzcc_edge  # one fixture

Only *data* (inputs and the reference's outputs) is written; no reference source travels.
Every fixture records the seed / parameters that produced it so it can be regenerated.

Reference entry points exercised (paths relative to /root/reference):
  micloc/snn_beamformer.py   SNNBeamformer.__init__/design_from_template/apply_to_template/apply_to_signal
  micloc/beamformer.py       Beamformer.__init__/design_from_template/apply_to_template/apply_to_signal
  micloc/spike_encoder.py    ZeroCrossingSpikeEncoder.evolve
  micloc/array_geometry.py   CenterCircularArray / CircularArray / LinearArray / Random2DArray .delays
  micloc/utils.py            find_peak_location, Envelope
  micloc/filterbank.py       ButterworthFilterbank.evolve
and the Monte-Carlo loop of paper_plots/target_snn_localization.py:435-467 (restated here because the
script itself needs cvxpy/soundfile, which are not installed).
"""
import os
import sys
import io
import contextlib

REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not present: golden vectors can only be regenerated in the build container")
sys.path.insert(0, REF)
# make sure the in-repo drop-in `micloc` shim does not shadow the reference package
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))]

import numpy as np  # noqa: E402

import micloc  # noqa: E402

assert os.path.abspath(micloc.__file__ if micloc.__file__ else list(micloc.__path__)[0]).startswith(REF), micloc

from micloc.array_geometry import (  # noqa: E402
    CenterCircularArray,
    CircularArray,
    LinearArray,
    Random2DArray,
)
from micloc.snn_beamformer import SNNBeamformer  # noqa: E402
from micloc.beamformer import Beamformer  # noqa: E402
from micloc.spike_encoder import ZeroCrossingSpikeEncoder  # noqa: E402
from micloc.utils import find_peak_location, Envelope  # noqa: E402
from micloc.filterbank import ButterworthFilterbank  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# --------------------------------------------------------------------------------------------------
def cfg2_beamformer(bipolar=True):
    """paper_plots/target_snn_localization.py:319-342"""
    radius, num_mic, fs = 4.5e-2, 7, 48_000
    freq_design = 2_000
    freq_range = [0.5 * freq_design, freq_design]
    geometry = CenterCircularArray(radius=radius, num_mic=num_mic)
    tau = 1.0 / (2 * np.pi * freq_design)
    beamf = SNNBeamformer(
        geometry=geometry,
        kernel_duration=10.0e-3,
        tau_vec=np.asarray([tau, tau]),
        freq_range=freq_range,
        fs=fs,
        bipolar_spikes=bipolar,
    )
    return beamf, geometry, fs, freq_design, freq_range


def chirp_template(fs, freq_range, duration=1000e-3):
    """paper_plots/target_snn_localization.py:345-356"""
    time_temp = np.arange(0, duration, step=1 / fs)
    f_min, f_max = freq_range
    period = time_temp[-1]
    freq_inst = f_min + (f_max - f_min) * (time_temp % period) / period
    phase_inst = 2 * np.pi * np.cumsum(freq_inst) * 1 / fs
    return time_temp, np.sin(phase_inst)


def capture_chain(beamf, bf_mat, time_vec, sig):
    """run apply_to_signal and capture pre-encoder signal + spikes by wrapping the encoder."""
    cap = {}
    enc = beamf.spk_encoder
    orig = enc.evolve

    def wrapped(x):
        cap["pre_enc"] = np.array(x, copy=True)
        s = orig(x)
        cap["spikes"] = np.array(s, copy=True)
        return s

    enc.evolve = wrapped
    try:
        y = beamf.apply_to_signal(bf_mat=bf_mat, sig_in_vec=(time_vec, sig))
    finally:
        del enc.evolve
    cap["y"] = y
    return cap


# --------------------------------------------------------------------------------------------------
def gen_kat_init():
    out = {}
    for tag, fs, fr in [("48k", 48_000, [1000.0, 2000.0]), ("96k", 96_000, [1000.0, 2000.0]), ("48k_4k", 48_000, [2000.0, 4000.0])]:
        geometry = CenterCircularArray(radius=4.5e-2, num_mic=7)
        tau = 1.0 / (2 * np.pi * fr[1])
        bf = SNNBeamformer(geometry, 10.0e-3, fr, np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
        b, a = bf.bandpass_filter
        out[f"kernel_{tag}"] = bf.kernel
        out[f"b_{tag}"] = b
        out[f"a_{tag}"] = a
        out[f"robust_width_{tag}"] = np.int64(bf.spk_encoder.robust_width)
        out[f"kernel_length_{tag}"] = np.int64(bf.kernel_length)
        # neuron kernel exactly as apply_to_signal builds it (snn_beamformer.py:342-361), T = 0.1 s
        T = len(np.arange(0, 100e-3, step=1 / fs)) - 1
        time_vec = np.arange(T) / fs
        t = time_vec - time_vec[0]
        h = (t / tau) * np.exp(-t / tau)
        h = h / np.sum(h)
        n = int(np.sum(np.cumsum(h) < 0.999))
        out[f"nir_{tag}"] = h[:n]
        out[f"nir_T_{tag}"] = np.int64(T)
    # geometries
    g = CenterCircularArray(4.5e-2, 7)
    out["ccirc_r"], out["ccirc_theta"] = g.r_vec, g.theta_vec
    g = CircularArray(4.5e-2, 7)
    out["circ_r"], out["circ_theta"] = g.r_vec, g.theta_vec
    g = LinearArray(spacing=0.02, num_mic=8, radius=0.07)
    out["lin_r"], out["lin_theta"] = g.r_vec, g.theta_vec
    np.random.seed(1)
    g = Random2DArray(radius=0.2, num_mic=16)
    out["rand_r"], out["rand_theta"] = g.r_vec, g.theta_vec
    thetas = np.array([-3.0, -1.0, 0.0, 0.7, 2.5])
    for nm, geo in [("ccirc", CenterCircularArray(4.5e-2, 7)), ("lin", LinearArray(0.02, 8, 0.07))]:
        out[f"{nm}_delays_norm"] = np.stack([geo.delays(th, True) for th in thetas])
        out[f"{nm}_delays_raw"] = np.stack([geo.delays(th, False) for th in thetas])
    out["thetas"] = thetas
    # utils.find_peak_location
    rng = np.random.RandomState(7)
    p = rng.rand(449)
    out["fpl_in"] = p
    out["fpl_out"] = np.array([find_peak_location(p, 15), find_peak_location(p, 1), find_peak_location(p, 15, periodic=False)])
    env = Envelope(rise_time=1e-3, fall_time=20e-3, fs=48_000)
    x = rng.randn(300, 3)
    out["env_in"] = x
    out["env_out"] = env.evolve(x)
    save("kat_init.npz", **out)


def gen_bf_mat_chirp():
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    t, s = chirp_template(fs, fr)
    doa_list = np.linspace(-np.pi, np.pi, 64 * 7 + 1)
    # capture the complex covariances fed to SVD (for design parity up to phase)
    covs = []
    orig_svd = np.linalg.svd

    def svd_spy(mat, *a, **k):
        covs.append(np.array(mat, copy=True))
        return orig_svd(mat, *a, **k)

    np.linalg.svd = svd_spy
    try:
        bf_mat = quiet(beamf.design_from_template, template=(t, s), doa_list=doa_list)
    finally:
        np.linalg.svd = orig_svd
    covs = np.asarray(covs)
    sel = np.arange(0, 449, 32)
    save("bf_mat_chirp449_bipolar.npz", bf_mat=bf_mat, doa_list=doa_list, cov_sel=covs[sel], cov_idx=sel)


def gen_bf_mat_unipolar():
    """paper_plots/array_resolution_snn.py:100-164 (config 1)"""
    out = {}
    for f in (1000, 2000, 4000):
        geometry = CenterCircularArray(radius=4.5e-2, num_mic=7)
        fs = 48_000
        tau = 1 / (2 * np.pi * f)
        beamf = SNNBeamformer(geometry, 10e-3, [0.5 * f, 2 * f], np.asarray([tau, tau]), bipolar_spikes=False, fs=fs)
        time_temp = np.arange(0, 0.4, step=1 / fs)
        sig_temp = np.sin(2 * np.pi * f * time_temp)
        doa_list = np.linspace(-np.pi, np.pi, 32 * 7 + 1)
        covs = []
        orig_svd = np.linalg.svd

        def svd_spy(mat, *a, **k):
            covs.append(np.array(mat, copy=True))
            return orig_svd(mat, *a, **k)

        np.linalg.svd = svd_spy
        try:
            bf_mat = quiet(beamf.design_from_template, template=(time_temp, sig_temp), doa_list=doa_list)
        finally:
            np.linalg.svd = orig_svd
        out[f"bf_mat_f{f}"] = bf_mat
        sel = np.arange(0, 225, 32)
        out[f"cov_sel_f{f}"] = np.asarray(covs)[sel]
        out["cov_idx"] = sel
        out["doa_list"] = doa_list
    save("bf_mat_sin225_unipolar.npz", **out)


def gen_trials_cfg2():
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    z = np.load(os.path.join(OUT, "bf_mat_chirp449_bipolar.npz"))
    bf_mat, doa_list = z["bf_mat"], z["doa_list"]
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * fd * time_test)
    snr_db = 0 - 10 * np.log10((fs / 2) / (fr[1] - fr[0]))
    np.random.seed(1234)
    doas, sigs, spikes, powers, argmaxes, yrows = [], [], [], [], [], []
    pre0 = None
    row_idx = np.array([0, 1, 100, 239, 240, 241, 1000, 4798])
    # capture the noisy array signal by wrapping apply_to_signal
    for trial in range(3):
        doa = np.random.rand(1)[0] * 2 * np.pi
        cap_in = {}
        orig_apply = beamf.apply_to_signal

        def spy(bf_mat, sig_in_vec):
            cap_in["time"] = np.array(sig_in_vec[0], copy=True)
            cap_in["sig"] = np.array(sig_in_vec[1], copy=True)
            return capture_chain_inner(bf_mat, sig_in_vec)

        def capture_chain_inner(bf_mat_, sig_in_vec_):
            cap = capture_chain_cls(beamf, orig_apply, bf_mat_, sig_in_vec_)
            cap_in.update(cap)
            return cap["y"]

        beamf.apply_to_signal = spy
        try:
            y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, doa), snr_db=snr_db)
        finally:
            del beamf.apply_to_signal
        power = np.mean(np.abs(y) ** 2, axis=0)
        doas.append(doa)
        sigs.append(cap_in["sig"])
        spikes.append(cap_in["spikes"].astype(np.int8))
        powers.append(power)
        argmaxes.append(int(np.argmax(power)))
        yrows.append(y[row_idx])
        if trial == 0:
            pre0 = cap_in["pre_enc"]
            time0 = cap_in["time"]
    save(
        "trials_cfg2.npz",
        seed=np.int64(1234),
        snr_db=np.float64(snr_db),
        doa=np.asarray(doas),
        sig_in=np.asarray(sigs),
        spikes=np.asarray(spikes),
        power=np.asarray(powers),
        argmax=np.asarray(argmaxes),
        y_rows=np.asarray(yrows),
        row_idx=row_idx,
        pre_enc0_head=pre0[:1200],
        pre_enc0_tail=pre0[-300:],
        time0=time0,
    )


def capture_chain_cls(beamf, apply_fn, bf_mat, sig_in_vec):
    cap = {}
    enc = beamf.spk_encoder
    orig = enc.evolve

    def wrapped(x):
        cap["pre_enc"] = np.array(x, copy=True)
        s = orig(x)
        cap["spikes"] = np.array(s, copy=True)
        return s

    enc.evolve = wrapped
    try:
        cap["y"] = apply_fn(bf_mat=bf_mat, sig_in_vec=sig_in_vec)
    finally:
        del enc.evolve
    return cap


def gen_sweep_seed0():
    """paper_plots/target_snn_localization.py:435-467 with 3 of the 11 SNRs."""
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    z = np.load(os.path.join(OUT, "bf_mat_chirp449_bipolar.npz"))
    bf_mat, doa_list = z["bf_mat"], z["doa_list"]
    snr_gain = (fs / 2) / (fr[1] - fr[0])
    snr_db_vec = np.array([-10.0, 5.0, 20.0])
    num_sim = 100
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * fd * time_test)
    np.random.seed(0)
    doa = np.zeros((3, num_sim))
    amax = np.zeros((3, num_sim), dtype=np.int64)
    err = np.zeros((3, num_sim))
    pmax = np.zeros((3, num_sim))
    for i, snr_db in enumerate(snr_db_vec):
        snr_t = snr_db - 10 * np.log10(snr_gain)
        for sim in range(num_sim):
            d = np.random.rand(1)[0] * 2 * np.pi
            y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, d), snr_db=snr_t)
            power = np.mean(np.abs(y) ** 2, axis=0)
            k = int(np.argmax(power))
            doa[i, sim], amax[i, sim], pmax[i, sim] = d, k, power[k]
            err[i, sim] = np.arcsin(np.abs(np.sin(doa_list[k] - d)))
    save("sweep_seed0.npz", seed=np.int64(0), snr_db_vec=snr_db_vec, doa=doa, argmax=amax, err=err, pmax=pmax,
         mae_deg=np.mean(err, axis=1) * 180 / np.pi)


def gen_sweep_full():
    """The complete accuracy sweep of paper_plots/target_snn_localization.py:435-467: 11 SNRs x 100 trials, seed 0."""
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    z = np.load(os.path.join(OUT, "bf_mat_chirp449_bipolar.npz"))
    bf_mat, doa_list = z["bf_mat"], z["doa_list"]
    snr_gain = (fs / 2) / (fr[1] - fr[0])
    snr_db_vec = np.linspace(-10, 20, 11)
    num_sim = 100
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * fd * time_test)
    np.random.seed(0)
    shape = (len(snr_db_vec), num_sim)
    doa, amax, err, pmax = np.zeros(shape), np.zeros(shape, dtype=np.int64), np.zeros(shape), np.zeros(shape)
    for i, snr_db in enumerate(snr_db_vec):
        snr_t = snr_db - 10 * np.log10(snr_gain)
        for sim in range(num_sim):
            d = np.random.rand(1)[0] * 2 * np.pi
            y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, d), snr_db=snr_t)
            power = np.mean(np.abs(y) ** 2, axis=0)
            k = int(np.argmax(power))
            doa[i, sim], amax[i, sim], pmax[i, sim] = d, k, power[k]
            err[i, sim] = np.arcsin(np.abs(np.sin(doa_list[k] - d)))
    save("sweep_full_seed0.npz", seed=np.int64(0), snr_db_vec=snr_db_vec, doa=doa, argmax=amax, err=err, pmax=pmax,
         mae_deg=np.mean(err, axis=1) * 180 / np.pi)


def gen_rzcc_edge():
    rng = np.random.RandomState(42)
    cases = {}

    def add(name, x, w, bipolar):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        enc = ZeroCrossingSpikeEncoder(fs=48_000, robust_width=w, bipolar=bipolar)
        s = enc.evolve(x)
        cases[f"{name}__in"] = x
        cases[f"{name}__w"] = np.int64(w)
        cases[f"{name}__bip"] = np.int64(bipolar)
        cases[f"{name}__out"] = s.astype(np.int8)

    t = np.arange(600)
    add("sine_w12_bip", np.sin(2 * np.pi * t / 31.7), 12, True)
    add("sine_w12_uni", np.sin(2 * np.pi * t / 31.7), 12, False)
    add("noise_w1_bip", rng.randn(500, 3), 1, True)
    add("noise_w2_bip", rng.randn(500, 3), 2, True)
    add("noise_w12_bip", rng.randn(800, 4), 12, True)
    add("noise_w24_uni", rng.randn(800, 2), 24, False)
    add("noise_w200_bip", rng.randn(700, 2), 200, True)
    # exact zeros at the start (design path: clamped interpolation repeats s[0] = 0) -> plateaus
    x = np.concatenate([np.zeros(40), np.sin(2 * np.pi * np.arange(400) / 29.3)])
    add("leading_zeros", x, 12, True)
    # integer-valued input -> exact ties in the cumsum priorities and flat plateaus
    add("int_ties", rng.randint(-2, 3, size=(600, 3)).astype(float), 5, True)
    add("int_ties_w12", rng.randint(-1, 2, size=(900, 2)).astype(float), 12, True)
    # alternating sign: a peak every 2 samples, one huge cluster
    add("alternating", np.where(np.arange(300) % 2 == 0, 1.0, -1.0), 12, True)
    add("alternating_drift_up", np.where(np.arange(400) % 2 == 0, 1.5, -1.0), 12, True)
    add("alternating_drift_dn", np.where(np.arange(400) % 2 == 0, 1.0, -1.5), 7, True)
    # peaks exactly `distance` apart: period-8 square pattern with w = 8 and w = 9
    sq = np.tile(np.array([1, 1, 1, 1, -1, -1, -1, -1.0]), 40)
    add("period8_w8", sq, 8, True)
    add("period8_w9", sq, 9, True)
    # tiny inputs
    add("len1", np.array([1.0]), 3, True)
    add("len2", np.array([1.0, -1.0]), 3, True)
    add("len3", np.array([1.0, -1.0, 1.0]), 3, True)
    add("len4", np.array([1.0, -1.0, -1.0, 2.0]), 1, True)
    add("all_zero", np.zeros((50, 2)), 12, True)
    add("const_pos", np.ones(64), 12, True)
    # plateau reaching the last sample, plateau in the middle
    add("plateau_end", np.array([1, 1, 0, 0, 0, 0.0]), 2, True)
    add("plateau_mid", np.array([1, 1, 0, 0, 0, -1, -1, 1, 0, 0, 1, -3, 0, 0, 0, 0, 2.0]), 2, True)
    # absorbed increments: |x| < ulp(cumsum)/2 makes plateaus although x != 0
    x = np.array([1e16, 1.0, 1.0, -4.0, 1.0, 1.0, 3.0, -1e16, 5.0, -1.0, -1.0])
    add("absorbed", x, 1, True)
    # random walk with drift (long increasing chains of close peaks)
    add("drift_noise", 0.3 + rng.randn(1500, 2), 12, True)
    add("drift_noise_neg", -0.3 + rng.randn(1500, 2), 6, True)
    save("rzcc_edge.npz", **cases)


def gen_beamformer_c128():
    """paper_plots/target_localization.py:310-372 (G = 8*7+1 = 57) and one noisy trial."""
    radius, num_mic, fs = 4.5e-2, 7, 48_000
    freq_design = 2_000
    freq_range = [0.5 * freq_design, freq_design]
    geometry = CenterCircularArray(radius=radius, num_mic=num_mic)
    beamf = Beamformer(geometry=geometry, kernel_duration=10.0e-3, freq_range=freq_range, fs=fs)
    t, s = chirp_template(fs, freq_range)
    doa_list = np.linspace(-np.pi, np.pi, 8 * num_mic + 1)
    bf_mat, cov_list = quiet(beamf.design_from_template, template=(t, s), doa_list=doa_list)
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * freq_design * time_test)
    np.random.seed(99)
    doa = np.random.rand(1)[0] * 2 * np.pi
    cap = {}
    orig = beamf.apply_to_signal

    def spy(bf_mat, sig_in):
        cap["sig"] = np.array(sig_in, copy=True)
        return orig(bf_mat=bf_mat, sig_in=sig_in)

    beamf.apply_to_signal = spy
    try:
        y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, doa), snr_db=3.0)
    finally:
        del beamf.apply_to_signal
    power = np.mean(np.abs(y) ** 2, axis=0)
    row_idx = np.array([0, 1, 239, 240, 241, 1000, 4798])
    # interference-removal design on a small grid as well
    bf_mat_ir, _ = quiet(beamf.design_from_template, template=(t[:9600], s[:9600]), doa_list=doa_list[::4], interference_removal=True)
    save("beamformer_c128.npz", bf_mat=bf_mat, cov_list=np.asarray(cov_list), doa_list=doa_list, doa=np.float64(doa), sig_in=cap["sig"],
         y_rows=y[row_idx], row_idx=row_idx, power=power, argmax=np.int64(np.argmax(power)), snr_db=np.float64(3.0),
         bf_mat_ir=bf_mat_ir)


def gen_wide_case():
    """Generic-shape case: 16-mic random array, 96 kHz (L = 960, w = 24), G = 90, T = 2000."""
    np.random.seed(1)
    geometry = Random2DArray(radius=0.2, num_mic=16)
    fs = 96_000
    fr = [1000.0, 2000.0]
    tau = 1 / (2 * np.pi * fr[1])
    beamf = SNNBeamformer(geometry, 10e-3, fr, [tau, tau], bipolar_spikes=True, fs=fs)
    rng = np.random.RandomState(5)
    G = 90
    bf_mat = rng.randn(32, G)
    bf_mat /= np.linalg.norm(bf_mat, axis=0, keepdims=True)
    T = 2000
    time_vec = np.arange(T) / fs
    d = geometry.delays(1.234, normalized=True)
    sig = np.sin(2 * np.pi * 1500 * (time_vec.reshape(-1, 1) - d.reshape(1, -1))) + 0.5 * rng.randn(T, 16)
    cap = capture_chain(beamf, bf_mat, time_vec, sig)
    y = cap["y"]
    power = np.mean(np.abs(y) ** 2, axis=0)
    row_idx = np.array([0, 479, 480, 481, 1999])
    save("wide_case.npz", r_vec=geometry.r_vec, theta_vec=geometry.theta_vec, bf_mat=bf_mat, sig_in=sig, time_vec=time_vec,
         spikes=cap["spikes"].astype(np.int8), pre_enc_head=cap["pre_enc"][:1100], power=power, argmax=np.int64(np.argmax(power)),
         y_rows=y[row_idx], row_idx=row_idx, fs=np.int64(fs))


def gen_unipolar_trial():
    f = 2000
    geometry = CenterCircularArray(radius=4.5e-2, num_mic=7)
    fs = 48_000
    tau = 1 / (2 * np.pi * f)
    beamf = SNNBeamformer(geometry, 10e-3, [0.5 * f, 2 * f], np.asarray([tau, tau]), bipolar_spikes=False, fs=fs)
    z = np.load(os.path.join(OUT, "bf_mat_sin225_unipolar.npz"))
    bf_mat = z["bf_mat_f2000"]
    time_test = np.arange(0, 50e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * f * time_test)
    np.random.seed(77)
    doa = 0.4321
    cap_in = {}
    orig_apply = beamf.apply_to_signal

    def spy(bf_mat, sig_in_vec):
        cap_in["sig"] = np.array(sig_in_vec[1], copy=True)
        cap = capture_chain_cls(beamf, orig_apply, bf_mat, sig_in_vec)
        cap_in.update(cap)
        return cap["y"]

    beamf.apply_to_signal = spy
    try:
        y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, doa), snr_db=10.0)
    finally:
        del beamf.apply_to_signal
    power = np.mean(np.abs(y) ** 2, axis=0)
    save("unipolar_trial.npz", doa=np.float64(doa), snr_db=np.float64(10.0), seed=np.int64(77), sig_in=cap_in["sig"],
         spikes=cap_in["spikes"].astype(np.int8), power=power, argmax=np.int64(np.argmax(power)))


def gen_synth():
    """Noise-free synthesis (snn_beamformer.py:239-267, beamformer.py:220-245) incl. a moving DoA."""
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    time_test = np.arange(0, 20e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * fd * time_test) * np.hanning(len(time_test))
    out = {}
    for name, doa in [("fixed", 1.2345), ("moving", np.linspace(0.2, 2.9, len(time_test)))]:
        cap = {}

        def spy(bf_mat, sig_in_vec):
            cap["time"] = np.array(sig_in_vec[0], copy=True)
            cap["sig"] = np.array(sig_in_vec[1], copy=True)
            return np.zeros((1, 1))

        beamf.apply_to_signal = spy
        try:
            # snr_db = +inf dB is not possible; use 300 dB (noise 1e-15 relative) and fixed seed
            np.random.seed(3)
            beamf.apply_to_template(bf_mat=np.zeros((14, 3)), template=(time_test, sig_test, doa), snr_db=3000.0)
        finally:
            del beamf.apply_to_signal
        out[f"{name}_sig"] = cap["sig"]
        out[f"{name}_time"] = cap["time"]
        out[f"{name}_doa"] = np.asarray(doa, dtype=np.float64)
    out["time_test"] = time_test
    out["sig_test"] = sig_test
    save("synth.npz", **out)


def gen_filterbank():
    """ButterworthFilterbank(order=1) + STHT + RZCC as in Demo.spike_encoding
    (micloc/xylo_snn_localization.py:315-356), restated from the reference's own components because
    that module cannot be imported here (rockpool is not installed)."""
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    fb = ButterworthFilterbank(freq_bands=[[1000, 2000]], order=1, fs=fs)
    rng = np.random.RandomState(11)
    T = 3000
    sig = np.sin(2 * np.pi * 1600 * np.arange(T).reshape(-1, 1) / fs + np.arange(7).reshape(1, -1)) + 0.3 * rng.randn(T, 7)
    from scipy.signal import lfilter

    sig_h = np.roll(sig, beamf.kernel_length // 2, axis=0) + 1j * lfilter(beamf.kernel, [1], sig, axis=0)
    sig_real = np.hstack([np.real(sig_h), np.imag(sig_h)])
    filt = fb.evolve(sig_real)[0]
    spikes = beamf.spk_encoder.evolve(filt).astype(np.int64)
    pos = (spikes > 0).astype(np.int64)
    neg = (spikes < 0).astype(np.int64)
    spikes_in = np.hstack([pos, neg])
    b, a = fb.ba_list[0]
    save("filterbank.npz", sig_in=sig, b=b, a=a, filt_head=filt[:800], spikes_in=spikes_in.astype(np.int8))


def gen_speech():
    """Speech configuration (paper_plots/target_snn_localization.py:148-154, 213-245): the LibriSpeech utterance
    resampled to 48 kHz by np.interp, one noisy trial through the reference.  The FLAC file is decoded with the
    in-repo decoder (soundfile is not installed); the decoder checks the STREAMINFO MD5, so the PCM is exactly what
    libsndfile would return.  The PCM is committed as a data fixture, the 4.6 MB spike raster as its SHA-256."""
    import hashlib
    import importlib.util

    spec = importlib.util.spec_from_file_location("flac", os.path.join(OUT, "..", "..", "haghighatshoarmuir2024_amd", "flac.py"))
    flac = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(flac)
    path = os.path.join(REF, "paper_plots", "84-121123-0020.flac")
    pcm, rate, bps = flac.decode(open(path, "rb").read())
    assert rate == 16000 and bps == 16 and pcm.shape == (110720, 1)
    sig_test = pcm[:, 0].astype(np.float64) / 32768.0
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    z = np.load(os.path.join(OUT, "bf_mat_chirp449_bipolar.npz"))
    bf_mat, doa_list = z["bf_mat"], z["doa_list"]
    time_test = np.arange(len(sig_test)) / rate
    time_fs = np.linspace(time_test[0], time_test[-1], int(len(sig_test) / rate * fs))
    sig_fs = np.interp(time_fs, time_test, sig_test)
    np.random.seed(5)
    doa = np.random.rand(1)[0] * 2 * np.pi
    cap_in = {}
    orig_apply = beamf.apply_to_signal

    def spy(bf_mat, sig_in_vec):
        cap = capture_chain_cls(beamf, orig_apply, bf_mat, sig_in_vec)
        cap_in.update(cap)
        return cap["y"]

    beamf.apply_to_signal = spy
    try:
        y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_fs, sig_fs, doa), snr_db=10.0)
    finally:
        del beamf.apply_to_signal
    power = np.mean(np.abs(y) ** 2, axis=0)
    spikes = cap_in["spikes"].astype(np.int8)
    row_idx = np.array([0, 240, 1000, 100000, 332156])
    save("speech_trial.npz", pcm16=pcm[:, 0].astype(np.int16), rate=np.int64(rate), seed=np.int64(5), snr_db=np.float64(10.0), doa=np.float64(doa),
         T=np.int64(y.shape[0]), power=power, argmax=np.int64(np.argmax(power)), spikes_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(spikes).tobytes()).digest(), dtype=np.uint8),
         spikes_head=spikes[:3000], n_spikes=np.int64((spikes != 0).sum()), y_rows=y[row_idx], row_idx=row_idx)


def gen_speech_sweep():
    """The speech accuracy sweep of paper_plots/target_snn_localization.py:213-245 (no bandwidth correction of the SNR,
    `snr_db_target = snr_db` at :227), 3 of the 11 SNRs x 2 trials with the reference's draw order
    (rand(1) then randn(T, M) inside apply_to_template).  Inputs are rebuilt from speech_trial.npz's PCM."""
    z0 = np.load(os.path.join(OUT, "speech_trial.npz"))
    rate = int(z0["rate"])
    sig_test = z0["pcm16"].astype(np.float64) / 32768.0
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    z = np.load(os.path.join(OUT, "bf_mat_chirp449_bipolar.npz"))
    bf_mat, doa_list = z["bf_mat"], z["doa_list"]
    time_test = np.arange(len(sig_test)) / rate
    time_fs = np.linspace(time_test[0], time_test[-1], int(len(sig_test) / rate * fs))
    sig_fs = np.interp(time_fs, time_test, sig_test)
    snr_db_vec = np.array([-10.0, 5.0, 20.0])
    num_sim = 2
    seed = 11
    np.random.seed(seed)
    shape = (len(snr_db_vec), num_sim)
    doa, amax, err, pmax = np.zeros(shape), np.zeros(shape, dtype=np.int64), np.zeros(shape), np.zeros(shape)
    for i, snr_db in enumerate(snr_db_vec):
        for sim in range(num_sim):
            d = np.random.rand(1)[0] * 2 * np.pi
            y = quiet(beamf.apply_to_template, bf_mat=bf_mat, template=(time_fs, sig_fs, d), snr_db=snr_db)
            power = np.mean(np.abs(y) ** 2, axis=0)
            k = int(np.argmax(power))
            doa[i, sim], amax[i, sim], pmax[i, sim] = d, k, power[k]
            err[i, sim] = np.arcsin(np.abs(np.sin(doa_list[k] - d)))
            del y
    save("speech_sweep.npz", seed=np.int64(seed), snr_db_vec=snr_db_vec, num_sim=np.int64(num_sim), doa=doa, argmax=amax, err=err, pmax=pmax,
         T=np.int64(len(time_fs) - 1))


def _reference_function(relpath, name, namespace):
    """Execute ONE function definition of a reference file whose module cannot be imported here (its other imports --
    rockpool, cvxpy, soundfile -- are not installed).  The source is read from /root/reference at generation time and
    never stored; only the function's outputs go into the fixture."""
    import ast

    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fn) == 1, (relpath, name)
    mod = ast.Module(body=fn, type_ignores=[])
    exec(compile(mod, os.path.join(REF, relpath), "exec"), namespace)
    return namespace[name]


def gen_synth_xylo():
    """Array-signal synthesis of the Xylo sweep (micloc/xylo_snn_localization.py:44-71 `signal_from_template`: t + delays,
    no min-shift, no clamp) and of the multi-target figures (paper_plots/multiple_targets_snn.py:87-160
    `signal_multiple_targets`), evaluated by the reference's own function bodies."""
    from numbers import Number
    from typing import List, Tuple

    from micloc.array_geometry import ArrayGeometry

    ns = dict(np=np, Number=Number, Tuple=Tuple, List=List, ArrayGeometry=ArrayGeometry)
    sft = _reference_function("micloc/xylo_snn_localization.py", "signal_from_template", dict(ns))
    smt = _reference_function("paper_plots/multiple_targets_snn.py", "signal_multiple_targets", dict(ns))
    geometry = CenterCircularArray(radius=4.5e-2, num_mic=7)
    fs = 48_000
    t = np.arange(0, 25e-3, step=1 / fs)
    period = t[-1]
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % period) / period) / fs)
    out = dict(time=t, sig=s)
    out["fixed_doa"] = np.float64(2.3456)
    out["fixed_sig"] = sft(geometry=geometry, template=(t, s, 2.3456))
    mv = np.linspace(-1.0, 2.5, len(t))
    out["moving_doa"] = mv
    out["moving_sig"] = sft(geometry=geometry, template=(t, s, mv))
    doa_ts = np.stack([np.linspace(0.3, 1.1, len(t)), np.full(len(t), -2.0), 2.0 + 0.5 * np.sin(2 * np.pi * 40 * t)], axis=1)
    pow_ts = np.stack([np.ones(len(t)), 0.5 + 0.5 * (t > 10e-3), np.linspace(0.0, 2.0, len(t))], axis=1)
    out["multi_doa"] = doa_ts
    out["multi_power"] = pow_ts
    out["multi_sig"] = smt(geometry, t, s, doa_ts, pow_ts)
    save("synth_xylo.npz", **out)


def gen_bf_mat_unipolar_hf():
    """The two remaining design frequencies of config 1 (paper_plots/array_resolution_snn.py:118-146): 3.6 kHz (robust width 3,
    band [1800, 7200]) and 8 kHz (robust width 1, band [4000, 16000]: every local extremum of the running sum is a spike)."""
    out = {}
    for f in (3600, 8000):
        geometry = CenterCircularArray(radius=4.5e-2, num_mic=7)
        fs = 48_000
        tau = 1 / (2 * np.pi * f)
        beamf = SNNBeamformer(geometry, 10e-3, [f / 2, 2 * f], [tau, tau], bipolar_spikes=False, fs=fs)
        time_temp = np.arange(0, 0.4, step=1 / fs)
        sig_temp = np.sin(2 * np.pi * f * time_temp)
        doa_list = np.linspace(-np.pi, np.pi, 32 * 7 + 1)
        covs = []
        orig_svd = np.linalg.svd

        def svd_spy(mat, *a, **k):
            covs.append(np.array(mat, copy=True))
            return orig_svd(mat, *a, **k)

        np.linalg.svd = svd_spy
        try:
            bf_mat = quiet(beamf.design_from_template, template=(time_temp, sig_temp), doa_list=doa_list)
        finally:
            np.linalg.svd = orig_svd
        assert len(covs) == 225
        out[f"bf_mat_f{f}"] = bf_mat
        sel = np.arange(0, 225, 32)
        out[f"cov_sel_f{f}"] = np.asarray(covs)[sel]
        out[f"robust_width_f{f}"] = np.int64(beamf.spk_encoder.robust_width)
        out["cov_idx"] = sel
        out["doa_list"] = doa_list
    save("bf_mat_sin225_unipolar_hf.npz", **out)


def gen_beamformer_c128_g449():
    """the complex Beamformer at the sweep's own grid, G = 64*7+1 = 449 (micloc/beamformer.py:73-192 design from the
    1 s chirp, :194-292 one noisy trial)."""
    radius, num_mic, fs = 4.5e-2, 7, 48_000
    freq_design = 2_000
    freq_range = [0.5 * freq_design, freq_design]
    geometry = CenterCircularArray(radius=radius, num_mic=num_mic)
    beamf = Beamformer(geometry=geometry, kernel_duration=10.0e-3, freq_range=freq_range, fs=fs)
    t, s = chirp_template(fs, freq_range)
    doa_list = np.linspace(-np.pi, np.pi, 64 * num_mic + 1)
    bf_mat, cov_list = quiet(beamf.design_from_template, template=(t, s), doa_list=doa_list)
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * freq_design * time_test)
    np.random.seed(4490)
    doa = np.random.rand(1)[0] * 2 * np.pi
    cap = {}
    orig = beamf.apply_to_signal

    def spy(bf_mat, sig_in):
        cap["sig"] = np.array(sig_in, copy=True)
        return orig(bf_mat=bf_mat, sig_in=sig_in)

    beamf.apply_to_signal = spy
    try:
        y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, doa), snr_db=0.0)
    finally:
        del beamf.apply_to_signal
    power = np.mean(np.abs(y) ** 2, axis=0)
    row_idx = np.array([0, 1, 239, 240, 241, 1000, 4798])
    sel = np.arange(0, 449, 64)
    save("beamformer_c128_g449.npz", bf_mat=bf_mat, cov_sel=np.asarray(cov_list)[sel], cov_idx=sel, doa_list=doa_list, doa=np.float64(doa),
         seed=np.int64(4490), sig_in=cap["sig"], y_rows=y[row_idx], row_idx=row_idx, power=power, argmax=np.int64(np.argmax(power)),
         snr_db=np.float64(0.0))


def stress_bf_mat(C=128, G=1440, seed=5):
    """The stress case's seeded unit-norm bf_mat.  Uniform draws and an element-wise column norm: no libm call and no BLAS
    reduction in it, so tests rebuild the same bits from the seed on any machine (the fixture stores its SHA-256)."""
    W = np.random.RandomState(seed).random_sample((C, G)) - 0.5
    return W / np.sqrt(np.add.reduce(W * W, axis=0))


def gen_stress_case():
    """SURVEY 8c.7, BASELINE config 5 at its real shape: Random2DArray(0.2, 64) after np.random.seed(1), 96 kHz (L = 960, robust
    width 24, 71-tap neuron kernel), G = 1440, one noisy 0.1 s trial: the reference's apply_to_template synthesises the 2 kHz
    sine at the array and adds noise (its own draw order); the samples are then quantised like a 16-bit converter (steps of
    2^-12, stored as int16) and that array goes through the reference's apply_to_signal."""
    import hashlib

    np.random.seed(1)
    geometry = Random2DArray(radius=0.2, num_mic=64)
    fs = 96_000
    fr = [1000.0, 2000.0]
    tau = 1 / (2 * np.pi * fr[1])
    beamf = SNNBeamformer(geometry, 10e-3, fr, [tau, tau], bipolar_spikes=True, fs=fs)
    G = 1440
    bf_mat = stress_bf_mat(128, G, 5)
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * 2000 * time_test)
    doa = 2.2222
    cap = {}

    def spy(bf_mat, sig_in_vec):
        cap["time"] = np.array(sig_in_vec[0], copy=True)
        cap["sig"] = np.array(sig_in_vec[1], copy=True)
        return np.zeros((1, 1))

    beamf.apply_to_signal = spy
    try:
        np.random.seed(64)
        beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, doa), snr_db=0.0)
    finally:
        del beamf.apply_to_signal
    q = np.rint(cap["sig"] * 4096.0)
    assert np.abs(q).max() < 32768
    sig_q = q.astype(np.int16)
    sig = sig_q.astype(np.float64) / 4096.0
    time_vec = cap["time"]
    assert sig.shape == (9599, 64)
    # noise-free synthesis rows of the 64-microphone random array (pins delays + interpolation at this geometry)
    clean = {}

    def spy2(bf_mat, sig_in_vec):
        clean["sig"] = np.array(sig_in_vec[1], copy=True)
        return np.zeros((1, 1))

    beamf.apply_to_signal = spy2
    try:
        np.random.seed(64)
        beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, doa), snr_db=3000.0)
    finally:
        del beamf.apply_to_signal
    out = capture_chain(beamf, bf_mat, time_vec, sig)
    y = out["y"]
    assert y.shape == (9599, G)
    power = np.mean(np.abs(y) ** 2, axis=0)
    row_idx = np.array([0, 479, 480, 481, 5000, 9598])
    pre_idx = np.concatenate([np.arange(0, 64), np.arange(470, 500), np.arange(950, 1000), np.arange(9599 - 32, 9599)])
    clean_idx = np.array([0, 1, 50, 100, 1000, 9598])
    spikes = out["spikes"].astype(np.int8)
    save("stress_case.npz", r_vec=geometry.r_vec, theta_vec=geometry.theta_vec, fs=np.int64(fs), doa=np.float64(doa), geometry_seed=np.int64(1),
         noise_seed=np.int64(64), snr_db=np.float64(0.0), bf_seed=np.int64(5), G=np.int64(G),
         bf_mat_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(bf_mat).tobytes()).digest(), dtype=np.uint8),
         sig_q=sig_q, sig_scale=np.float64(1 / 4096.0), time_vec=time_vec, spikes=spikes, n_spikes=np.int64((spikes != 0).sum()),
         pre_enc_rows=out["pre_enc"][pre_idx], pre_idx=pre_idx, power=power, argmax=np.int64(np.argmax(power)), y_rows=y[row_idx], row_idx=row_idx,
         clean_rows=clean["sig"][clean_idx], clean_idx=clean_idx)


def _design_with_covs(beamf, template, doa_list):
    covs = []
    orig_svd = np.linalg.svd

    def svd_spy(mat, *a, **k):
        covs.append(np.array(mat, copy=True))
        return orig_svd(mat, *a, **k)

    np.linalg.svd = svd_spy
    try:
        bf_mat = quiet(beamf.design_from_template, template=template, doa_list=doa_list)
    finally:
        np.linalg.svd = orig_svd
    return bf_mat, np.asarray(covs)


def gen_design_other_geometries():
    """Two more callers of design_from_template from SURVEY 8b's call surface, on the geometries they use:
    paper_plots/array_resolution_random_snn.py:100-170 (np.random.seed(1), Random2DArray(4.5e-2, 13): 26 channels, 833 DoAs shifted by
    pi, 0.6 s sine, unipolar) and paper_plots/array_resolution_linear_snn.py:120-190 (LinearArray(2 r / 7, 7, r), DoAs in [0, pi], a sine
    with 1 % frequency jitter -- drawn here after np.random.seed(1) and rounded to float32 so that the fixture stores the template in
    half the bytes and the tests feed the reference's exact input)."""
    fs, f = 48_000, 2000
    tau = 1 / (2 * np.pi * f)
    out = {}
    np.random.seed(1)
    geometry = Random2DArray(radius=4.5e-2, num_mic=13)
    beamf = SNNBeamformer(geometry, 10e-3, [f / 2, 2 * f], [tau, tau], bipolar_spikes=False, fs=fs)
    time_temp = np.arange(0, 0.6, step=1 / fs)
    doa_list = np.linspace(-np.pi, np.pi, 64 * 13 + 1) + np.pi
    bf_mat, covs = _design_with_covs(beamf, (time_temp, np.sin(2 * np.pi * f * time_temp)), doa_list)
    assert covs.shape == (833, 26, 26)
    sel = np.arange(0, 833, 119)
    out.update(rand_r=geometry.r_vec, rand_theta=geometry.theta_vec, rand_doa_list=doa_list, rand_bf_mat=bf_mat, rand_cov_sel=covs[sel],
               rand_cov_idx=sel)
    radius, num_mic = 4.5e-2, 7
    geometry = LinearArray(spacing=2 * radius / num_mic, num_mic=num_mic, radius=radius)
    beamf = SNNBeamformer(geometry, 10e-3, [f / 2, 2 * f], [tau, tau], bipolar_spikes=False, fs=fs)
    np.random.seed(1)
    freq_inst = f * (1 + 0.01 * np.random.randn(len(time_temp)))
    sig_temp = np.sin(2 * np.pi * np.cumsum(freq_inst) / fs).astype(np.float32)
    doa_list = np.linspace(0, np.pi, 64 * num_mic + 1)
    bf_mat, covs = _design_with_covs(beamf, (time_temp, sig_temp.astype(np.float64)), doa_list)
    sel = np.arange(0, 449, 64)
    out.update(lin_r=geometry.r_vec, lin_theta=geometry.theta_vec, lin_doa_list=doa_list, lin_template_f32=sig_temp, lin_bf_mat=bf_mat,
               lin_cov_sel=covs[sel], lin_cov_idx=sel)
    out["freq_design"] = np.int64(f)
    save("design_other_geometries.npz", **out)


def gen_beamformer_sweep():
    """The accuracy sweep of the NON-spiking complex Beamformer (paper_plots/target_localization.py:400-440: the loop of
    target_snn_localization.py on Beamformer.apply_to_template, G = 8*7+1 = 57), 3 of the 11 SNRs x 40 trials, np.random.seed(0)."""
    radius, num_mic, fs = 4.5e-2, 7, 48_000
    freq_design = 2_000
    freq_range = [0.5 * freq_design, freq_design]
    beamf = Beamformer(geometry=CenterCircularArray(radius=radius, num_mic=num_mic), kernel_duration=10.0e-3, freq_range=freq_range, fs=fs)
    z = np.load(os.path.join(OUT, "beamformer_c128.npz"))
    bf_mat, doa_list = z["bf_mat"], z["doa_list"]
    snr_gain = (fs / 2) / (freq_range[1] - freq_range[0])
    snr_db_vec = np.array([-10.0, 5.0, 20.0])
    num_sim = 40
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * freq_design * time_test)
    np.random.seed(0)
    shape = (len(snr_db_vec), num_sim)
    doa, amax, err, pmax = np.zeros(shape), np.zeros(shape, dtype=np.int64), np.zeros(shape), np.zeros(shape)
    for i, snr_db in enumerate(snr_db_vec):
        snr_t = snr_db - 10 * np.log10(snr_gain)
        for sim in range(num_sim):
            d = np.random.rand(1)[0] * 2 * np.pi
            y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, d), snr_db=snr_t)
            power = np.mean(np.abs(y) ** 2, axis=0)
            k = int(np.argmax(power))
            doa[i, sim], amax[i, sim], pmax[i, sim] = d, k, power[k]
            err[i, sim] = np.arcsin(np.abs(np.sin(doa_list[k] - d)))
    save("beamformer_sweep_seed0.npz", seed=np.int64(0), snr_db_vec=snr_db_vec, num_sim=np.int64(num_sim), doa=doa, argmax=amax, err=err, pmax=pmax,
         mae_deg=np.mean(err, axis=1) * 180 / np.pi)


def gen_live_demo_frame():
    """The body of the live demo's loop (micloc/localization_demo_snn.py:125-193 with test_demo's configuration :196-222: one band
    [1600, 2400], 112 DoAs, 0.25 s packs, bipolar) restated from the reference's own components -- the module itself needs the sound
    card's recorder --: design per band, order-1 filterbank, apply_to_signal, power summed over the bands, arg-max.  The pack is a
    synthetic recording (2 kHz tone from one direction + noise) in the devkit's format: int32, 8 channels, the last one unused."""
    fs, num_mic = 48_000, 7
    geometry = CenterCircularArray(radius=4.5e-2, num_mic=num_mic)
    freq_bands = [[1600, 2400]]
    doa_list = np.linspace(-np.pi, np.pi, 16 * num_mic)
    rec = 0.25
    beamfs, bf_mats = [], []
    for fr in freq_bands:
        fm = np.mean(fr)
        tau = 1 / (2 * np.pi * fm)
        bfm = SNNBeamformer(geometry=geometry, kernel_duration=10e-3, freq_range=fr, tau_vec=[tau, tau], bipolar_spikes=True, fs=fs)
        t = np.arange(0, rec, step=1 / fs)
        bf_mats.append(quiet(bfm.design_from_template, template=(t, np.sin(2 * np.pi * fm * t)), doa_list=doa_list))
        beamfs.append(bfm)
    fb = ButterworthFilterbank(freq_bands=freq_bands, order=1, fs=fs)
    T = int(rec * fs)
    tt = np.arange(T) / fs
    rng = np.random.RandomState(21)
    d = geometry.delays(2.1, normalized=True)
    sig = np.sin(2 * np.pi * 2000 * (tt.reshape(-1, 1) - d.reshape(1, -1))) + 0.4 * rng.randn(T, num_mic)
    pack = np.zeros((T, 8), dtype=np.int32)
    pack[:, :7] = (np.rint(sig * 4096).astype(np.int32)) << 8
    data = np.asarray(pack[:, :-1], dtype=np.float64)
    assert np.sqrt(np.mean(data**2)) > 1e-4 * np.iinfo(np.int32).max
    time_vec = np.arange(0, T) / fs
    data_filt = fb.evolve(sig_in=data)
    power_grid = 0
    for filt, W, bfm in zip(data_filt, bf_mats, beamfs):
        y = bfm.apply_to_signal(bf_mat=W, sig_in_vec=(time_vec, filt))
        power_grid = power_grid + np.mean(np.abs(y) ** 2, axis=0)
    k = int(np.argmax(power_grid))
    save("live_demo_frame.npz", pack=pack, bf_mat0=bf_mats[0], doa_list=doa_list, freq_bands=np.asarray(freq_bands, dtype=np.float64), power_grid=power_grid,
         doa_index=np.int64(k), doa_deg=np.float64(doa_list[k] * 180 / np.pi), true_doa=np.float64(2.1))


def moving_target_synthetic(seed, T, G):
    """A beamformer-output stand-in the tests rebuild bit for bit from the seed (legacy MT19937 normals, products and comparisons
    only -- no libm): bursts over a quiet floor, a block of exact zeros (|y| == state ties), a dead column."""
    rng = np.random.RandomState(seed)
    y = rng.randn(T, G) * (0.2 + (np.arange(T)[:, None] % 1500 < 400) * 2.0)
    y[100:140] = 0.0
    y[:, 7] = 0.0
    return y


def gen_moving_target():
    """micloc/utils.py:36-81 (Envelope.evolve) and the moving-target experiment of paper_plots/target_snn_localization.py:585-622:
    (a) the reference class on seeded inputs (three window settings incl. a one-sample rise window): SHA-256 of the envelope array,
    sampled columns, the last row, the per-step arg-max; (b) a moving-DoA trial (chirp, doa(t) = pi/2 sin(pi t / 2 d), 0.5 s instead of
    the script's 5 s, quantised like a 16-bit converter) through the reference's apply_to_signal -> Envelope -> argmax."""
    import hashlib

    out = {}
    for k, (seed, T, G, rise, fall, fs) in enumerate([(11, 6000, 449, 10e-3, 100e-3, 48_000), (12, 3000, 130, 1e-3, 5e-3, 1_000), (13, 2000, 64, 2e-3, 2e-3, 8_000)]):
        y = moving_target_synthetic(seed, T, G)
        env = quiet(Envelope(rise_time=rise, fall_time=fall, fs=fs).evolve, y)
        cols = np.asarray([0, 7, G // 3, G - 1])
        out.update({f"syn{k}_params": np.asarray([seed, T, G, rise, fall, fs], dtype=np.float64), f"syn{k}_cols": cols, f"syn{k}_env_cols": env[:, cols],
                    f"syn{k}_env_last": env[-1], f"syn{k}_index": np.argmax(env, axis=1).astype(np.int16),
                    f"syn{k}_env_sha256": np.frombuffer(hashlib.sha256(np.ascontiguousarray(env).tobytes()).digest(), dtype=np.uint8)})
    # the other arrays the scripts hand to Envelope.evolve: a complex beamformer output (target_localization.py:597-600: np.abs = hypot, whose
    # last bit belongs to the host's libm: compared with a tolerance) and an integer spike raster (target_xylo_localization.py:757-768: exact)
    zc = moving_target_synthetic(31, 3000, 200) + 1j * moving_target_synthetic(32, 3000, 200)
    env = quiet(Envelope(rise_time=10e-3, fall_time=100e-3, fs=48_000).evolve, zc)
    top2 = np.partition(env, -2, axis=1)[:, -2:]
    out.update(cplx_params=np.asarray([31, 32, 3000, 200, 10e-3, 100e-3, 48_000]), cplx_env_cols=env[:, [0, 7, 66, 199]], cplx_env_last=env[-1],
               cplx_index=np.argmax(env, axis=1).astype(np.int16), cplx_margin=((top2[:, 1] - top2[:, 0]) / np.maximum(top2[:, 1], 1e-300)).astype(np.float32))
    rng = np.random.RandomState(41)
    spk = (rng.rand(4000, 449) < 0.05).astype(np.int64) * rng.randint(1, 4, size=(4000, 449))
    spk[:, 7] = 0
    env = quiet(Envelope(rise_time=40e-3, fall_time=200e-3, fs=48_000).evolve, spk)  # target_xylo_localization.py:759-762
    out.update(spk_params=np.asarray([41, 4000, 449, 40e-3, 200e-3, 48_000]), spk_env_cols=env[:, [0, 7, 100, 448]], spk_env_last=env[-1],
               spk_index=np.argmax(env, axis=1).astype(np.int16),
               spk_env_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(env).tobytes()).digest(), dtype=np.uint8))
    beamf, geometry, fs, fd, fr = cfg2_beamformer(True)
    bfz = np.load(os.path.join(OUT, "bf_mat_chirp449_bipolar.npz"))
    bf_mat, doa_list = bfz["bf_mat"], bfz["doa_list"]
    duration = 500e-3
    time_test = np.arange(0, duration, step=1 / fs)
    period = time_test[-1]
    freq_inst = fr[0] + (fr[1] - fr[0]) * (time_test % period) / period
    sig_test = np.sin(2 * np.pi * np.cumsum(freq_inst) * 1 / fs)
    doa_test = 0.5 * np.pi * np.sin(0.5 * np.pi / duration * time_test)  # :596-598
    cap = {}
    orig = beamf.apply_to_signal

    def spy(bf_mat, sig_in_vec):
        cap["time"] = np.array(sig_in_vec[0], copy=True)
        cap["sig"] = np.array(sig_in_vec[1], copy=True)
        return np.zeros((1, 1))

    beamf.apply_to_signal = spy
    try:
        np.random.seed(21)
        snr_db = 20.0 - 10 * np.log10(24.0)
        quiet(beamf.apply_to_template, bf_mat=bf_mat, template=(time_test, sig_test, doa_test), snr_db=snr_db)
    finally:
        beamf.apply_to_signal = orig
    q = np.clip(np.round(cap["sig"] * 4096.0), -32768, 32767).astype(np.int16)  # a 16-bit converter: steps of 2^-12
    sig_q = q.astype(np.float64) / 4096.0
    sig_bf = quiet(beamf.apply_to_signal, bf_mat=bf_mat, sig_in_vec=(cap["time"], sig_q))
    env = quiet(Envelope(rise_time=10e-3, fall_time=100e-3, fs=fs).evolve, sig_bf)
    index = np.argmax(env, axis=1)
    top2 = np.partition(env, -2, axis=1)[:, -2:]
    margin = (top2[:, 1] - top2[:, 0]) / np.maximum(top2[:, 1], 1e-300)
    out.update(trial_sig_q=q, trial_time=cap["time"], trial_doa=doa_test, trial_index=index.astype(np.int16), trial_margin=margin.astype(np.float32),
               trial_env_last=env[-1], trial_env_cols=env[:, [0, 224, 448]], trial_y_rows=sig_bf[[0, 1, 5000, 23998]], trial_rows=np.asarray([0, 1, 5000, 23998]),
               trial_doa_est_deg_rms=np.asarray(np.sqrt(np.mean((doa_list[index] - doa_test[: len(index)]) ** 2)) * 180 / np.pi))
    save("moving_target.npz", **out)


GENS = {
    "kat_init": gen_kat_init,
    "bf_mat_chirp": gen_bf_mat_chirp,
    "bf_mat_unipolar": gen_bf_mat_unipolar,
    "trials_cfg2": gen_trials_cfg2,
    "sweep_seed0": gen_sweep_seed0,
    "rzcc_edge": gen_rzcc_edge,
    "beamformer_c128": gen_beamformer_c128,
    "wide_case": gen_wide_case,
    "unipolar_trial": gen_unipolar_trial,
    "synth": gen_synth,
    "filterbank": gen_filterbank,
    "speech": gen_speech,
    "sweep_full": gen_sweep_full,
    "speech_sweep": gen_speech_sweep,
    "synth_xylo": gen_synth_xylo,
    "bf_mat_unipolar_hf": gen_bf_mat_unipolar_hf,
    "beamformer_c128_g449": gen_beamformer_c128_g449,
    "stress_case": gen_stress_case,
    "design_other_geometries": gen_design_other_geometries,
    "beamformer_sweep": gen_beamformer_sweep,
    "live_demo_frame": gen_live_demo_frame,
    "moving_target": gen_moving_target,
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GENS)
    for n in names:
        GENS[n]()
