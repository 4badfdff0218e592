"""ctypes front-end of the CPU oracle + NumPy restatement of the reference's host-side glue.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; nothing under haghighatshoarmuir2024_amd/ does.

Reference citations (relative to /root/reference/):
  stht_kernel        micloc/snn_beamformer.py:48-53
  bandpass           micloc/snn_beamformer.py:68-72
  robust_width       micloc/snn_beamformer.py:75-76
  neuron_kernel      micloc/snn_beamformer.py:342-361
  delays             micloc/array_geometry.py:40-57
  synth_template     micloc/snn_beamformer.py:239-267
  add_noise          micloc/snn_beamformer.py:270-275
  doa_error          paper_plots/target_snn_localization.py:466
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmicloc_oracle.so")
# sanitizer run of the CPU suite (tests/test_oracle_sanitized.py): MICLOC_ORACLE_SO selects the AddressSanitizer + UBSan build
# (oracle/Makefile: _build/libmicloc_oracle_asan.so); the process must then be started with LD_PRELOAD=libasan.so
_SO_NAME = os.path.basename(os.environ.get("MICLOC_ORACLE_SO", "")) or "libmicloc_oracle.so"
if _SO_NAME not in ("libmicloc_oracle.so", "libmicloc_oracle_asan.so"):
    raise ImportError(f"MICLOC_ORACLE_SO must name libmicloc_oracle.so or libmicloc_oracle_asan.so, got {_SO_NAME}")
_SO = os.path.join(_HERE, "_build", _SO_NAME)
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_bp = ctypes.POINTER(ctypes.c_byte)
_ubp = ctypes.POINTER(ctypes.c_ubyte)


def build(force=False):
    """Compile oracle/_build/libmicloc_oracle.so with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "micloc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/" + _SO_NAME])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_stht.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp]
        L.oracle_iir_df2t.argtypes = [_dp, _dp, ctypes.c_int, _dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int]
        L.oracle_local_maxima.argtypes = [_dp, ctypes.c_int, _ip]
        L.oracle_local_maxima.restype = ctypes.c_int
        L.oracle_select_by_distance.argtypes = [_ip, _dp, ctypes.c_int, ctypes.c_int, _ubp]
        L.oracle_rzcc.argtypes = [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _bp]
        L.oracle_lif_fir.argtypes = [_bp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp]
        L.oracle_beamform.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp]
        L.oracle_power_argmax.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp]
        L.oracle_power_argmax.restype = ctypes.c_int
        L.oracle_snn_chain.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                       _dp, _bp, _dp, _dp, _dp]
        L.oracle_snn_chain.restype = ctypes.c_int
        L.oracle_beamformer_chain.argtypes = [_dp, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int,
                                              _dp, _dp, ctypes.c_int, _dp, _dp, _dp, _dp]
        L.oracle_beamformer_chain.restype = ctypes.c_int
        L.oracle_snn_chain_batch.argtypes = [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp,
                                             ctypes.c_int, _dp, _ip]
        L.oracle_xylo_lif.argtypes = [_ubp, ctypes.c_int, ctypes.c_int, _bp, ctypes.c_int, ctypes.c_int, _ubp, _ubp,
                                      ctypes.POINTER(ctypes.c_short), ctypes.c_int, _ubp, _ip]
        _up = ctypes.POINTER(ctypes.c_uint)
        L.oracle_philox4x32_10.argtypes = [_up, _up, _up]
        L.oracle_uniform.argtypes = [_dp, ctypes.c_longlong, ctypes.c_ulonglong, ctypes.c_uint, ctypes.c_uint, ctypes.c_double, ctypes.c_double]
        L.oracle_normals.argtypes = [_dp, ctypes.c_longlong, ctypes.c_ulonglong, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _pad_ba(b, a):
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    n = max(len(b), len(a))
    bb = np.zeros(n)
    aa = np.zeros(n)
    bb[: len(b)] = b / a[0]
    aa[: len(a)] = a / a[0]
    return bb, aa, n


# ------------------------------------------------------------------------------------------------
# C oracle wrappers (single trial, reference layout T x M / T x C row-major)
# ------------------------------------------------------------------------------------------------
def stht(x, kernel):
    x, xp = _d(x)
    k, kp = _d(kernel)
    T, M = x.shape
    re = np.empty((T, M))
    im = np.empty((T, M))
    lib().oracle_stht(xp, T, M, kp, len(k), re.ctypes.data_as(_dp), im.ctypes.data_as(_dp))
    return re, im


def iir(b, a, x):
    """lfilter(b, a, x, axis=0) for real x of shape (T, C)."""
    x, _ = _d(x)
    T, C = x.shape
    bb, aa, n = _pad_ba(b, a)
    y = np.empty_like(x)
    for c in range(C):
        lib().oracle_iir_df2t(bb.ctypes.data_as(_dp), aa.ctypes.data_as(_dp), n,
                              x[:, c:].ctypes.data_as(_dp), T, C, y[:, c:].ctypes.data_as(_dp), C)
    return y


def rzcc(r, robust_width, bipolar):
    r, rp = _d(r)
    if r.ndim == 1:
        r = r.reshape(-1, 1)
    T, C = r.shape
    s = np.zeros((T, C), dtype=np.int8)
    if T > 0 and C > 0:
        lib().oracle_rzcc(r.ctypes.data_as(_dp), T, C, int(robust_width), int(bool(bipolar)), s.ctypes.data_as(_bp))
    return s


def lif_fir(spikes, nir):
    s = np.ascontiguousarray(spikes, dtype=np.int8)
    T, C = s.shape
    k, kp = _d(nir)
    v = np.empty((T, C))
    lib().oracle_lif_fir(s.ctypes.data_as(_bp), T, C, kp, len(k), v.ctypes.data_as(_dp))
    return v


def beamform(v, W):
    v, vp = _d(v)
    W, wp = _d(W)
    T, C = v.shape
    G = W.shape[1]
    y = np.empty((T, G))
    lib().oracle_beamform(vp, T, C, wp, G, y.ctypes.data_as(_dp))
    return y


def snn_chain(x, kernel, b, a, robust_width, bipolar, nir, W, want=("pre_enc", "spikes", "vmem", "y", "power")):
    """SNNBeamformer.apply_to_signal + power/argmax for one trial.  Returns a dict."""
    x, xp = _d(x)
    T, M = x.shape
    C = 2 * M
    k, kp = _d(kernel)
    bb, aa, n = _pad_ba(b, a)
    nir, nirp = _d(nir)
    W, wp = _d(W)
    assert W.shape[0] == C
    G = W.shape[1]
    out = {}
    pre = np.empty((T, C)) if "pre_enc" in want else None
    spk = np.zeros((T, C), dtype=np.int8) if "spikes" in want else None
    vm = np.empty((T, C)) if "vmem" in want else None
    y = np.empty((T, G)) if "y" in want else None
    pw = np.empty(G)
    null_d = ctypes.cast(None, _dp)
    am = lib().oracle_snn_chain(
        xp, T, M, kp, len(k), bb.ctypes.data_as(_dp), aa.ctypes.data_as(_dp), n, int(robust_width), int(bool(bipolar)),
        nirp, len(nir), wp, G,
        pre.ctypes.data_as(_dp) if pre is not None else null_d,
        spk.ctypes.data_as(_bp) if spk is not None else ctypes.cast(None, _bp),
        vm.ctypes.data_as(_dp) if vm is not None else null_d,
        y.ctypes.data_as(_dp) if y is not None else null_d,
        pw.ctypes.data_as(_dp),
    )
    out.update(pre_enc=pre, spikes=spk, vmem=vm, y=y, power=pw, argmax=int(am))
    return out


def snn_chain_batch(x, kernel, b, a, robust_width, bipolar, nir, W):
    """B trials -> (power [B,G], argmax [B]); the loop bench.py times as the CPU baseline."""
    x, xp = _d(x)
    B, T, M = x.shape
    k, kp = _d(kernel)
    bb, aa, n = _pad_ba(b, a)
    nir, nirp = _d(nir)
    W, wp = _d(W)
    G = W.shape[1]
    pw = np.empty((B, G))
    am = np.empty(B, dtype=np.int32)
    lib().oracle_snn_chain_batch(xp, B, T, M, kp, len(k), bb.ctypes.data_as(_dp), aa.ctypes.data_as(_dp), n,
                                 int(robust_width), int(bool(bipolar)), nirp, len(nir), wp, G,
                                 pw.ctypes.data_as(_dp), am.ctypes.data_as(_ip))
    return pw, am


def snn_chain_numpy(x, kernel, b, a, robust_width, bipolar, nir, W):
    """The same chain as snn_chain, op for op as the reference runs it on the CPU (NumPy / SciPy calls, BLAS threads at
    their default): snn_beamformer.py:325-368 + spike_encoder.py:115-137 + target_snn_localization.py:462-464.
    bench.py times this as the `numpy_ops` leg of cpu_baseline; tests check it against snn_chain."""
    from scipy.signal import find_peaks, lfilter

    x = np.asarray(x, dtype=np.float64)
    L = len(kernel)
    sig_h = np.roll(x, L // 2, axis=0) + 1j * lfilter(kernel, [1], x, axis=0)
    sig_h = lfilter(b, a, sig_h, axis=0)
    r = np.hstack([np.real(sig_h), np.imag(sig_h)])
    spikes = np.zeros_like(r)
    for ch in range(r.shape[1]):
        peaks, _ = find_peaks(np.cumsum(r[:, ch]), distance=robust_width)
        spikes[peaks, ch] = 1
        if bipolar:
            valleys, _ = find_peaks(-np.cumsum(r[:, ch]), distance=robust_width)
            spikes[valleys, ch] = -1
    vmem = lfilter(nir, [1], spikes, axis=0)
    y = vmem @ W
    power = np.mean(np.abs(y) ** 2, axis=0)
    return dict(spikes=spikes.astype(np.int8), power=power, argmax=int(np.argmax(power)))


def envelope(y, win_fall, win_rise):
    """Envelope.evolve (ref:micloc/utils.py:36-81) restated: state_0 = |y_0|; per step `rise = |y_t| >= state`,
    `state = (1 - 1/w) * state + 1/w * |y_t| * rise` with w = win_rise if rise else win_fall, in NumPy's order of operations;
    out[t] = state after sample t.  y [T, G] -> [T, G].  Checker of micloc_envelope_track_f64 (tests only)."""
    mag = np.abs(np.asarray(y)).astype(np.float64)  # (complex: the modulus; integers: exact -- the reference's mixed list ends as float64)
    wl = np.asarray([int(win_fall), int(win_rise)])
    state = mag[0].copy()
    out = np.empty_like(mag)
    out[0] = state
    for t in range(1, len(mag)):
        rise = (mag[t] >= state).astype(int)
        w = wl[rise]
        state = (1 - 1 / w) * state + 1 / w * mag[t] * rise
        out[t] = state
    return out


def snn_chain_batch_parallel(x, kernel, b, a, robust_width, bipolar, nir, W, threads):
    """snn_chain_batch with the trials split over `threads` host threads (the C call releases the GIL): the
    trial-parallel leg of bench.py's cpu_baseline (target_snn_localization.py:447-467 has no cross-trial state)."""
    from concurrent.futures import ThreadPoolExecutor

    x = np.ascontiguousarray(x, dtype=np.float64)
    B = x.shape[0]
    threads = max(1, min(int(threads), B))
    bounds = [(i * B // threads, (i + 1) * B // threads) for i in range(threads)]
    lib()
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda lh: snn_chain_batch(x[lh[0] : lh[1]], kernel, b, a, robust_width, bipolar, nir, W), bounds))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def beamformer_chain(x, kernel, b, a, W, want_y=True):
    """Beamformer.apply_to_signal (complex W of shape M x G) + power/argmax."""
    x, xp = _d(x)
    T, M = x.shape
    k, kp = _d(kernel)
    bb, aa, n = _pad_ba(b, a)
    Wre, wrp = _d(np.real(W))
    Wim, wip = _d(np.imag(W))
    G = Wre.shape[1]
    pre = np.empty((T, 2 * M))
    yre = np.empty((T, G)) if want_y else None
    yim = np.empty((T, G)) if want_y else None
    pw = np.empty(G)
    null_d = ctypes.cast(None, _dp)
    am = lib().oracle_beamformer_chain(xp, T, M, kp, len(k), bb.ctypes.data_as(_dp), aa.ctypes.data_as(_dp), n, wrp, wip, G,
                                       pre.ctypes.data_as(_dp),
                                       yre.ctypes.data_as(_dp) if want_y else null_d,
                                       yim.ctypes.data_as(_dp) if want_y else null_d,
                                       pw.ctypes.data_as(_dp))
    return dict(pre=pre, y=(yre + 1j * yim) if want_y else None, power=pw, argmax=int(am))


# ------------------------------------------------------------------------------------------------
# NumPy restatement of the host-side parameter builders and of the synthesis (small, O(T) work)
# ------------------------------------------------------------------------------------------------
def stht_kernel(fs, kernel_duration):
    """fftshift(imag(hilbert(delta_L))) in closed form: for even L, imag(hilbert(delta))[n] =
    (2/L) cot(pi n / L) for odd n and 0 for even n; scipy builds it with an FFT, so the
    closed form agrees to ~1e-16, not bit for bit -- the product code calls scipy like the reference."""
    from scipy.signal import hilbert

    L = int(fs * kernel_duration)
    imp = np.zeros(L)
    imp[0] = 1
    return np.fft.fftshift(np.imag(hilbert(imp)))


def bandpass(fs, freq_range, order=2):
    from scipy.signal import butter

    return butter(order, freq_range, btype="bandpass", analog=False, output="ba", fs=fs)


def robust_width(fs, f_high):
    return int(fs / f_high) // 2


def neuron_kernel(time_vec, tau_vec):
    tau_syn, tau_mem = tau_vec[0], tau_vec[1]
    if tau_syn != tau_mem:
        raise ValueError("only tau_syn == tau_mem is supported (the reference's other branch trips its own assert)")
    t = np.asarray(time_vec) - time_vec[0]
    h = (t / tau_syn) * np.exp(-t / tau_syn)
    h = h / np.sum(h)
    n = int(np.sum(np.cumsum(h) < 0.999))
    return h[:n]


def delays(r_vec, theta_vec, theta, normalized=True, speed=340):
    d = -np.asarray(r_vec) * np.cos(np.asarray(theta_vec) - theta) / speed
    if normalized:
        d = d - np.min(d)
    return d


def center_circular(radius, num_mic):
    r_vec = np.array([*list(radius * np.ones(num_mic - 1)), 0.0])
    theta_vec = np.array([*list(np.linspace(0, 2 * np.pi, num_mic - 1)), 0.0])
    return r_vec, theta_vec


def synth_template(r_vec, theta_vec, time_temp, sig_temp, doa_temp, fs, speed=340):
    """Noise-free array signal of SNNBeamformer.apply_to_template; returns (time[T], sig[T,M])."""
    time_temp = np.asarray(time_temp, dtype=np.float64)
    sig_temp = np.asarray(sig_temp, dtype=np.float64)
    if np.isscalar(doa_temp) or np.ndim(doa_temp) == 0:
        doa_temp = float(doa_temp) * np.ones_like(sig_temp)
    time_in = np.arange(time_temp.min(), time_temp.max(), step=1 / fs)
    sig_in = np.interp(time_in, time_temp, sig_temp)
    doa_in = np.interp(time_in, time_temp, doa_temp)
    # delays[m, t] = -r_m cos(theta_m - doa_t) / c   (vectorised form of the reference's list-comp)
    d = -np.asarray(r_vec).reshape(-1, 1) * np.cos(np.asarray(theta_vec).reshape(-1, 1) - doa_in.reshape(1, -1)) / speed
    d = d - d.min()
    time_delayed = time_in.reshape(1, -1) - d
    time_delayed[time_delayed < time_in.min()] = time_in.min()
    sig = np.interp(time_delayed.ravel(), time_in, sig_in).reshape(time_delayed.shape).T
    return time_in, np.ascontiguousarray(sig)


def add_noise(sig, snr_db, randn=None):
    """In-place AWGN exactly like snn_beamformer.py:270-275 (global legacy NumPy stream by default)."""
    randn = randn or np.random.randn
    snr = 10 ** (snr_db / 10)
    noise = np.sqrt(np.mean(sig**2)) / np.sqrt(snr) * randn(*sig.shape)
    sig += noise
    return sig


def doa_error(doa_est, doa_true):
    return np.arcsin(np.abs(np.sin(doa_est - doa_true)))


def xylo_lif(spikes_in, W_in, w_rec, dash_syn, dash_mem, thr, max_spikes=31):
    """Integer LIF of the Xylo hidden layer (parity unpinned, see micloc_oracle.c). spikes_in [T, Cin] uint8,
    W_in [Cin, N] int8 -> (spikes_out [T, N] uint8, rate [N] int32)."""
    s = np.ascontiguousarray(spikes_in, dtype=np.uint8)
    W = np.ascontiguousarray(W_in, dtype=np.int8)
    T, Cin = s.shape
    N = W.shape[1]
    ds = np.ascontiguousarray(np.broadcast_to(dash_syn, (N,)), dtype=np.uint8)
    dm = np.ascontiguousarray(np.broadcast_to(dash_mem, (N,)), dtype=np.uint8)
    th = np.ascontiguousarray(np.broadcast_to(thr, (N,)), dtype=np.int16)
    out = np.zeros((T, N), dtype=np.uint8)
    rate = np.zeros(N, dtype=np.int32)
    lib().oracle_xylo_lif(s.ctypes.data_as(_ubp), T, Cin, W.ctypes.data_as(_bp), N, int(w_rec), ds.ctypes.data_as(_ubp),
                          dm.ctypes.data_as(_ubp), th.ctypes.data_as(ctypes.POINTER(ctypes.c_short)), int(max_spikes),
                          out.ctypes.data_as(_ubp), rate.ctypes.data_as(_ip))
    return out, rate


# ------------------------------------------------------------------------------------------------
# counter-based random numbers of the throughput-mode sweep (csrc/rng.hip) and the signal generators of the
# other scripts (micloc/xylo_snn_localization.py:44-71, paper_plots/multiple_targets_snn.py:87-160)
# ------------------------------------------------------------------------------------------------
def philox4x32_10(ctr, key):
    c = (ctypes.c_uint * 4)(*[int(v) & 0xFFFFFFFF for v in ctr])
    k = (ctypes.c_uint * 2)(*[int(v) & 0xFFFFFFFF for v in key])
    o = (ctypes.c_uint * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def uniform(n, seed, substream=0, lo=0.0, hi=1.0, epoch=0):
    out = np.empty(int(n))
    lib().oracle_uniform(out.ctypes.data_as(_dp), int(n), int(seed), int(substream), int(epoch), float(lo), float(hi))
    return out


def normals(n, seed, substream, trial, epoch=0):
    z = np.empty(int(n))
    lib().oracle_normals(z.ctypes.data_as(_dp), int(n), int(seed), int(substream), int(epoch), int(trial))
    return z


def awgn(x, snr_db, seed, substream=0, first_trial=0, epoch=0):
    """x [B, T, M] + sigma_b * N(0, 1), sigma_b = sqrt(mean(x_b^2)) / sqrt(10^(snr_db_b / 10)) (snn_beamformer.py:270-275)
    with the device's Philox stream.  Returns (noisy, sigma)."""
    x = np.asarray(x, dtype=np.float64)
    B = x.shape[0]
    snr_db = np.broadcast_to(np.asarray(snr_db, dtype=np.float64), (B,))
    sigma = np.sqrt(np.mean(x.reshape(B, -1) ** 2, axis=1)) / np.sqrt(10 ** (snr_db / 10))
    out = x.copy()
    for b in range(B):
        out[b] += sigma[b] * normals(x[b].size, seed, substream, first_trial + b, epoch).reshape(x[b].shape)
    return out, sigma


def signal_from_template(r_vec, theta_vec, time_temp, sig_temp, doa, speed=340):
    """micloc/xylo_snn_localization.py:44-71: interp(t + delays(doa_t), t, s); no min-shift, no clamp."""
    time_temp = np.asarray(time_temp, dtype=np.float64)
    doa = np.broadcast_to(np.asarray(doa, dtype=np.float64), time_temp.shape)
    d = -np.asarray(r_vec)[None, :] * np.cos(np.asarray(theta_vec)[None, :] - doa.reshape(-1, 1)) / speed  # [T, M], array_geometry.py:52
    td = time_temp.reshape(-1, 1) + d
    return np.interp(td.ravel(), time_temp, sig_temp).reshape(td.shape)


def signal_multiple_targets(r_vec, theta_vec, time_temp, sig_temp, doa_ts, power_ts, speed=340):
    """paper_plots/multiple_targets_snn.py:87-160: sum over targets of power_k[t] * interp(t + delays(doa_k[t]), t, s)."""
    doa_ts = np.asarray(doa_ts, dtype=np.float64)
    power_ts = np.asarray(power_ts, dtype=np.float64)
    sig_in = 0
    for k in range(doa_ts.shape[1]):
        sig_in = sig_in + power_ts[:, k].reshape(-1, 1) * signal_from_template(r_vec, theta_vec, time_temp, sig_temp, doa_ts[:, k], speed)
    return sig_in
