"""Cluster sizes of the RZCC candidate trains on the config-4 workload (order-1 band-pass, w = 12): local maxima / minima of the running
sum by scipy.signal.find_peaks on the device's band-passed streams, split where the gap reaches w.  Result (8 trials, 112 streams): 82.6 %
single candidates, 11.1 % pairs, 3.8 % triples, 2.4 % four to eight, 0.03 % longer -- what the resolver wave of csrc/rzcc.hip takes is the 2.4 %.
usage: python tools/dev/cluster_hist.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scipy.signal import find_peaks
from haghighatshoarmuir2024_amd import runtime, synthesis
from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray
from haghighatshoarmuir2024_amd.xylo_snn_localization import Demo
fs, M, G, B = 48000, 7, 360, 8
geometry = CenterCircularArray(radius=4.5e-2, num_mic=M)
demo = Demo(geometry=geometry, freq_bands=[[1000.0, 2000.0]], doa_list=np.linspace(-np.pi, np.pi, G), recording_duration=0.25, bipolar_spikes=True, fs=fs)
t = np.arange(0, 1.0, 1 / fs)
s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % t[-1]) / t[-1]) / fs)
rng = np.random.RandomState(0)
doa = rng.rand(B) * 2 * np.pi
snr_db = np.linspace(-10, 20, B) - 10 * np.log10(24)
x = synthesis.signal_from_template_batch(geometry, (t, s), doa, device_delays=True)
synthesis.add_noise_(x, snr_db, seed=4321, first_trial=0)
enc = demo.beamfs[0].spk_encoder
bb, aa = demo.filterbank.ba_list[0]
plan = runtime.Plan(M, demo.beamfs[0].kernel, bb, aa, enc.robust_width, enc.bipolar)
T = x.shape[1]
hq = plan.stht(x)
out = plan.bandpass_rzcc(hq, T, want_pre=True, want_spikes=False)
pre = out["pre"].cpu().numpy() if isinstance(out, dict) else out[0].cpu().numpy()
w = enc.robust_width
print("robust width", w, "pre", pre.shape)
hist = np.zeros(64, dtype=np.int64)
for b in range(B):
    for c in range(pre.shape[1]):
        cs = np.cumsum(pre[b, c, :T])
        for sign in (1, -1):
            p, _ = find_peaks(sign * cs)
            if len(p) == 0: continue
            gaps = np.diff(p)
            sizes = np.diff(np.concatenate([[0], np.nonzero(gaps >= w)[0] + 1, [len(p)]]))
            for k in sizes: hist[min(k, 63)] += 1
tot = hist.sum()
print("clusters", tot, "candidates", int((hist * np.arange(64)).sum()))
cum = 0
for k in range(1, 40):
    if hist[k]:
        print(k, hist[k], "%.3f" % (hist[k] / tot), "cand share %.3f" % (hist[k] * k / (hist * np.arange(64)).sum()))
