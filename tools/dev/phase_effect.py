"""How much does the arbitrary unit phase of a bipolar singular vector move power / arg-max?  (svd='device' vs the reference's fixture)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray
from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer
from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep
g = "tests/golden/"
z = np.load(g + "sweep_full_seed0.npz"); bfz = np.load(g + "bf_mat_chirp449_bipolar.npz")
tau = 1 / (2 * np.pi * 2000.0)
bf = SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=48_000)
fs = 48_000
t = np.arange(0, 1.0, 1 / fs); period = t[-1]
s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % period) / period) / fs)
Wd = bf.design_from_template((t, s), bfz["doa_list"], svd="device")
res_h = noisy_target_sweep(bf, bfz["bf_mat"], bfz["doa_list"], num_sim=100, seed=0, mode="parity")
res_d = noisy_target_sweep(bf, Wd, bfz["doa_list"], num_sim=100, seed=0, mode="parity")
print("argmax equal", np.mean(res_h["argmax"] == res_d["argmax"]), "per snr", np.mean(res_h["argmax"] == res_d["argmax"], axis=1))
print("mae host", np.round(res_h["mae_deg"], 3)); print("mae dev ", np.round(res_d["mae_deg"], 3))
print("max |dmae|", np.abs(res_h["mae_deg"] - res_d["mae_deg"]).max(), "pmax rel diff max", np.abs(res_d["pmax"] / res_h["pmax"] - 1).max())
