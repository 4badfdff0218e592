"""Where the time of one SNNBeamformer.apply_to_template call goes (the unchanged-script path, B = 1)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray
from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bfz = np.load(os.path.join(ROOT, "tests", "golden", "bf_mat_chirp449_bipolar.npz"))
tau = 1.0 / (2 * np.pi * 2000.0)
beamf = SNNBeamformer(geometry=CenterCircularArray(radius=4.5e-2, num_mic=7), kernel_duration=10.0e-3, tau_vec=np.asarray([tau, tau]),
                      freq_range=[1000.0, 2000.0], fs=48_000, bipolar_spikes=True)
time_test = np.arange(0, 100e-3, step=1 / 48_000)
sig_test = np.sin(2 * np.pi * 2000.0 * time_test)
np.random.seed(0)
bf_mat = bfz["bf_mat"]
snr_db = 5.0 - 10 * np.log10(24.0)

def one():
    doa = np.random.rand(1)[0] * 2 * np.pi
    y = beamf.apply_to_template(bf_mat=bf_mat, template=(time_test, sig_test, doa), snr_db=snr_db)
    power = np.mean(np.abs(y) ** 2, axis=0)
    return y, int(np.argmax(power))

for _ in range(5):
    one()
t0 = time.perf_counter()
for _ in range(40):
    one()
print("ms per call %.3f" % ((time.perf_counter() - t0) / 40 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    one()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
