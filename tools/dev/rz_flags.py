"""Debug: how many (stream, chunk) units of a chunked encoder launch are flagged for the fallback (config-4-like data)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd import runtime, synthesis
from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray
from haghighatshoarmuir2024_amd.xylo_snn_localization import Demo

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 110
order1 = (sys.argv[3] != "noisy") if len(sys.argv) > 3 else True
dev = torch.device("cuda", 0)
fs, M, G = 48000, 7, 360
geometry = CenterCircularArray(radius=4.5e-2, num_mic=M)
doa_list = np.linspace(-np.pi, np.pi, G)
demo = Demo(geometry=geometry, freq_bands=[[1000.0, 2000.0]], doa_list=doa_list, recording_duration=0.25, bipolar_spikes=True, fs=fs, device=dev)
t = np.arange(0, 1.0, 1 / fs)
s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % t[-1]) / t[-1]) / fs)
rng = np.random.RandomState(0)
doa = rng.rand(B) * 2 * np.pi
snr_vec = np.linspace(-10, 20, 11)
snr_db = snr_vec[(np.arange(B) * 11) // B] - 10 * np.log10(24)
x = synthesis.signal_from_template_batch(geometry, (t, s), doa, device=dev, device_delays=True)
synthesis.add_noise_(x, snr_db, seed=4321, first_trial=0)
T = x.shape[1]
enc = demo.beamfs[0].spk_encoder
bb, aa = demo.filterbank.ba_list[0]
plan = runtime.Plan(M, demo.beamfs[0].kernel, bb, aa, enc.robust_width, enc.bipolar, device=dev)
hq = plan.stht(x)
plan.set_encoder_chunk(-1)
_, ref = plan.bandpass_rzcc(hq, T, want_pre=False, want_spikes=True)
torch.cuda.synchronize()
plan.set_encoder_chunk(chunk)
P = plan.encoder_chunks(B, T)
_, spk = plan.bandpass_rzcc(hq, T, want_pre=False, want_spikes=True)
torch.cuda.synchronize()
ws, n = plan.workspace(B, T)
raw = ws[: 256 + 4 * P * B * 14].cpu().numpy().view(np.int32)
cnt = int(raw[0])
units = raw[64: 64 + cnt]
nl = B * 14
print(f"chunk {chunk}: P={P}, flagged {cnt} of {P * nl} units ({100.0 * cnt / (P * nl):.2f} %), equal to one pass: {bool(torch.equal(ref, spk))}")
print("per chunk:", np.bincount(units // nl, minlength=P))
print("per snr group:", np.bincount(((units % nl) // 14 * 11) // B, minlength=11))
print("per channel:", np.bincount((units % nl) % 14, minlength=14))
