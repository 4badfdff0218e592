"""The moving-target read-out of ref:paper_plots/target_snn_localization.py:585-622 at the script's own shape (5 s at 48 kHz = 240 000
frames, 449 DoAs): Envelope.track on the device (micloc_envelope_track_f64: envelope_kernel + rows_argmax_kernel, T int32 indices back)
against what an unchanged script does (the T x G array to the host, the reference's Python loop over T, np.argmax).  HIP events for the
kernels, wall clock for the host route (the loop is timed on the first 24 000 frames and scaled)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd.utils import Envelope

T, G = int(os.environ.get("TRACK_T", 240_000)), int(os.environ.get("TRACK_G", 449))
env = Envelope(rise_time=10e-3, fall_time=100e-3, fs=48_000)
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn((T, G), dtype=torch.float64, device="cuda", generator=g) * (0.2 + 2.0 * (torch.arange(T, device="cuda")[:, None] % 1500 < 400))
idx = env.track(y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 5
e0.record()
for _ in range(N):
    idx = env.track(y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
t0 = time.perf_counter(); idx_h = idx.cpu().numpy(); t_idx = time.perf_counter() - t0
t0 = time.perf_counter(); yh = y.cpu().numpy(); t_d2h = time.perf_counter() - t0
n = min(T, 24_000)
t0 = time.perf_counter(); eh = env.evolve(yh[:n]); ih = np.argmax(eh, axis=1); t_host = (time.perf_counter() - t0) * T / n
assert np.array_equal(ih, idx_h[:n])
print(f"T={T} G={G}: device Envelope.track {ms:.2f} ms ({T * G * 8 * 3 / ms / 1e6:.0f} GB/s over y read + envelope written + read), "
      f"indices D2H {t_idx * 1e3:.2f} ms ({idx_h.nbytes / 1e6:.2f} MB); host route: y D2H {t_d2h * 1e3:.0f} ms ({yh.nbytes / 1e6:.0f} MB) + "
      f"Envelope.evolve + argmax on the host ~{t_host:.1f} s (timed on {n} frames); first {n} indices equal")
