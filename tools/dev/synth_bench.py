"""Time micloc_synth_awgn_f64 alone on the headline shapes (B = 1100, T = 4800, M = 7); MICLOC_DEV_LIB selects a variant build."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd import _lib
if os.environ.get("MICLOC_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MICLOC_DEV_LIB"])
from haghighatshoarmuir2024_amd import runtime, synthesis
from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray

dev = torch.device("cuda", 0)
fs, M, B = 48000, 7, 1100
geometry = CenterCircularArray(radius=4.5e-2, num_mic=M)
t = np.arange(0, 0.1, 1 / fs)
s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * t / t[-1]) / fs)
tpl = runtime.Template(t, s, fs, device=dev)
geo = runtime.Geometry(geometry, device=dev)
doa = torch.rand(B, 1, dtype=torch.float64, device=dev) * 2 * np.pi
shift = runtime.delay_min(doa, geo)
snr = torch.linspace(-10, 20, B, dtype=torch.float64, device=dev)
x = torch.empty((B, len(t), M), dtype=torch.float64, device=dev)
ws = runtime.awgn_workspace(B, len(t), M, dev)
def run():
    runtime.synth_awgn(tpl, "apply_to_template", snr, seed=7, ws=ws, doa=doa, geometry=geo, shift=shift, out=x)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 30
e0.record()
for _ in range(N): run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("MICLOC_DEV_LIB", "default"), "synth_awgn %.1f us per call" % (e0.elapsed_time(e1) / N * 1e3), " checksum %.6f" % float(x.sum()))
