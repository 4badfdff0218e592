"""Time the LIF + beamforming + power stage alone (beamform_ws_kernel + power_argmax_kernel, HIP events) on the sweep's batch with a
REAL raster (the encoder's output for the bench's noisy 2 kHz tone); MICLOC_DEV_LIB selects a variant build; prints a checksum of the
power so that variants can be compared bit for bit.  BF_G: DoA count (default 360); BF_REPS."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd import _lib
if os.environ.get("MICLOC_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MICLOC_DEV_LIB"])
from haghighatshoarmuir2024_amd import runtime
from oracle import oracle as O

fs, M, B, T = 48000, 7, 1100, 4799
G = int(os.environ.get("BF_G", 360))
ker = O.stht_kernel(fs, 10e-3)
b, a = O.bandpass(fs, [1000.0, 2000.0])
tau = 1 / (2 * np.pi * 2000.0)
p = runtime.Plan(M, ker, b, a, O.robust_width(fs, 2000.0), True)
p.set_neuron_kernel(O.neuron_kernel(np.arange(T) / fs, [tau, tau]))
rng = np.random.RandomState(0)
W = rng.randn(2 * M, G)
p.set_bf_mat(W / np.linalg.norm(W, axis=0, keepdims=True))
t = np.arange(T) / fs
x = torch.from_numpy(np.sin(2 * np.pi * 2000 * t)[None, :, None] * np.ones((B, 1, M)) + 0.7 * rng.randn(B, T, M)).cuda()
out = p.snn_pipeline(x, want_spikes=True, want_power=True)
torch.cuda.synchronize()
mode = os.environ.get("BF_RASTER", "encoder")  # encoder | zero | dense | sparse_random: what the stage multiplies (timing against the DATA)
if mode == "zero":
    out["spikes"].zero_()
elif mode == "dense":
    out["spikes"].copy_(torch.randint(-1, 2, out["spikes"].shape, device="cuda", dtype=torch.int8))
elif mode == "sparse_random":
    r = torch.rand(out["spikes"].shape, device="cuda")
    out["spikes"].copy_(((r < 0.04).to(torch.int8) - (r > 0.96).to(torch.int8)))
torch.cuda.synchronize()
dens = float((out["spikes"] != 0).double().mean())
def run():
    p.snn_pipeline(x, stages=4, out=out)
for _ in range(3): run()
torch.cuda.synchronize()
N = int(os.environ.get("BF_REPS", 30))
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / N * 1e3)
print(os.path.basename(os.environ.get("MICLOC_DEV_LIB", "default")), "lif+beamform+power %.1f / %.1f / %.1f us" % tuple(ts), " G", G, " raster", mode, " spike density %.4f" % float((out["spikes"] != 0).double().mean()),
      " power checksum %.17g" % float(out["power"].sum()), " argmax sum", int(out["argmax"].sum()))
