"""Where config 5's design_from_template(svd="device") spends its time: the chain per DoA batch (synthesis, STHT, encoder, covariance) against
the one decomposition launch, for several batch sizes.  python tools/dev/design_cfg5_time.py [doa_batch ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd import runtime
from haghighatshoarmuir2024_amd.array_geometry import Random2DArray
from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer

fs, M, G = 96_000, 64, 1440
np.random.seed(1)
geo = Random2DArray(radius=0.2, num_mic=M)
tau = 1 / (2 * np.pi * 2000.0)
bf = SNNBeamformer(geo, 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
t = np.arange(0, 1.0, step=1 / fs)
chirp = np.sin(2 * np.pi * np.cumsum(1000.0 + 1000.0 * (t % t[-1]) / t[-1]) / fs)
doa = np.linspace(-np.pi, np.pi, G)
bf.design_from_template((t, chirp), doa[:8], svd="device")
for nb in [int(a) for a in sys.argv[1:]] or [48, 96, 144, 240]:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    W = bf.design_from_template((t, chirp), doa, svd="device", doa_batch=nb)
    torch.cuda.synchronize()
    print(f"doa_batch {nb}: design {time.perf_counter() - t0:.3f} s", flush=True)
cov = torch.randn(G, 2 * M, 2 * M, dtype=torch.float64, device="cuda")
cov = cov @ cov.transpose(1, 2) / (2 * M)
out = torch.empty((2 * M, G), dtype=torch.float64, device="cuda")
runtime.design_vectors(cov, True, out, 0); torch.cuda.synchronize()
t0 = time.perf_counter(); runtime.design_vectors(cov, True, out, 0); torch.cuda.synchronize()
print(f"decomposition launch alone (random covariances): {time.perf_counter() - t0:.3f} s")
