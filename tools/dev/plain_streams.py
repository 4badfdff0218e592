"""The speech workload through plain launches on three streams (no HIP graphs), to compare with bench.py --config speech
(graph replay) on the same box.  usage: python tools/dev/plain_streams.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd import runtime
from oracle import oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dev = torch.device("cuda", 0)
fs, M, B, T, G = 48000, 7, 125, 332157, 360
ker = O.stht_kernel(fs, 10e-3)
b, a = O.bandpass(fs, [1000.0, 2000.0])
tau = 1 / (2 * np.pi * 2000.0)
nir = O.neuron_kernel(np.arange(2000) / fs, [tau, tau])
W = np.random.RandomState(0).randn(2 * M, G)
x = torch.randn(B, T, M, dtype=torch.float64, device=dev)
for nstreams in (3, 3):
    plans = []
    for _ in range(nstreams):
        p = runtime.Plan(M, ker, b, a, O.robust_width(fs, 2000.0), True)
        p.set_neuron_kernel(nir)
        p.set_bf_mat(W)
        plans.append(p)
    mains = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    outs = [p.snn_pipeline(x, want_power=True) for p in plans]
    torch.cuda.synchronize()
    def step(i):
        with torch.cuda.stream(mains[i % nstreams]):
            plans[i % nstreams].snn_pipeline(x, want_power=True, out=outs[i % nstreams])
    for i in range(3): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    print("plain launches, %d streams: ms/step %.2f" % (nstreams, (time.perf_counter() - t0) / steps * 1e3))
