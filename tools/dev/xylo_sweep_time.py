"""xylo_target_sweep(mode="throughput") on one GPU, wall clock of the whole call (BASELINE config 4: 11 SNRs x 100 trials of 1 s):
one batch of 1100 (the default) against batches of 50 (the default until round 4).  usage: python tools/dev/xylo_sweep_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray  # noqa: E402
from haghighatshoarmuir2024_amd.sweep import xylo_target_sweep  # noqa: E402
from haghighatshoarmuir2024_amd.xylo_snn_localization import Demo  # noqa: E402

fs, M, G = 48_000, 7, 360
demo = Demo(geometry=CenterCircularArray(radius=4.5e-2, num_mic=M), freq_bands=[[1000.0, 2000.0]], doa_list=np.linspace(-np.pi, np.pi, G),
            recording_duration=0.25, bipolar_spikes=True, fs=fs)
ref = None
for name, kw in (("warm-up", dict(num_sim=10)), ("one batch of 1100", {}), ("batches of 50", dict(batch_trials=50)), ("one batch of 1100", {})):
    kw = dict(dict(num_sim=100), **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = xylo_target_sweep(demo, seed=4, mode="throughput", **kw)
    dt = time.perf_counter() - t0
    n = res["index"].size
    print(f"{name:18s}: {n} trials in {dt * 1e3:8.1f} ms = {n * 48000 / dt / 1e9:.2f} e9 frames/s", flush=True)
    if name != "warm-up":
        if ref is None:
            ref = res
        else:
            assert np.array_equal(res["index"], ref["index"])
print("identical peak indices")
