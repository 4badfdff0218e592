"""Time the packed Xylo LIF kernel alone on the sweep's shape (1100 trials x 48000 steps, 14 -> 28 channels, 360 neurons);
MICLOC_DEV_LIB selects a variant build."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd import _lib
if os.environ.get("MICLOC_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MICLOC_DEV_LIB"])
from haghighatshoarmuir2024_amd.xylo_snn_localization import XyloNetwork

rng = np.random.RandomState(0)
C, N, B, T = 14, 360, 1100, 48000
spec = dict(W_in=rng.randint(-127, 128, size=(2 * C, N)).astype(np.int8), w_rec=0, dash_syn=rng.randint(1, 4, size=N).astype(np.uint8),
            dash_mem=rng.randint(1, 4, size=N).astype(np.uint8), threshold=rng.randint(3000, 6000, size=N).astype(np.int16))
net = XyloNetwork(spec)
raster = (torch.rand(B, T, C, device="cuda") < 0.02).to(torch.int8) * (torch.randint(0, 2, (B, T, C), device="cuda", dtype=torch.int8) * 2 - 1)
for queued in (True, False, True, False):
    def run():
        return net.run(raster, ternary=True, queued=queued)[1]
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): c = run()
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("MICLOC_DEV_LIB", "default"), "ticket queue" if queued else "one workgroup per trial", "xylo LIF %.2f ms" % (e0.elapsed_time(e1) / 5), "spikes", int(c.sum()), flush=True)
