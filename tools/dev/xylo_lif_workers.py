"""The Xylo LIF alone on the bench's batch: one workgroup per trial against the ticket queue with 2..8 persistent workgroups per CU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd.xylo_snn_localization import XyloNetwork
rng = np.random.RandomState(0)
C, N, B, T = 14, 360, 1100, 48000
spec = dict(W_in=rng.randint(-127, 128, size=(2 * C, N)).astype(np.int8), w_rec=0, dash_syn=rng.randint(1, 4, size=N).astype(np.uint8),
            dash_mem=rng.randint(1, 4, size=N).astype(np.uint8), threshold=rng.randint(3000, 6000, size=N).astype(np.int16))
net = XyloNetwork(spec)
raster = (torch.rand(B, T, C, device="cuda") < 0.07).to(torch.int8) * (torch.randint(0, 2, (B, T, C), device="cuda", dtype=torch.int8) * 2 - 1)
def timed(**kw):
    for _ in range(2): net.run(raster, ternary=True, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): c = net.run(raster, ternary=True, **kw)[1]
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4, int(c.sum())
print("static", "%.2f ms" % timed(queued=False)[0], flush=True)
for w in (2, 3, 4, 5, 6, 8):
    print("queue, %d workgroups per CU" % w, "%.2f ms" % timed(queued=True, workers_per_cu=w)[0], flush=True)
print("static", "%.2f ms" % timed(queued=False)[0], flush=True)
