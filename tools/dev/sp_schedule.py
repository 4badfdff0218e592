"""Experiment: a software-pipelined step -- the STHT of a stream's NEXT batch beside the encoder (and beamformer) of its current one.

The three streams of the headline run in step: three STHTs side by side, then three encoders, then three beamformers, and
the encoder phase is the only one that leaves the SIMDs idle (0.20 ms per step for 0.06 ms of issue).  Here every stream owns TWO plans
(workspaces) that alternate: graph `even` = [STHT of plan B on a forked stream] beside [encoder + beamformer of plan A]; graph `odd` the
same with the roles swapped.  One replay is still one STHT + one encoder + one beamformer launch -- a step -- but the STHT belongs to the
batch the NEXT replay encodes.

    python tools/dev/sp_schedule.py [steps] [streams] [order]      order: enc_first | stht_first
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from haghighatshoarmuir2024_amd import _lib  # noqa: E402

if os.environ.get("MICLOC_DEV_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MICLOC_DEV_LIB"])
from haghighatshoarmuir2024_amd import runtime  # noqa: E402
from oracle import oracle as O  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
order = sys.argv[3] if len(sys.argv) > 3 else "enc_first"
fs, M, B, T, G = 48000, 7, 1100, 4799, 360
ker = O.stht_kernel(fs, 10e-3)
b, a = O.bandpass(fs, [1000.0, 2000.0])
tau = 1 / (2 * np.pi * 2000.0)
nir = O.neuron_kernel(np.arange(T) / fs, [tau, tau])
rng = np.random.RandomState(0)
W = rng.randn(2 * M, G)
W /= np.linalg.norm(W, axis=0, keepdims=True)
t = np.arange(T) / fs
dev = torch.device("cuda", 0)


def new_plan():
    p = runtime.Plan(M, ker, b, a, O.robust_width(fs, 2000.0), True)
    p.set_neuron_kernel(nir)
    p.set_bf_mat(W)
    return p


xs = [torch.from_numpy(np.sin(2 * np.pi * 2000 * t)[None, :, None] * np.ones((B, 1, M)) + 0.7 * np.random.RandomState(10 + s).randn(B, T, M)).to(dev)
      for s in range(NS)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
sides = [torch.cuda.Stream(device=dev) for _ in range(NS)]

# ---- baseline: one graph per stream holding the whole pipeline --------------------------------------------------------------------
plans0 = [new_plan() for _ in range(NS)]
base = []
for p, x, s in zip(plans0, xs, streams):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = p.snn_pipeline(x, want_power=True)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        out = p.snn_pipeline(x, want_power=True, out=out)
    base.append((g, out))

# ---- software-pipelined: two plans per stream ------------------------------------------------------------------------------------
sp = []
for x, s, side in zip(xs, streams, sides):
    pa, pb = new_plan(), new_plan()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        oa = pa.snn_pipeline(x, want_power=True)  # (also leaves STHT(x) in plan A's workspace: primed)
        ob = pb.snn_pipeline(x, want_power=True)
    s.synchronize()

    def cap(cur, cur_out, nxt, nxt_out):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            if order == "stht_first":
                side.wait_stream(s)
                with torch.cuda.stream(side):
                    nxt.snn_pipeline(x, want_power=True, stages=1, out=nxt_out)
                cur.snn_pipeline(x, want_power=True, stages=2 | 4, out=cur_out)
            else:
                side.wait_stream(s)
                cur.snn_pipeline(x, want_power=True, stages=2, out=cur_out)
                with torch.cuda.stream(side):
                    nxt.snn_pipeline(x, want_power=True, stages=1, out=nxt_out)
                cur.snn_pipeline(x, want_power=True, stages=4, out=cur_out)
            s.wait_stream(side)
        return g

    sp.append(((cap(pa, oa, pb, ob), oa), (cap(pb, ob, pa, oa), ob), (pa, pb)))


def run_base(n):
    for k in range(n):
        i = k % NS
        with torch.cuda.stream(streams[i]):
            base[i][0].replay()


state = {"k": [0] * NS}


def run_sp(n):
    for k in range(n):
        i = k % NS
        par = state["k"][i] & 1
        state["k"][i] += 1
        with torch.cuda.stream(streams[i]):
            sp[i][par][0].replay()


def sync():
    for s in streams + sides:
        s.synchronize()
    torch.cuda.synchronize()


def timed(fn):
    fn(2 * NS)
    sync()
    t0 = time.perf_counter()
    fn(steps)
    sync()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(3):
    print("baseline (one graph per stream)      %.4f ms/step" % timed(run_base), "   software-pipelined STHT (%s)  %.4f ms/step" % (order, timed(run_sp)), flush=True)
# same results: the argmax of stream 0's batch from both schedules
sync()
a0 = base[0][1]["argmax"].cpu().numpy()
a1 = sp[0][0][1]["argmax"].cpu().numpy()
a2 = sp[0][1][1]["argmax"].cpu().numpy()
p0, p1 = base[0][1]["power"].cpu().numpy(), sp[0][0][1]["power"].cpu().numpy()
print("argmax equal:", bool(np.array_equal(a0, a1) and np.array_equal(a0, a2)), " power equal:", bool(np.array_equal(p0, p1)))
