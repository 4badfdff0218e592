"""Speech workload (config 3), three schedules of the same steps on three HIP streams:
  graph    one captured hipGraph per stream, replayed round-robin (what bench.py did until round 4)
  eager    the same calls launched eagerly, round-robin
  lane     eager, with the band-pass / RZCC stage of every step on ONE shared stream (fork / join by events): the serial scans of
           consecutive steps run one after the other instead of side by side, the other stages overlap them
usage: python tools/dev/speech_lane.py [steps] [streams]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from haghighatshoarmuir2024_amd import runtime  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 3
args = bench.parse(["--config", "speech"])
args.world = 1
dev = torch.device("cuda", 0)
wl = bench.build_workload(args, 0, dev)
plans = [wl["plan"]]
inputs = [(wl["x"], wl["doa"])]
for i in range(1, nstreams):
    p = wl["beamf"].new_plan()
    p.set_neuron_kernel(wl["nir"])
    p.set_bf_mat(wl["bf_mat"])
    plans.append(p)
    _, x_i, doa_i = wl["make_batch"](i)
    inputs.append((x_i, torch.from_numpy(doa_i).to(dev)))
pipe = runtime.StreamPipeline(plans)
S, doa_list = wl["snr_groups"], wl["doa_list"]
outs = [None] * nstreams


def body(i, stages=7):
    outs[i] = plans[i].snn_pipeline(inputs[i][0], want_power=True, stages=stages, out=outs[i])
    return outs[i]


def tail(i):
    return runtime.doa_error(outs[i]["argmax"], doa_list, inputs[i][1], groups=S, want_err=False)[1]


replay = pipe.capture(lambda plan: (body(plans.index(plan)), tail(plans.index(plan))))
lane = torch.cuda.Stream(device=dev)
ev_a = [torch.cuda.Event() for _ in range(nstreams)]
ev_b = [torch.cuda.Event() for _ in range(nstreams)]


def step_eager(k):
    i = k % nstreams
    with torch.cuda.stream(pipe.streams[i]):
        body(i)
        tail(i)


def step_lane(k):
    i = k % nstreams
    s = pipe.streams[i]
    with torch.cuda.stream(s):
        body(i, 1)
        ev_a[i].record(s)
    lane.wait_event(ev_a[i])
    with torch.cuda.stream(lane):
        body(i, 2)
        ev_b[i].record(lane)
    s.wait_event(ev_b[i])
    with torch.cuda.stream(s):
        body(i, 4)
        tail(i)


import ctypes  # noqa: E402

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


def make_scan_lane(scan_stream, work_streams):
    # only the serial scan on the lane; zero fill, chunk pass, fallback stay on the step's own stream
    def step(k):
        i = k % nstreams
        s = work_streams[i]
        with torch.cuda.stream(s):
            body(i, 1)
            ev_a[i].record(s)
        scan_stream.wait_event(ev_a[i])
        with torch.cuda.stream(scan_stream):
            body(i, 8)
            ev_b[i].record(scan_stream)
        s.wait_event(ev_b[i])
        with torch.cuda.stream(s):
            body(i, 16 | 4)
            tail(i)
    return step


def timed(fn, name):
    for k in range(nstreams):
        fn(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        fn(k)
    torch.cuda.synchronize()
    print(name, "%.3f ms/step" % ((time.perf_counter() - t0) / steps * 1e3), flush=True)


ref = replay(0)[0]["argmax"].clone()
only = sys.argv[3] if len(sys.argv) > 3 else ""
if not only:
    for rep in range(2):
        timed(lambda k: replay(), "graph")
        timed(step_eager, "eager")
        timed(step_lane, "lane ")
    timed(make_scan_lane(lane, pipe.streams), "scan on its own stream (unmasked)")
cases = ((32, "block"), (32, "spread"), (28, "spread"), (64, "spread"))
if only and only != "graph":
    cases = tuple((int("".join(ch for ch in c if ch.isdigit())), "".join(ch for ch in c if not ch.isdigit())) for c in only.split(","))
if only:
    timed(lambda k: replay(), "graph")
for ncu, mode in cases:
    if only == "graph":
        break
    sel = list(range(ncu)) if mode == "block" else list(range(NCU - ncu, NCU)) if mode == "top" else sorted(set(int(round(j * NCU / ncu)) for j in range(ncu)))
    rest = [c for c in range(NCU) if c not in set(sel)]
    sl = masked_stream(sel)
    ws_ = [masked_stream(rest) for _ in range(nstreams)]
    timed(make_scan_lane(sl, ws_), f"scan on {len(sel)} CUs ({mode}), the rest on {len(rest)}")
    torch.cuda.synchronize()
step_lane(0)
torch.cuda.synchronize()
print("argmax equal:", bool((outs[0]["argmax"] == ref).all()))
