"""Experiment: the Xylo step (BASELINE config 4) with the chip PARTITIONED between its two halves instead of three free-for-all streams --
front end (STHT, scan, chunked encoder) of step i + 1 on one set of CUs beside the integer LIF of step i on the other
(hipExtStreamCreateWithCUMask), plain launches, two plans alternating.  Prints ms per step for several splits and for unmasked streams.
usage: python tools/dev/xylo_partition.py [steps]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from haghighatshoarmuir2024_amd import runtime, synthesis
from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray
from haghighatshoarmuir2024_amd.xylo_snn_localization import Demo

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
NCU = torch.cuda.get_device_properties(0).multi_processor_count

def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)

fs, M, G, B = 48_000, 7, 360, 1100
geometry = CenterCircularArray(radius=4.5e-2, num_mic=M)
doa_list = np.linspace(-np.pi, np.pi, G)
demo = Demo(geometry=geometry, freq_bands=[[1000.0, 2000.0]], doa_list=doa_list, recording_duration=0.25, bipolar_spikes=True, fs=fs, device=dev)
t = np.arange(0, 1.0, step=1 / fs)
sig = np.sin(2 * np.pi * np.cumsum(1000.0 + 1000.0 * (t % t[-1]) / t[-1]) / fs)
rng = np.random.RandomState(2000)
doa = rng.rand(B) * 2 * np.pi
snr_db = np.linspace(-10, 20, 11)[(np.arange(B) * 11) // B] - 10 * np.log10(24.0)
x = synthesis.signal_from_template_batch(geometry, (t, sig), doa, device=dev, device_delays=True)
synthesis.add_noise_(x, snr_db, seed=4321)
net = demo.network()
enc = demo.beamfs[0].spk_encoder
bb, aa = demo.filterbank.ba_list[0]
plans = [runtime.Plan(M, demo.beamfs[0].kernel, bb, aa, enc.robust_width, enc.bipolar, device=dev) for _ in range(3)]
outs = [p.snn_pipeline(x, want_spikes=True, want_power=False, stages=3) for p in plans]
win = 2 * ((G // 32) // 2) + 1
for o in outs:
    net.run(o["spikes"], ternary=True, queued=False)
torch.cuda.synchronize()

def run(front, back, nplans, label):
    ev_f = [torch.cuda.Event() for _ in range(nplans)]
    ev_b = [torch.cuda.Event() for _ in range(nplans)]
    def step(i):
        k = i % nplans
        with torch.cuda.stream(front):
            front.wait_event(ev_b[k])  # the LIF that read this plan's raster last has finished
            plans[k].snn_pipeline(x, want_spikes=True, want_power=False, stages=3, out=outs[k])
            ev_f[k].record(front)
        with torch.cuda.stream(back):
            back.wait_event(ev_f[k])
            counts = net.run(outs[k]["spikes"], ternary=True, queued=False)[1]
            runtime.peak_location(counts, G, win)
            ev_b[k].record(back)
    for i in range(nplans + 1): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step", flush=True)

plain = [torch.cuda.Stream(device=dev) for _ in range(2)]
run(plain[0], plain[1], 2, "two unmasked streams (front end | LIF), 2 plans")
run(plain[0], plain[1], 3, "two unmasked streams (front end | LIF), 3 plans")
# compute units [0, k) of every XCD for the LIF, [k, 32) for the front end (mask layout of gfx950: micloc_stream_create_cu_range)
keep = []
for k in (int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["20"])):
    back, front = runtime.CuRangeStream(dev, 0, k), runtime.CuRangeStream(dev, k, 32)
    keep += [back, front]
    run(front.stream, back.stream, 2, f"masked: LIF on {k} CUs per XCD, front end on {32 - k}, 2 plans")
    run(front.stream, back.stream, 3, f"masked: LIF on {k} CUs per XCD, front end on {32 - k}, 3 plans")
