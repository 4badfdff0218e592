#!/usr/bin/env python3
"""What a non-root process on the GPU box can read about the shader clock and the socket power while kernels run: amdsmi
(in-process) and the amdgpu hwmon / pp_dpm files in sysfs.  bench.py's `sustained` block uses whichever works (see GpuTelemetry).

    python tools/dev/clock_sources.py
"""
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

import torch  # noqa: E402

props = torch.cuda.get_device_properties(0)
print("device:", props.name, "CUs", props.multi_processor_count, "clock_rate", getattr(props, "clock_rate", None))
bdf = None
try:
    bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
except Exception as e:
    print("no pci ids on the device properties:", e)
print("bdf:", bdf)

x = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)


def busy(seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            torch.mm(x, x)
        torch.cuda.synchronize()


print("---- amdsmi")
try:
    import amdsmi

    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("handles:", len(hs))
    for i, h in enumerate(hs):
        try:
            print(i, amdsmi.amdsmi_get_gpu_device_bdf(h))
        except Exception as e:
            print(i, "bdf:", e)
    h = hs[0]
    for phase in ("idle", "busy"):
        if phase == "busy":
            import threading

            th = threading.Thread(target=busy, args=(3.0,))
            th.start()
            time.sleep(1.5)
        for name, fn in (("clock_info GFX", lambda: amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)),
                         ("power_info", lambda: amdsmi.amdsmi_get_power_info(h)),
                         ("gpu_activity", lambda: amdsmi.amdsmi_get_gpu_activity(h)),
                         ("gpu_metrics", lambda: {k: v for k, v in amdsmi.amdsmi_get_gpu_metrics_info(h).items()
                                                  if any(s in k for s in ("gfxclk", "socket_power", "temperature_hotspot", "throttle", "current_gfxclks"))})):
            try:
                t0 = time.perf_counter()
                v = fn()
                print(phase, name, json.dumps(v, default=str)[:600], f"({(time.perf_counter() - t0) * 1e3:.2f} ms)")
            except Exception as e:
                print(phase, name, "FAILED:", type(e).__name__, e)
        if phase == "busy":
            th.join()
except Exception as e:
    print("amdsmi unavailable:", type(e).__name__, e)

print("---- sysfs")
for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
    dev = os.path.realpath(os.path.join(card, "device"))
    print(card, "->", os.path.basename(dev))
    for f in ["pp_dpm_sclk"] + sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*", "*"))):
        p = f if os.path.isabs(f) else os.path.join(dev, f)
        if os.path.isfile(p) and any(s in os.path.basename(p) for s in ("sclk", "freq1_input", "power1_average", "power1_input", "power1_cap", "temp1_input")):
            try:
                print("   ", os.path.basename(p), open(p).read().strip().replace("\n", " | ")[:200])
            except Exception as e:
                print("   ", os.path.basename(p), "unreadable:", e)
    break
