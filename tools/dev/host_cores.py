"""How the headline step depends on the host: cores the process may use (--cpu-cores) and how the host waits for the GPU
(hipDeviceScheduleSpin, the default, against hipDeviceScheduleBlockingSync / Yield set before the first HIP call).
usage: python tools/dev/host_cores.py   (one child per configuration)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import ctypes, os, sys
flag = int(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
sys.path.insert(0, %r)
import bench
args = bench.parse(sys.argv[1:])
if args.cpu_cores:
    bench.pin_cpu_cores(args.cpu_cores, 0)
if flag:
    import torch
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(flag))
    assert rc == 0, rc
sys.exit(bench.run(args))
""" % ROOT
for cores in tuple(int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["16", "4", "2", "1"])):
    for flag, name in ((0, "default"), (4, "blocking sync"), (2, "yield")):
        cmd = [sys.executable, "-c", CHILD, str(flag), "--cpu-cores", str(cores), "--steps", (sys.argv[2] if len(sys.argv) > 2 else "40"), "--warmup", "5", "--repeats", "3", "--no-cpu-baseline", "--no-other-configs"]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
        try:
            d = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
            print(f"{cores:2d} cores, {name:13s}: {d['ms_per_step']:.4f} ms/step  e2e {d['e2e']['ms_per_step']:.4f}", flush=True)
        except Exception as e:
            print(cores, name, "failed", p.stderr.decode()[-300:], flush=True)
