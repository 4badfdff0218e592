#!/usr/bin/env python3
"""Build a measurement variant of the library: tools/_variants/libmicloc_hip_<name>.so.

    python tools/dev/make_variant.py stht_valu      stride-2 STHT kernels on the vector ALU (VARIANT_STHT_VECTOR_FORM)
    python tools/dev/make_variant.py ws_k4          beamform_ws_kernel with four k-steps whatever the channel count (VARIANT_WS_FOUR_KSTEPS)
    python tools/dev/make_variant.py stht_one_tile  the matrix-core STHT with one time tile per workgroup instead of the walk (VARIANT_STHT_ONE_TILE)
    python tools/dev/make_variant.py stht_wide2     the 480-tap walking STHT with two time tiles per wave, one workgroup per CU (VARIANT_STHT_WIDE_TWO_TILES)
    python tools/dev/make_variant.py ws_sparse_lif  beamform_ws_kernel's LIF stage event by event on the vector ALU instead of the dense Toeplitz product
                                                    (round 5's rejected experiment: NOT in the product sources -- tools/experiments/ws_sparse_lif/ws_sparse_lif.patch
                                                    is applied to a copy of csrc/beamform.hip)

The shipped library has no run-time switches: a variant is the same sources with ONE constant of csrc/micloc_internal.h flipped, or (an
experiment that was rejected and moved out of the product) with a patch of tools/experiments/ applied to the copy.
Only the translation unit that reads the constant is recompiled (in build_dev/variants/<name>/); the other objects are the product's.
A/B runs load it with `MICLOC_DEV_LIB=<path>` understood by the tools (never by the package), the stht_valu one is also what
tests/test_hip_parity.py::test_stht_vector_form_still_exact loads."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "haghighatshoarmuir2024_amd", "csrc")
VARIANTS = {"stht_wide2": ("VARIANT_STHT_WIDE_TWO_TILES", "stht"), "stht_valu": ("VARIANT_STHT_VECTOR_FORM", "stht"), "ws_k4": ("VARIANT_WS_FOUR_KSTEPS", "beamform"), "stht_one_tile": ("VARIANT_STHT_ONE_TILE", "stht"),
            "ws_sparse_lif": (os.path.join(ROOT, "tools", "experiments", "ws_sparse_lif", "ws_sparse_lif.patch"), "beamform")}
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result".split()
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(name):
    const, unit = VARIANTS[name]
    out = os.path.join(ROOT, "tools", "_variants", f"libmicloc_hip_{name}.so")
    srcs = [os.path.join(CS, f) for f in (unit + ".hip", "micloc_internal.h", "synth_dev.h")] + [os.path.join(ROOT, "include", "micloc_hip.h")]
    if const.endswith(".patch"):
        srcs.append(const)
    objs = [os.path.join(CS, f) for f in sorted(os.listdir(CS)) if f.endswith(".o") and f != unit + ".o"]
    newest = max(os.path.getmtime(p) for p in srcs + objs + [os.path.abspath(__file__)])
    if os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    work = os.path.join(ROOT, "build_dev", "variants", name, "csrc")  # (micloc_internal.h includes ../../include/micloc_hip.h)
    os.makedirs(work, exist_ok=True)
    inc = os.path.join(ROOT, "build_dev", "variants", "include")
    os.makedirs(inc, exist_ok=True)
    shutil.copy(os.path.join(ROOT, "include", "micloc_hip.h"), inc)
    for f in (unit + ".hip", "synth_dev.h"):
        shutil.copy(os.path.join(CS, f), work)
    hdr = open(os.path.join(CS, "micloc_internal.h")).read()
    if const.endswith(".patch"):
        # an experiment kept outside the product: the patch touches the unit's translation unit only (no shared declaration changes)
        open(os.path.join(work, "micloc_internal.h"), "w").write(hdr)
        subprocess.check_call(["patch", "-s", "-p1", "-d", work, "-i", const])
    else:
        needle = f"constexpr bool {const} = false;"
        if hdr.count(needle) != 1:
            raise SystemExit(f"make_variant: '{needle}' not found exactly once in micloc_internal.h")
        open(os.path.join(work, "micloc_internal.h"), "w").write(hdr.replace(needle, f"constexpr bool {const} = true;"))
    obj = os.path.join(work, unit + ".o")
    subprocess.check_call([HIPCC] + FLAGS + ["-c", "-o", obj, os.path.join(work, unit + ".hip")])
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objs)
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    for n in names:
        if n not in VARIANTS:
            raise SystemExit(f"unknown variant {n}: one of {', '.join(VARIANTS)}")
        print(build(n))
