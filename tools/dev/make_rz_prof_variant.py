"""Builds tools/_variants/libmicloc_hip_prof.so: rzcc.hip with per-wave work / barrier-wait cycle counters (measurement only).
Every __syncthreads() of the encoder kernel and its loaders is bracketed by s_memtime reads; lane 0 of each wave adds its totals to
a device array indexed by the hardware wave id, read back through micloc_debug_rz_prof()."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "haghighatshoarmuir2024_amd", "csrc")
s = open(os.path.join(CS, "rzcc.hip")).read()
pre = '''
__device__ unsigned long long rz_prof_work[8], rz_prof_wait[8], rz_prof_n[8];
#define RZ_PROF_DECL long long prof_last = clock64(), prof_work = 0, prof_wait = 0
#define RZ_SYNC() do { const long long _t0 = clock64(); prof_work += _t0 - prof_last; __syncthreads(); prof_last = clock64(); prof_wait += prof_last - _t0; } while (0)
#define RZ_FLUSH() do { if ((threadIdx.x & 63) == 0) { atomicAdd(&rz_prof_work[threadIdx.x >> 6], (unsigned long long)prof_work); atomicAdd(&rz_prof_wait[threadIdx.x >> 6], (unsigned long long)prof_wait); atomicAdd(&rz_prof_n[threadIdx.x >> 6], 1ull); } } while (0)
'''
s = s.replace("constexpr int RZ_RING = 64;", pre + "constexpr int RZ_RING = 64;")

def patch(body, decl_after):
    body = body.replace(decl_after, decl_after + "\n    RZ_PROF_DECL;", 1)
    body = body.replace("__syncthreads();", "RZ_SYNC();")
    return body

# loaders (rz_loader_np only: the spikes-only launches)
a = s.index("template <int NP, int PHASE, int SW, typename XT>")
b = s.index("template <int N, unsigned ZMASK = 0u>\n__global__ __launch_bounds__(192) void rzcc_scan_kernel")
body = s[a:b]
body = patch(body, "    constexpr int phase = PHASE;")
# flush at the end of the function: before its closing brace
idx = body.rstrip().rfind("}")
body = body[:idx] + "    RZ_FLUSH();\n}\n\n"
s = s[:a] + body + s[b:]
# the scan kernel must keep plain barriers with its loaders: it calls rz_loader_np too -> fine (they flush, harmless)
a = s.index("template <int N, bool WANT_PRE, bool WANT_SPIKES, int RING = RZ_RING, bool WRITER = false, int SW = 64>")
b = s.index("// ---------------------------------------------------------------------------------------------------\n// Fallback for flagged")
body = s[a:b]
body = patch(body, "    const int wave_hw = threadIdx.x >> 6;")
# returns of the loader branches come before any prof use; other returns flush
body = re.sub(r"\n(\s+)return;", r"\n\1{ RZ_FLUSH(); return; }", body)
# undo for the loader-branch returns (they flushed inside the loader)
body = body.replace("lane);\n            { RZ_FLUSH(); return; }", "lane);\n            return;")
body = body.replace("lane);\n        { RZ_FLUSH(); return; }", "lane);\n        return;")
# the kernel's natural end (select waves, one-pass form)
idx = body.rstrip().rfind("}")
body = body[:idx] + "    RZ_FLUSH();\n}\n\n"
s = s[:a] + body + s[b:]
s = s.replace("}  // namespace micloc", '''}  // namespace micloc
extern "C" int micloc_debug_rz_prof(unsigned long long *out24, int reset)
{
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(micloc::rz_prof_work), 64) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out24 + 8, HIP_SYMBOL(micloc::rz_prof_wait), 64) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out24 + 16, HIP_SYMBOL(micloc::rz_prof_n), 64) != hipSuccess) return -1;
    if (reset) {
        (void)hipMemcpyToSymbol(HIP_SYMBOL(micloc::rz_prof_work), z, 64);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(micloc::rz_prof_wait), z, 64);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(micloc::rz_prof_n), z, 64);
    }
    return 0;
}
''')
# ablations (wrong results, timing attribution only): --ablate nostore | noresolve | nodetectloop
abl = sys.argv[sys.argv.index("--ablate") + 1] if "--ablate" in sys.argv else ""
if "nostore" in abl:
    s = s.replace("            sp[(size_t)pos * C] = mark;\n", "            asm volatile(\"\" ::\"v\"(pos));\n")
if "noresolve" in abl:
    s = s.replace("            resolve_cluster(s, e, stride, w, emit, word_at, val_at, sgn);", "            emit(lastpos);")
if "nodetectloop" in abl:
    s = s.replace("                while (__any(Ew != 0u)) {\n                    if (Ew) {", "                while (false) {\n                    if (Ew) {")
if "noA" in abl:   # detect: no fp64 compares (rise / fall words constant)
    old = "                    rise_fall_step(Rw, Fw, c[jj], pj);  // ordered > / ordered <"
    assert old in s
    s = s.replace(old, "                    Rw = 0x1111u + (unsigned)(pj > 1e300); Fw = 0x4444u;")
if "noC" in abl:   # detect: no events appended
    old = "                Ew = livel ? Ew : 0u;"
    assert old in s
    s = s.replace(old, "                Ew = 0u;")
if "size2" in abl:
    s = s.replace("""        if (e - s <= stride)
            emit(lastpos);
        else
            resolve_cluster(s, e, stride, w, emit, word_at, val_at, sgn);""", """        if (e - s <= stride)
            emit(lastpos);
        else if (e - s <= 2 * stride) {
            const double v0 = *val_at(s) * sgn, v1 = *val_at(s + stride) * sgn;
            emit(v1 >= v0 ? lastpos : first);
        } else
            resolve_cluster(s, e, stride, w, emit, word_at, val_at, sgn);""")
    assert "v1 >= v0 ? lastpos : first" in s
if "no16" in abl:
    s = s.replace("""    if (remaining <= 16) {
        resolve_cluster_regs<16>(s, stride, remaining, w, emit, word_at, val_at, sgn);
        return;
    }""", "")
if "only4" in abl:
    s = s.replace("""    if (remaining <= 8) {
        resolve_cluster_regs<8>(s, stride, remaining, w, emit, word_at, val_at, sgn);
        return;
    }
    if (remaining <= 16) {
        resolve_cluster_regs<16>(s, stride, remaining, w, emit, word_at, val_at, sgn);
        return;
    }""", "")
tmp = os.path.join(CS, "rzcc_prof_tmp_%s.hip" % (abl or "plain"))
obj = "/tmp/rzcc_prof_%s.o" % (abl or "plain")
open(tmp, "w").write(s)
try:
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-result".split()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", "-o", obj, tmp])
finally:
    if "--keep" not in sys.argv:
        os.remove(tmp)
objs = [os.path.join(CS, o) for o in "api.o stht.o beamform.o xylo.o synth.o covariance.o beamform_f32.o sweep.o rng.o design.o".split()]
os.makedirs(os.path.join(ROOT, "tools", "_variants"), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "tools", "_variants", "libmicloc_hip_prof%s.so" % (("_" + abl) if abl else "")), obj] + objs)
print("built variant", abl or "(plain)")
