// C-ABI of libmicloc_hip.so (see include/micloc_hip.h for the contract and reference citations).
#include <math.h>
#include <string.h>

#include <new>
#include <vector>

#include "micloc_internal.h"

using namespace micloc;

static thread_local int g_last_hip = 0;

// (a failed runtime call also leaves its code in HIP's sticky "last error": clear it, or the next, unrelated launch
// check -- hipGetLastError() after a successful launch -- would report it again)
#define HIP_TRY(expr)                   \
    do {                                \
        hipError_t _e = (expr);         \
        if (_e != hipSuccess) {         \
            g_last_hip = (int)_e;       \
            (void)hipGetLastError();    \
            return MICLOC_ERR_HIP;      \
        }                               \
    } while (0)

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct micloc_plan {
    int device = 0;
    int M = 0, C = 0, L = 0;
    int robust_width = 1, bipolar = 0;
    IirCoef iir{};
    // STHT taps
    double *d_taps = nullptr;
    SthtTaps taps{};
    // neuron kernel
    double *d_ntab = nullptr;
    NeuronTab ntab{};
    // beamforming matrices
    double *d_W = nullptr;
    BeamformW W{};
    int W_is_complex = 0;
    int G_out = 0;  // DoA grid size seen by the caller
    // bumped whenever a captured hipGraph of this plan becomes stale: a device table was re-allocated (the graph holds the
    // old pointer) or replaced by one of another shape (the graph holds the old dimensions BY VALUE: W.GT / W.G / W.CT,
    // ntab.n / NK are kernel arguments, and the packed bf_mat layout depends on Gp)
    int generation = 0;
    int chunk_frames = 0;  // encoder time chunking: 0 automatic, < 0 off, > 0 owned frames per chunk
    size_t taps_cap = 0, ntab_cap = 0, W_cap = 0;  // allocated doubles
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            changed = hipSetDevice(dev) == hipSuccess;
        }
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

// Device that owns a device pointer (plan-less entry points: their launches belong to the device of their buffers and
// stream, whatever the caller's current device is).  Falls back to the current device for pointers HIP does not know.
int device_of(const void *ptr)
{
    hipPointerAttribute_t at;
    int dev = 0;
    if (ptr && hipPointerGetAttributes(&at, ptr) == hipSuccess) return at.device;
    (void)hipGetLastError();
    (void)hipGetDevice(&dev);
    return dev;
}

// Uploads a host table.  A table that fits the existing allocation is overwritten in place (device pointer unchanged:
// hipGraphs captured against this plan stay valid IF the caller keeps the table's shape -- set_W / set_neuron_kernel bump
// the generation themselves when the shape changes); otherwise the buffer is re-allocated and *generation is bumped.
int upload(double **dptr, size_t *cap, int *generation, const std::vector<double> &host)
{
    const size_t n = host.empty() ? 1 : host.size();
    if (!*dptr || *cap < n) {
        if (*dptr) {
            HIP_TRY(hipFree(*dptr));
            *dptr = nullptr;
            *cap = 0;
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(dptr), n * sizeof(double)));
        *cap = n;
        ++*generation;
    }
    if (!host.empty()) HIP_TRY(hipMemcpy(*dptr, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
    return MICLOC_OK;
}

int pad_ct(int C)
{
    int ct = (C + 15) / 16;
    if (ct > 4 && ct <= 8) ct = 8;
    return ct;
}

struct WsLayout {
    size_t h, pre, scratch, spikes, partial, total;
};

WsLayout ws_layout(const micloc_plan *p, int B, int T)
{
    WsLayout w{};
    const size_t Ts = (size_t)micloc_padded_T(T);
    const size_t planar = align256((size_t)B * p->C * Ts * sizeof(double));
    int Gp = p->W.GT > 0 ? 16 * p->W.GT : 16;
    size_t off = 0;
    w.h = off;
    off += planar;
    w.pre = off;
    off += planar;
    w.scratch = off;
    off += rzcc_scratch_bytes(B * p->C, T, p->robust_width, p->chunk_frames);
    w.spikes = off;
    off += align256((size_t)B * T * p->C);
    w.partial = off;
    {
        const size_t pb = beamform_partial_bytes(B, T, Gp);
        const int ct = pad_ct(p->C);
        const size_t pc = ct <= 8 ? cov_partial_bytes(B, T, ct) : 0;
        off += pb > pc ? pb : pc;
    }
    w.total = off;
    return w;
}

// trials map to gridDim.y / gridDim.z, which HIP limits to 65535
bool bad_batch(int B) { return B < 1 || B > 65535; }

bool bad_ws(const void *ws, size_t have, size_t need)
{
    return ws == nullptr || have < need || (reinterpret_cast<uintptr_t>(ws) & 255) != 0;
}

}  // namespace

extern "C" {

int micloc_abi_version(void) { return MICLOC_ABI_VERSION; }

int micloc_last_hip_error(void) { return g_last_hip; }

const char *micloc_status_string(int status)
{
    switch (status) {
        case MICLOC_OK: return "ok";
        case MICLOC_ERR_INVALID: return "invalid argument";
        case MICLOC_ERR_SHAPE: return "shape mismatch";
        case MICLOC_ERR_WORKSPACE: return "workspace too small or misaligned (256-byte alignment required)";
        case MICLOC_ERR_NOT_SET: return "neuron kernel or beamforming matrix not set on the plan";
        case MICLOC_ERR_HIP: return "HIP runtime error (see micloc_last_hip_error)";
        case MICLOC_ERR_NO_DEVICE: return "no usable HIP device";
        default: return "unknown status";
    }
}

int micloc_padded_T(int T) { return T <= 0 ? 8 : (T + 7) & ~7; }

int micloc_plan_create(const micloc_config *cfg, micloc_plan **out)
{
    if (!cfg || !out) return MICLOC_ERR_INVALID;
    *out = nullptr;
    if (cfg->num_mic <= 0 || cfg->stht_len <= 0 || !cfg->stht_kernel) return MICLOC_ERR_INVALID;
    if (cfg->iir_len < 1 || cfg->iir_len > MICLOC_MAX_IIR || !cfg->iir_b || !cfg->iir_a) return MICLOC_ERR_INVALID;
    if (cfg->iir_a[0] == 0.0 || cfg->robust_width < 1) return MICLOC_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
        return MICLOC_ERR_NO_DEVICE;
    {
        // the code objects in this library are gfx950 only
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return MICLOC_ERR_NO_DEVICE;
    }

    micloc_plan *p = new (std::nothrow) micloc_plan();
    if (!p) return MICLOC_ERR_INVALID;
    p->device = cfg->device;
    p->M = cfg->num_mic;
    p->C = 2 * cfg->num_mic;
    p->L = cfg->stht_len;
    p->robust_width = cfg->robust_width;
    p->bipolar = cfg->bipolar ? 1 : 0;
    p->iir.n = cfg->iir_len;
    for (int i = 0; i < MICLOC_MAX_IIR; ++i) {
        p->iir.b[i] = i < cfg->iir_len ? cfg->iir_b[i] / cfg->iir_a[0] : 0.0;
        p->iir.a[i] = i < cfg->iir_len ? cfg->iir_a[i] / cfg->iir_a[0] : 0.0;
    }

    // compact tap table: delays klo, klo + kstep, ... ; exact-zero taps are dropped when every second tap is zero
    const int L = cfg->stht_len;
    int klo = -1, khi = -1;
    for (int k = 0; k < L; ++k)
        if (cfg->stht_kernel[k] != 0.0) {
            if (klo < 0) klo = k;
            khi = k;
        }
    SthtTaps tp{};
    std::vector<double> compact;
    if (klo < 0) {
        tp.klo = 0;
        tp.kstep = 2;
        tp.ngroups = 0;
        compact.assign(4, 0.0);
    } else {
        bool stride2 = true;
        for (int k = klo; k <= khi; ++k)
            if (((k - klo) & 1) && cfg->stht_kernel[k] != 0.0) stride2 = false;
        tp.kstep = stride2 ? 2 : 1;
        tp.klo = klo;
        const int ntaps = (khi - klo) / tp.kstep + 1;
        const int U = STHT_R / tp.kstep;
        tp.ngroups = (ntaps + U - 1) / U;
        compact.assign((size_t)(tp.ngroups + 1) * U, 0.0);  // +1 all-zero group: the kernel prefetches one group ahead
        for (int j = 0; j < ntaps; ++j) compact[j] = cfg->stht_kernel[klo + j * tp.kstep];
    }
    {
        const int U = STHT_R / tp.kstep;
        const int kmax = tp.klo + (tp.ngroups * U > 0 ? (tp.ngroups * U - 1) * tp.kstep : 0);
        tp.halo = tp.klo + STHT_R * ((kmax - tp.klo + STHT_R - 1) / STHT_R);
        tp.shift = L / 2;
    }
    if (stht_lds_bytes(tp, p->M) > 160 * 1024) {
        delete p;
        return MICLOC_ERR_INVALID;  // kernel too long for the LDS-staged tile
    }

    DeviceGuard guard(p->device);
    int rc = upload(&p->d_taps, &p->taps_cap, &p->generation, compact);
    if (rc != MICLOC_OK) {
        delete p;
        return rc;
    }
    tp.taps = p->d_taps;
    p->taps = tp;
    *out = p;
    return MICLOC_OK;
}

void micloc_plan_destroy(micloc_plan *p)
{
    if (!p) return;
    DeviceGuard guard(p->device);
    if (p->d_taps) (void)hipFree(p->d_taps);
    if (p->d_ntab) (void)hipFree(p->d_ntab);
    if (p->d_W) (void)hipFree(p->d_W);
    delete p;
}

int micloc_plan_set_neuron_kernel(micloc_plan *p, const double *nir, int n)
{
    if (!p || !nir || n < 1) return MICLOC_ERR_INVALID;
    const int NK = (n + 15 + 3) / 4;
    if ((size_t)(256 + 4 * NK) * 16 * pad_ct(p->C) > 96 * 1024) return MICLOC_ERR_INVALID;  // the int8 spike tile of a sub-chunk must fit in LDS
    std::vector<double> tab((size_t)4 * NK + 16, 0.0);
    for (int k = 0; k < n; ++k) tab[(size_t)k + 15] = nir[k];
    DeviceGuard guard(p->device);
    // the table may be in use by queued kernels: replace it only after the device went idle
    HIP_TRY(hipDeviceSynchronize());
    int rc = upload(&p->d_ntab, &p->ntab_cap, &p->generation, tab);
    if (rc != MICLOC_OK) return rc;
    if (p->ntab.tab && (p->ntab.n != n || p->ntab.NK != NK)) ++p->generation;  // captured graphs hold n / NK by value
    p->ntab.tab = p->d_ntab;
    p->ntab.n = n;
    p->ntab.NK = NK;
    return MICLOC_OK;
}

static int set_W(micloc_plan *p, const std::vector<double> &Wp, int CT, int GT, int C, int Gcols, int is_complex,
                 int G_out)
{
    DeviceGuard guard(p->device);
    HIP_TRY(hipDeviceSynchronize());
    int rc = upload(&p->d_W, &p->W_cap, &p->generation, Wp);
    if (rc != MICLOC_OK) return rc;
    // same allocation, other shape: a captured graph would read the new table with the old stride and dimensions
    if (p->W.Wp && (p->W.CT != CT || p->W.GT != GT || p->W.C != C || p->W.G != Gcols || p->W_is_complex != is_complex || p->G_out != G_out))
        ++p->generation;
    p->W.Wp = p->d_W;
    p->W.CT = CT;
    p->W.GT = GT;
    p->W.C = C;
    p->W.G = Gcols;
    p->W.complex_pairs = is_complex;
    p->W_is_complex = is_complex;
    p->G_out = G_out;
    return MICLOC_OK;
}

int micloc_plan_set_bf_mat(micloc_plan *p, const double *W, int C, int G)
{
    if (!p || !W || G < 1) return MICLOC_ERR_INVALID;
    if (C != p->C) return MICLOC_ERR_SHAPE;
    const int CT = pad_ct(C);
    if (CT > 8) return MICLOC_ERR_INVALID;
    const int GT = (G + 15) / 16;
    const int Gp = 16 * GT;
    // (+16: the general kernel streams bf_mat in slabs of two DoA tiles and reads a whole slab even when the last one is half empty)
    std::vector<double> Wp((size_t)16 * CT * Gp + 16, 0.0);
    for (int c = 0; c < C; ++c)
        for (int g = 0; g < G; ++g) Wp[(size_t)c * Gp + g] = W[(size_t)c * G + g];
    return set_W(p, Wp, CT, GT, C, G, 0, G);
}

int micloc_plan_set_bf_mat_c128(micloc_plan *p, const double *Wre, const double *Wim, int M, int G)
{
    if (!p || !Wre || !Wim || G < 1) return MICLOC_ERR_INVALID;
    if (M != p->M) return MICLOC_ERR_SHAPE;
    const int C = 2 * M;
    const int CT = pad_ct(C);
    if (CT > 8) return MICLOC_ERR_INVALID;
    const int Ghp = 16 * ((G + 15) / 16);  // Gp / 16 is even by construction
    const int Gp = 2 * Ghp;
    // (hr + j hi)(wr - j wi):  re = [hr hi] . [wr; wi]   im = [hr hi] . [-wi; wr]
    std::vector<double> Wp((size_t)16 * CT * Gp + 16, 0.0);
    for (int m = 0; m < M; ++m)
        for (int g = 0; g < G; ++g) {
            const double wr = Wre[(size_t)m * G + g], wi = Wim[(size_t)m * G + g];
            Wp[(size_t)m * Gp + g] = wr;
            Wp[(size_t)(M + m) * Gp + g] = wi;
            Wp[(size_t)m * Gp + Ghp + g] = -wi;
            Wp[(size_t)(M + m) * Gp + Ghp + g] = wr;
        }
    return set_W(p, Wp, CT, Gp / 16, C, 2 * G, 1, G);
}

int micloc_plan_generation(const micloc_plan *p) { return p ? p->generation : -1; }

// ---- streams restricted to a part of every XCD ------------------------------------------------------------------------
int micloc_stream_create_cu_range(int device, int cu_lo, int cu_hi, void **stream)
{
    if (!stream || device < 0) return MICLOC_ERR_INVALID;
    *stream = nullptr;
    DeviceGuard guard(device);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MICLOC_ERR_INVALID;
    // gfx950: 8 XCDs; bit i of the mask is compute unit i / 8 of XCD i % 8 (measured: tools/dev/cu_mask_probe.hip), and an XCD
    // whose bits are all clear is NOT excluded -- it runs the stream on all of its compute units -- so a range is given per XCD
    // The layout was measured on the unpartitioned device (SPX: 256 compute units in 8 XCDs) and holds for that only: a partition
    // (CPX: 32 compute units, one XCD) has the same arch name and a mask that means something else -- refuse, callers fall back to
    // ordinary streams (runtime.StreamPipeline / sweep: scan_lane = 0).
    constexpr int NXCD = 8;
    const int ncu = prop.multiProcessorCount;
    if (ncu != 256) return MICLOC_ERR_INVALID;
    const int per = ncu / NXCD;
    if (cu_lo < 0 || cu_hi > per || cu_lo >= cu_hi) return MICLOC_ERR_INVALID;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int c = cu_lo; c < cu_hi; ++c)
        for (int x = 0; x < NXCD; ++x) {
            const int bit = c * NXCD + x;
            mask[bit / 32] |= 1u << (bit % 32);
        }
    hipStream_t st = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    *stream = st;
    return MICLOC_OK;
}

int micloc_stream_destroy(void *stream)
{
    if (!stream) return MICLOC_ERR_INVALID;
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_plan_set_encoder_chunk(micloc_plan *p, int chunk_frames)
{
    if (!p) return MICLOC_ERR_INVALID;
    p->chunk_frames = chunk_frames;
    return MICLOC_OK;
}

int micloc_plan_encoder_chunks(const micloc_plan *p, int B, int T)
{
    if (!p || bad_batch(B) || T < 1) return MICLOC_ERR_INVALID;
    return rzcc_chunks(B * p->C, T, p->robust_width, p->chunk_frames);
}

size_t micloc_workspace_bytes(const micloc_plan *p, int B, int T)
{
    if (!p || bad_batch(B) || T < 1) return 0;
    return ws_layout(p, B, T).total;
}

size_t micloc_lif_beamform_workspace_bytes(const micloc_plan *p, int B, int T)
{
    if (!p || bad_batch(B) || T < 1 || p->W.GT < 1) return 0;
    return align256(beamform_partial_bytes(B, T, 16 * p->W.GT));
}

// ---- stages ----------------------------------------------------------------------------------------------
int micloc_stht_f64(const micloc_plan *p, const double *x, int B, int T, double *h, int Ts, void *stream)
{
    if (!p || !x || !h || bad_batch(B) || T < 1) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (Ts != micloc_padded_T(T)) return MICLOC_ERR_SHAPE;
    HIP_TRY(launch_stht(p->taps, x, h, B, T, p->M, Ts, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_bandpass_rzcc_f64(const micloc_plan *p, const double *h, int B, int T, int Ts, double *pre,
                             int8_t *spikes, void *ws, size_t ws_bytes, void *stream)
{
    if (!p || !h || bad_batch(B) || T < 1 || (!pre && !spikes)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (Ts != micloc_padded_T(T)) return MICLOC_ERR_SHAPE;
    const int nl = B * p->C;
    if (spikes && bad_ws(ws, ws_bytes, rzcc_scratch_bytes(nl, T, p->robust_width, p->chunk_frames))) return MICLOC_ERR_WORKSPACE;
    HIP_TRY(launch_bandpass_rzcc(p->iir, h, nl, p->C, T, Ts, p->robust_width, p->bipolar, pre, spikes, ws,
                                 (hipStream_t)stream, nullptr, 0, 0, p->chunk_frames));
    return MICLOC_OK;
}

int micloc_lif_beamform_f64(const micloc_plan *p, const int8_t *spikes, int B, int T, double *y, double *power,
                            int32_t *argmax, void *ws, size_t ws_bytes, void *stream)
{
    if (!p || !spikes || bad_batch(B) || T < 1 || (!y && !power && !argmax)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (!p->d_ntab || !p->d_W) return MICLOC_ERR_NOT_SET;
    if (p->W_is_complex) return MICLOC_ERR_SHAPE;
    const int Gp = 16 * p->W.GT;
    const bool want_power = power || argmax;
    if (want_power && bad_ws(ws, ws_bytes, beamform_partial_bytes(B, T, Gp))) return MICLOC_ERR_WORKSPACE;
    double *partial = want_power ? reinterpret_cast<double *>(ws) : nullptr;
    int nch = 0;
    HIP_TRY(launch_lif_beamform(p->W, p->ntab, spikes, B, T, y, partial, (hipStream_t)stream, &nch));
    if (want_power)
        HIP_TRY(launch_power_argmax(partial, B, T, nch, Gp, p->G_out, 0, 0, power, argmax, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_beamform_c128_f64(const micloc_plan *p, const double *pre, int B, int T, int Ts, double *y, double *power,
                             int32_t *argmax, void *ws, size_t ws_bytes, void *stream)
{
    if (!p || !pre || bad_batch(B) || T < 1 || (!y && !power && !argmax)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (Ts != micloc_padded_T(T)) return MICLOC_ERR_SHAPE;
    if (!p->d_W) return MICLOC_ERR_NOT_SET;
    if (!p->W_is_complex) return MICLOC_ERR_SHAPE;
    const int Gp = 16 * p->W.GT;
    const bool want_power = power || argmax;
    if (want_power && bad_ws(ws, ws_bytes, beamform_partial_bytes(B, T, Gp))) return MICLOC_ERR_WORKSPACE;
    double *partial = want_power ? reinterpret_cast<double *>(ws) : nullptr;
    int nch = 0;
    HIP_TRY(launch_planar_beamform(p->W, pre, B, T, Ts, y, 1, partial, (hipStream_t)stream, &nch));
    if (want_power)
        HIP_TRY(launch_power_argmax(partial, B, T, nch, Gp, p->G_out, 1, Gp / 2, power, argmax, (hipStream_t)stream));
    return MICLOC_OK;
}

// ---- pipelines -------------------------------------------------------------------------------------------
int micloc_snn_pipeline_f64(const micloc_plan *p, const double *x, int B, int T, int8_t *spikes, double *y,
                            double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream)
{
    return micloc_snn_pipeline_stages_f64(p, x, B, T, spikes, y, power, argmax, ws, ws_bytes, stream, MICLOC_STAGE_ALL);
}

int micloc_snn_pipeline_stages_f64(const micloc_plan *p, const double *x, int B, int T, int8_t *spikes, double *y,
                                   double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream, int stages)
{
    if (!p || !x || bad_batch(B) || T < 1 || (!spikes && !y && !power && !argmax)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (stages <= 0 || (stages & ~(MICLOC_STAGE_ALL | MICLOC_STAGE_ENCODE_SCAN | MICLOC_STAGE_ENCODE_REST))) return MICLOC_ERR_INVALID;
    const bool want_bf = y || power || argmax;
    if (want_bf && (!p->d_ntab || !p->d_W)) return MICLOC_ERR_NOT_SET;
    if (want_bf && p->W_is_complex) return MICLOC_ERR_SHAPE;
    const int phases = (stages & MICLOC_STAGE_ENCODE) ? RZ_PHASE_ALL
                                                      : ((stages & MICLOC_STAGE_ENCODE_SCAN) ? RZ_PHASE_SCAN : 0) |
                                                            ((stages & MICLOC_STAGE_ENCODE_REST) ? RZ_PHASE_ENCODE : 0);
    const WsLayout w = ws_layout(p, B, T);
    if (bad_ws(ws, ws_bytes, w.total)) return MICLOC_ERR_WORKSPACE;
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    double *h = reinterpret_cast<double *>(base + w.h);
    int8_t *spk = spikes ? spikes : reinterpret_cast<int8_t *>(base + w.spikes);
    const int Ts = micloc_padded_T(T);
    hipStream_t st = (hipStream_t)stream;
    // the in-phase channels are the rolled input frames: the band-pass kernel reads them from x directly
    if (stages & MICLOC_STAGE_STHT) HIP_TRY(launch_stht(p->taps, x, h, B, T, p->M, Ts, st, false));
    if (phases)
        HIP_TRY(launch_bandpass_rzcc(p->iir, h, B * p->C, p->C, T, Ts, p->robust_width, p->bipolar, nullptr, spk,
                                     base + w.scratch, st, x, p->M, p->taps.shift, p->chunk_frames, phases));
    if (want_bf && (stages & MICLOC_STAGE_BEAMFORM)) {
        const int Gp = 16 * p->W.GT;
        const bool want_power = power || argmax;
        double *partial = want_power ? reinterpret_cast<double *>(base + w.partial) : nullptr;
        int nch = 0;
        HIP_TRY(launch_lif_beamform(p->W, p->ntab, spk, B, T, y, partial, st, &nch));
        if (want_power) HIP_TRY(launch_power_argmax(partial, B, T, nch, Gp, p->G_out, 0, 0, power, argmax, st));
    }
    return MICLOC_OK;
}

int micloc_beamformer_pipeline_f64(const micloc_plan *p, const double *x, int B, int T, double *y, double *power,
                                   int32_t *argmax, void *ws, size_t ws_bytes, void *stream)
{
    if (!p || !x || bad_batch(B) || T < 1 || (!y && !power && !argmax)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (!p->d_W) return MICLOC_ERR_NOT_SET;
    if (!p->W_is_complex) return MICLOC_ERR_SHAPE;
    const WsLayout w = ws_layout(p, B, T);
    if (bad_ws(ws, ws_bytes, w.total)) return MICLOC_ERR_WORKSPACE;
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    double *h = reinterpret_cast<double *>(base + w.h);
    double *pre = reinterpret_cast<double *>(base + w.pre);
    const int Ts = micloc_padded_T(T);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(launch_stht(p->taps, x, h, B, T, p->M, Ts, st, false));
    HIP_TRY(launch_bandpass_rzcc(p->iir, h, B * p->C, p->C, T, Ts, p->robust_width, p->bipolar, pre, nullptr,
                                 nullptr, st, x, p->M, p->taps.shift));
    const int Gp = 16 * p->W.GT;
    const bool want_power = power || argmax;
    double *partial = want_power ? reinterpret_cast<double *>(base + w.partial) : nullptr;
    int nch = 0;
    HIP_TRY(launch_planar_beamform(p->W, pre, B, T, Ts, y, 1, partial, st, &nch));
    if (want_power) HIP_TRY(launch_power_argmax(partial, B, T, nch, Gp, p->G_out, 1, Gp / 2, power, argmax, st));
    return MICLOC_OK;
}

// ---- streaming: the band-pass / RZCC stage tile by tile, exact state hand-off -----------------------------------
size_t micloc_stream_state_bytes(const micloc_plan *p, int B)
{
    if (!p || bad_batch(B)) return 0;
    return rzcc_stream_state_bytes(B * p->C);
}

int micloc_stream_overflow(const void *state, int *count, void *stream)
{
    if (!state || !count) return MICLOC_ERR_INVALID;
    HIP_TRY(hipMemcpyAsync(count, state, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MICLOC_OK;
}

// ---- streaming localisation: LIF + beamforming + power as the spikes become final ---------------------------------------------
// loc_state: [256 B: 64 control ints][acc: B x 2 x G doubles]
size_t micloc_stream_localize_state_bytes(const micloc_plan *p, int B)
{
    if (!p || bad_batch(B) || p->G_out < 1) return 0;
    return 256 + align256((size_t)B * 2 * p->G_out * sizeof(double));
}

int micloc_stream_chunk_frames(const micloc_plan *p)
{
    if (!p || !p->d_ntab || !p->d_W) return MICLOC_ERR_NOT_SET;
    return lif_beamform_chunk_frames(p->W, p->ntab);
}

size_t micloc_stream_localize_workspace_bytes(const micloc_plan *p, int B, int window_frames)
{
    if (!p || bad_batch(B) || window_frames < 1 || p->W.GT < 1) return 0;
    return align256(beamform_partial_bytes(B, window_frames, 16 * p->W.GT));
}

// ---- clocked streaming: the same stages with the absolute time in a device word ----------------------------------------------
// One tile = begin_tile (clock + window slide) -> STHT of [history | tile] (micloc_stht_f64) -> wrap_rows -> encode_tile ->
// localize_tile (which ends with the clock's tick).  No call takes an absolute time: every launch of a tile of n frames has the
// same arguments, so the sequence can be captured into ONE hipGraph and replayed per tile (StreamingLocalizer.capture).
int micloc_stream_reset(const micloc_plan *p, int B, void *enc_state, size_t enc_bytes, void *loc_state, size_t loc_bytes, int8_t *window,
                        int window_frames, void *stream)
{
    if (!p || !enc_state || !loc_state || !window || bad_batch(B) || window_frames < 1) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);
    if (bad_ws(enc_state, enc_bytes, rzcc_stream_state_bytes(B * p->C))) return MICLOC_ERR_WORKSPACE;
    if (bad_ws(loc_state, loc_bytes, micloc_stream_localize_state_bytes(p, B))) return MICLOC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(launch_zero_fill(enc_state, 256, st));  // overflow counter (the encoder's own state is written by the first tile)
    HIP_TRY(launch_zero_fill(loc_state, micloc_stream_localize_state_bytes(p, B), st));  // control words, clock, accumulators
    HIP_TRY(launch_zero_fill(window, (size_t)B * window_frames * p->C, st));
    return MICLOC_OK;
}

int micloc_stream_begin_tile(const micloc_plan *p, void *loc_state, int8_t *window, int8_t *scratch_window, int B, int n, int window_frames,
                             void *stream)
{
    if (!p || !loc_state || !window || !scratch_window || window == scratch_window || bad_batch(B) || n < 1 || window_frames < 1) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);
    if (!p->d_ntab || !p->d_W) return MICLOC_ERR_NOT_SET;
    const int CH = lif_beamform_chunk_frames(p->W, p->ntab);
    if (window_frames % CH != 0 || n > window_frames) return MICLOC_ERR_SHAPE;
    HIP_TRY(launch_stream_begin_tile(reinterpret_cast<int *>(loc_state), window, scratch_window, B, (size_t)window_frames * p->C, p->C, n, window_frames,
                                     CH, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_stream_wrap_rows_f64(const micloc_plan *p, const void *loc_state, double *h, int B, int Ts, int first_col, int n, const double *wrap_tail,
                                void *stream)
{
    if (!p || !loc_state || !h || bad_batch(B) || n < 1 || first_col < 0 || first_col + n > Ts) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);
    HIP_TRY(launch_stht_wrap_rows(h, wrap_tail, B, p->M, Ts, first_col, n, p->L / 2, reinterpret_cast<const int *>(loc_state), (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_stream_encode_tile_f64(const micloc_plan *p, const double *h, int B, int T_tile, int row_stride, int final_tile, int8_t *window,
                                  int window_frames, void *enc_state, size_t enc_bytes, const void *loc_state, void *stream)
{
    if (!p || !h || !window || !loc_state || bad_batch(B) || T_tile < 1 || window_frames < T_tile || row_stride < T_tile) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);
    if (!final_tile && T_tile % 16 != 0) return MICLOC_ERR_SHAPE;
    if (bad_ws(enc_state, enc_bytes, rzcc_stream_state_bytes(B * p->C))) return MICLOC_ERR_WORKSPACE;
    HIP_TRY(launch_stream_encode(p->iir, h, B * p->C, p->C, T_tile, row_stride, p->robust_width, p->bipolar, window, window_frames, 0, 0,
                                 final_tile, enc_state, (hipStream_t)stream, 0, reinterpret_cast<const int *>(loc_state)));
    return MICLOC_OK;
}

int micloc_stream_localize_tile_f64(const micloc_plan *p, const void *enc_state, void *loc_state, size_t loc_bytes, const int8_t *window, int B,
                                    int window_frames, int final_tile, double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream)
{
    if (!p || !enc_state || !loc_state || !window || bad_batch(B) || window_frames < 1) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);
    if (!p->d_ntab || !p->d_W) return MICLOC_ERR_NOT_SET;
    if (p->W_is_complex) return MICLOC_ERR_SHAPE;
    const int CH = lif_beamform_chunk_frames(p->W, p->ntab);
    if (window_frames % CH != 0) return MICLOC_ERR_SHAPE;
    const int G = p->G_out, Gp = 16 * p->W.GT;
    if (bad_ws(loc_state, loc_bytes, micloc_stream_localize_state_bytes(p, B))) return MICLOC_ERR_WORKSPACE;
    if (bad_ws(ws, ws_bytes, beamform_partial_bytes(B, window_frames, Gp))) return MICLOC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int *ctl = reinterpret_cast<int *>(loc_state);
    double *acc = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(loc_state) + 256);
    const int nwin = window_frames / CH;
    HIP_TRY(launch_stream_horizon(enc_state, B * p->C, p->bipolar, 0, final_tile ? 1 : 0, CH, 0, nwin, ctl, st, 1));
    BeamformW W = p->W;
    W.chunk_range = ctl + 4;
    double *partial = reinterpret_cast<double *>(ws);
    int nch = 0;
    // the whole window as the launch shape; the kernel takes the frames that exist from the control words
    HIP_TRY(launch_lif_beamform(W, p->ntab, window, B, window_frames, nullptr, partial, st, &nch));
    HIP_TRY(launch_stream_accumulate(partial, B, nch, Gp, G, ctl + 4, ctl, acc, ctl + 8, power, argmax, st));
    HIP_TRY(launch_stream_commit(ctl, STREAM_BLOCK_CHUNKS, st));
    HIP_TRY(launch_stream_tick(ctl, st));
    return MICLOC_OK;
}

/* status[0] = chunks beamformed, [1] = frames beamformed, [2] = window-lag failures, [3] = chunks in the open reduction block
 * (synchronises the stream) */
int micloc_stream_localize_status(const void *loc_state, int *status4, void *stream)
{
    if (!loc_state || !status4) return MICLOC_ERR_INVALID;
    int ctl[16];
    HIP_TRY(hipMemcpyAsync(ctl, loc_state, sizeof(ctl), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    status4[0] = ctl[0];
    status4[1] = ctl[8];
    status4[2] = ctl[13];
    status4[3] = ctl[1];
    return MICLOC_OK;
}

// ---- stand-alone operators -------------------------------------------------------------------------------
size_t micloc_rzcc_workspace_bytes(int B, int T, int C) { return micloc_rzcc_workspace_bytes_ex(B, T, C, 1, 0); }

size_t micloc_rzcc_workspace_bytes_ex(int B, int T, int C, int robust_width, int chunk_frames)
{
    if (B < 1 || T < 1 || C < 1 || robust_width < 1) return 0;
    const size_t planar = align256((size_t)B * C * micloc_padded_T(T) * sizeof(double));
    return planar + rzcc_scratch_bytes(B * C, T, robust_width, chunk_frames);
}

int micloc_rzcc_encode_f64(const double *sig, int B, int T, int C, int robust_width, int bipolar, int8_t *spikes,
                           void *ws, size_t ws_bytes, void *stream)
{
    return micloc_rzcc_encode_ex_f64(sig, B, T, C, robust_width, bipolar, 0, spikes, ws, ws_bytes, stream);
}

int micloc_rzcc_encode_ex_f64(const double *sig, int B, int T, int C, int robust_width, int bipolar, int chunk_frames,
                              int8_t *spikes, void *ws, size_t ws_bytes, void *stream)
{
    if (!sig || !spikes || bad_batch(B) || T < 1 || C < 1 || robust_width < 1) return MICLOC_ERR_INVALID;
    // (the automatic choice does not depend on robust_width: micloc_rzcc_workspace_bytes(B, T, C) covers chunk_frames 0)
    if (bad_ws(ws, ws_bytes, micloc_rzcc_workspace_bytes_ex(B, T, C, robust_width, chunk_frames))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(spikes));
    const int Ts = micloc_padded_T(T);
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    double *planar = reinterpret_cast<double *>(base);
    void *scratch = base + align256((size_t)B * C * Ts * sizeof(double));
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(launch_pack_planar(sig, planar, B, T, C, Ts, st));
    IirCoef id{};
    id.n = 1;
    id.b[0] = 1.0;  // fma(1, x, +0) == x: the identity filter keeps the stream bit-exact
    id.a[0] = 1.0;
    HIP_TRY(launch_bandpass_rzcc(id, planar, B * C, C, T, Ts, robust_width, bipolar ? 1 : 0, nullptr, spikes, scratch,
                                 st, nullptr, 0, 0, chunk_frames));
    return MICLOC_OK;
}

size_t micloc_lfilter_workspace_bytes(int B, int T, int C)
{
    if (B < 1 || T < 1 || C < 1) return 0;
    return 2 * align256((size_t)B * C * micloc_padded_T(T) * sizeof(double));
}

int micloc_lfilter_f64(const double *b, const double *a, int n, const double *x, int B, int T, int C, double *y,
                       void *ws, size_t ws_bytes, void *stream)
{
    if (!b || !a || !x || !y || n < 1 || n > MICLOC_MAX_IIR || bad_batch(B) || T < 1 || C < 1 || a[0] == 0.0)
        return MICLOC_ERR_INVALID;
    if (bad_ws(ws, ws_bytes, micloc_lfilter_workspace_bytes(B, T, C))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(y));
    const int Ts = micloc_padded_T(T);
    const size_t planar_bytes = align256((size_t)B * C * Ts * sizeof(double));
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    double *pin = reinterpret_cast<double *>(base);
    double *pout = reinterpret_cast<double *>(base + planar_bytes);
    IirCoef co{};
    co.n = n;
    for (int i = 0; i < n; ++i) {
        co.b[i] = b[i] / a[0];
        co.a[i] = a[i] / a[0];
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(launch_pack_planar(x, pin, B, T, C, Ts, st));
    HIP_TRY(launch_bandpass_rzcc(co, pin, B * C, C, T, Ts, 1, 0, pout, nullptr, nullptr, st));
    HIP_TRY(launch_unpack_planar(pout, y, B, T, C, Ts, st));
    return MICLOC_OK;
}

// ---- fp32-MFMA variant of the beamforming tail ---------------------------------------------------------------------
int micloc_lif_beamform_f32(const micloc_plan *p, const int8_t *spikes, int B, int T, double *power, int32_t *argmax,
                            void *ws, size_t ws_bytes, void *stream)
{
    if (!p || !spikes || bad_batch(B) || T < 1 || (!power && !argmax)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (!p->d_ntab || !p->d_W) return MICLOC_ERR_NOT_SET;
    if (p->W_is_complex || p->W.CT > 4) return MICLOC_ERR_SHAPE;
    const int Gp = 16 * p->W.GT;
    if (bad_ws(ws, ws_bytes, beamform_partial_bytes(B, T, Gp))) return MICLOC_ERR_WORKSPACE;
    double *partial = reinterpret_cast<double *>(ws);
    int nch = 0;
    HIP_TRY(launch_lif_beamform_f32(p->W, p->ntab, spikes, B, T, partial, (hipStream_t)stream, &nch));
    HIP_TRY(launch_power_argmax(partial, B, T, nch, Gp, p->G_out, 0, 0, power, argmax, (hipStream_t)stream));
    return MICLOC_OK;
}

// ---- covariance-form power and membrane covariance ------------------------------------------------------------
int micloc_lif_covariance_f64(const micloc_plan *p, const int8_t *spikes, int B, int T, int t_start, double *cov,
                              double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream)
{
    if (!p || !spikes || bad_batch(B) || T < 1 || t_start < 0 || t_start >= T || (!cov && !power && !argmax))
        return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (!p->d_ntab) return MICLOC_ERR_NOT_SET;
    const bool want_power = power || argmax;
    if (want_power && (!p->d_W || p->W_is_complex)) return p->d_W ? MICLOC_ERR_SHAPE : MICLOC_ERR_NOT_SET;
    const int CT = pad_ct(p->C);
    if (CT > 8) return MICLOC_ERR_SHAPE;
    if (bad_ws(ws, ws_bytes, cov_partial_bytes(B, T, CT))) return MICLOC_ERR_WORKSPACE;
    double *partial = reinterpret_cast<double *>(ws);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(launch_lif_cov(p->ntab, spikes, B, T, p->C, CT, t_start, partial, st));
    HIP_TRY(launch_cov_power(partial, B, T, CT, p->C, T - t_start, want_power ? p->W.Wp : nullptr,
                             want_power ? 16 * p->W.GT : 16, want_power ? p->G_out : 0, cov, power, argmax, st));
    return MICLOC_OK;
}

int micloc_snn_pipeline_cov_f64(const micloc_plan *p, const double *x, int B, int T, int t_start, int8_t *spikes,
                                double *cov, double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream)
{
    if (!p || !x || bad_batch(B) || T < 1 || t_start < 0 || t_start >= T || (!cov && !power && !argmax))
        return MICLOC_ERR_INVALID;
    DeviceGuard guard(p->device);  // launches and stream belong to the plan's device, whatever the caller's current one
    if (!p->d_ntab) return MICLOC_ERR_NOT_SET;
    const bool want_power = power || argmax;
    if (want_power && (!p->d_W || p->W_is_complex)) return p->d_W ? MICLOC_ERR_SHAPE : MICLOC_ERR_NOT_SET;
    const int CT = pad_ct(p->C);
    if (CT > 8) return MICLOC_ERR_SHAPE;
    const WsLayout w = ws_layout(p, B, T);
    if (bad_ws(ws, ws_bytes, w.total)) return MICLOC_ERR_WORKSPACE;
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    double *h = reinterpret_cast<double *>(base + w.h);
    int8_t *spk = spikes ? spikes : reinterpret_cast<int8_t *>(base + w.spikes);
    const int Ts = micloc_padded_T(T);
    hipStream_t st = (hipStream_t)stream;
    // the in-phase channels are the rolled input frames: the band-pass kernel reads them from x directly
    HIP_TRY(launch_stht(p->taps, x, h, B, T, p->M, Ts, st, false));
    HIP_TRY(launch_bandpass_rzcc(p->iir, h, B * p->C, p->C, T, Ts, p->robust_width, p->bipolar, nullptr, spk,
                                 base + w.scratch, st, x, p->M, p->taps.shift, p->chunk_frames));
    double *partial = reinterpret_cast<double *>(base + w.partial);
    HIP_TRY(launch_lif_cov(p->ntab, spk, B, T, p->C, CT, t_start, partial, st));
    HIP_TRY(launch_cov_power(partial, B, T, CT, p->C, T - t_start, want_power ? p->W.Wp : nullptr,
                             want_power ? 16 * p->W.GT : 16, want_power ? p->G_out : 0, cov, power, argmax, st));
    return MICLOC_OK;
}

// ---- Gram matrix of a planar signal (complex covariance of Beamformer.design_from_template) ------------------------------
size_t micloc_planar_gram_workspace_bytes(int B, int T, int C, int t_start)
{
    if (B < 1 || T < 1 || C < 1 || t_start < 0 || t_start >= T) return 0;
    return align256(planar_gram_partial_bytes(B, T, C, t_start));
}

int micloc_planar_gram_f64(const double *planar, int B, int C, int T, int Ts, int t_start, int normalise, double *gram, void *ws,
                           size_t ws_bytes, void *stream)
{
    if (!planar || !gram || bad_batch(B) || C < 1 || C > 128 || T < 1 || t_start < 0 || t_start >= T) return MICLOC_ERR_INVALID;
    if (Ts < T) return MICLOC_ERR_SHAPE;
    if (bad_ws(ws, ws_bytes, planar_gram_partial_bytes(B, T, C, t_start))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(gram));
    HIP_TRY(launch_planar_gram(planar, B, C, T, Ts, t_start, normalise, gram, reinterpret_cast<double *>(ws), (hipStream_t)stream));
    return MICLOC_OK;
}

// ---- array-signal synthesis (noise-free part of apply_to_template) -------------------------------------------
int micloc_synth_delay_f64(const double *time, const double *sig, const double *slopes, int T, const double *delays,
                           int B, int M, double fs, double *x, void *stream)
{
    if (!time || !sig || !slopes || !delays || !x || T < 2 || bad_batch(B) || M < 1 || !(fs > 0.0)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(device_of(x));
    HIP_TRY(launch_synth(time, sig, slopes, T, delays, B, M, fs, x, (hipStream_t)stream));
    return MICLOC_OK;
}

static int synth_args_from_abi(const micloc_synth_args *a, SynthArgs *out)
{
    if (!a || !a->time || !a->sig || !a->slopes || !a->x || a->T < 2 || bad_batch(a->B) || a->K < 1 || a->M < 1 || !(a->fs > 0.0))
        return MICLOC_ERR_INVALID;
    if (a->mode != MICLOC_SYNTH_APPLY_TO_TEMPLATE && a->mode != MICLOC_SYNTH_SIGNAL_FROM_TEMPLATE) return MICLOC_ERR_INVALID;
    if (!a->delays && (!a->doa || !a->r_vec || !a->theta_vec || !(a->speed > 0.0))) return MICLOC_ERR_INVALID;
    SynthArgs k{};
    k.time = a->time;
    k.sig = a->sig;
    k.slopes = a->slopes;
    k.T = a->T;
    k.B = a->B;
    k.K = a->K;
    k.M = a->M;
    k.delays = a->delays;
    k.doa = a->doa;
    k.moving = a->moving ? 1 : 0;
    k.r_vec = a->r_vec;
    k.theta_vec = a->theta_vec;
    k.speed = a->speed;
    k.shift = a->shift;
    k.gain = a->gain;
    k.mode = a->mode;
    k.inv_step = a->fs;
    k.x = a->x;
    *out = k;
    return MICLOC_OK;
}

int micloc_synth_targets_f64(const micloc_synth_args *a, void *stream)
{
    SynthArgs k{};
    const int rc = synth_args_from_abi(a, &k);
    if (rc != MICLOC_OK) return rc;
    DeviceGuard guard(device_of(a->x));
    HIP_TRY(launch_synth_targets(k, (hipStream_t)stream));
    return MICLOC_OK;
}

size_t micloc_synth_awgn_workspace_bytes(int B, int T, int M, int K)
{
    if (B < 1 || T < 1 || M < 1 || K < 1) return 0;
    return synth_awgn_ws_bytes(B, (size_t)T * M, K, M);
}

int micloc_synth_awgn_f64(const micloc_synth_args *a, const double *snr_db, uint64_t seed, uint32_t substream, const uint32_t *epoch,
                          uint32_t first_trial, void *ws, size_t ws_bytes, void *stream)
{
    SynthArgs k{};
    const int rc = synth_args_from_abi(a, &k);
    if (rc != MICLOC_OK) return rc;
    if (!snr_db) return MICLOC_ERR_INVALID;
    if (((size_t)k.T * k.M + 1) / 2 > 0xFFFFFFFFull || (uint64_t)first_trial + (uint64_t)k.B > 0xFFFFFFFFull) return MICLOC_ERR_INVALID;
    if (bad_ws(ws, ws_bytes, synth_awgn_ws_bytes(k.B, (size_t)k.T * k.M, k.K, k.M))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(a->x));
    HIP_TRY(launch_synth_awgn(k, snr_db, seed, substream, epoch, first_trial, ws, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_delay_min_f64(const double *doa, int B, int K, int moving_T, const double *r_vec, const double *theta_vec, int M,
                         double speed, double *shift, void *stream)
{
    if (!doa || !r_vec || !theta_vec || !shift || bad_batch(B) || K < 1 || moving_T < 1 || M < 1 || !(speed > 0.0)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(device_of(shift));
    HIP_TRY(launch_delay_min(doa, B, K, moving_T, r_vec, theta_vec, M, speed, shift, (hipStream_t)stream));
    return MICLOC_OK;
}

// ---- counter-based random numbers ----------------------------------------------------------------------------
int micloc_uniform_f64(double *out, size_t n, uint64_t seed, uint32_t substream, const uint32_t *epoch, double lo, double hi,
                       void *stream)
{
    if (!out || n < 1 || (n + 1) / 2 > 0xFFFFFFFFull) return MICLOC_ERR_INVALID;  // the pair index is one counter word
    DeviceGuard guard(device_of(out));
    HIP_TRY(launch_uniform(out, n, seed, substream, epoch, lo, hi, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_counter_add_u32(uint32_t *counter, uint32_t inc, void *stream)
{
    if (!counter) return MICLOC_ERR_INVALID;
    DeviceGuard guard(device_of(counter));
    HIP_TRY(launch_counter_add(counter, inc, (hipStream_t)stream));
    return MICLOC_OK;
}

size_t micloc_awgn_workspace_bytes(int B, int T, int M)
{
    if (B < 1 || T < 1 || M < 1) return 0;
    return awgn_ws_bytes(B, (size_t)T * M);
}

int micloc_awgn_f64(double *x, int B, int T, int M, const double *snr_db, const double *sigma, uint64_t seed, uint32_t substream,
                    const uint32_t *epoch, uint32_t first_trial, void *ws, size_t ws_bytes, void *stream)
{
    if (!x || bad_batch(B) || T < 1 || M < 1 || (!snr_db && !sigma)) return MICLOC_ERR_INVALID;
    // counter words: pair index < 2^32; trial ids stay below the uniform generator's reserved word 0xFFFFFFFF
    if (((size_t)T * M + 1) / 2 > 0xFFFFFFFFull || (uint64_t)first_trial + (uint64_t)B > 0xFFFFFFFFull) return MICLOC_ERR_INVALID;
    if (!sigma && bad_ws(ws, ws_bytes, awgn_ws_bytes(B, (size_t)T * M))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(x));
    HIP_TRY(launch_awgn(x, B, (size_t)T * M, M, snr_db, sigma, seed, substream, epoch, first_trial, ws, (hipStream_t)stream));
    return MICLOC_OK;
}

// ---- per-trial DoA error and per-SNR mean absolute error ------------------------------------------------------
int micloc_doa_error_f64(const int32_t *argmax, const double *doa_list, int G, const double *doa_true, int B, int groups,
                         double *err, double *mae, void *stream)
{
    if (!argmax || !doa_list || !doa_true || G < 1 || bad_batch(B) || groups < 1 || B % groups != 0 || (!err && !mae))
        return MICLOC_ERR_INVALID;
    DeviceGuard guard(device_of(argmax));
    HIP_TRY(launch_doa_error(argmax, doa_list, G, doa_true, B, groups, err, mae, (hipStream_t)stream));
    return MICLOC_OK;
}

// ---- Xylo integer LIF (BASELINE config 4; parity unpinned, see xylo.hip) ------------------------------------
size_t micloc_xylo_workspace_bytes(int Cin, int N)
{
    if (Cin < 1 || N < 1) return 0;
    return xylo_ws_bytes(Cin, N);
}

int micloc_xylo_lif_i16(const uint8_t *spikes_in, int B, int T, int Cin, const int8_t *W_in, int N, int w_rec,
                        const uint8_t *dash_syn, const uint8_t *dash_mem, const int16_t *thr, int max_spikes,
                        uint8_t *spikes_out, int32_t *rate, void *ws, size_t ws_bytes, void *stream)
{
    if (!spikes_in || !W_in || !dash_syn || !dash_mem || !thr || bad_batch(B) || T < 1 || Cin < 1 || N < 1 || max_spikes < 1 ||
        (!spikes_out && !rate))
        return MICLOC_ERR_INVALID;
    if (Cin > 64 || (w_rec != 0 && N > 1024)) return MICLOC_ERR_SHAPE;
    for (int g = 0; g < N; ++g)
        if (thr[g] <= 0 || dash_syn[g] > 15 || dash_mem[g] > 15) return MICLOC_ERR_INVALID;
    if (bad_ws(ws, ws_bytes, xylo_ws_bytes(Cin, N))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(ws));
    HIP_TRY(launch_xylo(spikes_in, B, T, Cin, W_in, N, w_rec, dash_syn, dash_mem, thr, max_spikes, spikes_out, rate, ws,
                        (hipStream_t)stream));
    return MICLOC_OK;
}

// Two-phase form of the same network: constants uploaded once, then any number of launches that touch no host memory
// and never synchronise (capturable into a HIP graph).  `ternary_channels` > 0: the input is the encoder's int8 raster.
int micloc_xylo_upload(int Cin, const int8_t *W_in, int N, const uint8_t *dash_syn, const uint8_t *dash_mem, const int16_t *thr,
                       void *ws, size_t ws_bytes, void *stream)
{
    if (!W_in || !dash_syn || !dash_mem || !thr || Cin < 1 || N < 1) return MICLOC_ERR_INVALID;
    if (Cin > 64) return MICLOC_ERR_SHAPE;
    for (int g = 0; g < N; ++g)
        if (thr[g] <= 0 || dash_syn[g] > 15 || dash_mem[g] > 15) return MICLOC_ERR_INVALID;
    if (bad_ws(ws, ws_bytes, xylo_ws_bytes(Cin, N))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(ws));
    HIP_TRY(xylo_upload(Cin, W_in, N, dash_syn, dash_mem, thr, ws, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_xylo_lif_resident_i16(const void *spikes_in, int ternary_channels, int B, int T, int Cin, int N, int w_rec,
                                 int max_spikes, uint8_t *spikes_out, int32_t *rate, void *ws, size_t ws_bytes, void *stream)
{
    if (!spikes_in || bad_batch(B) || T < 1 || Cin < 1 || N < 1 || max_spikes < 1 || (!spikes_out && !rate)) return MICLOC_ERR_INVALID;
    if (Cin > 64 || (w_rec != 0 && N > 1024)) return MICLOC_ERR_SHAPE;
    if (ternary_channels < 0 || (ternary_channels > 0 && Cin != 2 * ternary_channels)) return MICLOC_ERR_SHAPE;
    if (bad_ws(ws, ws_bytes, xylo_ws_bytes(Cin, N))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(ws));
    HIP_TRY(launch_xylo_resident(spikes_in, ternary_channels, B, T, Cin, N, w_rec, max_spikes, spikes_out, rate, ws, (hipStream_t)stream));
    return MICLOC_OK;
}

size_t micloc_xylo_sweep_scratch_bytes(int B)
{
    if (bad_batch(B)) return 0;
    return xylo_sweep_scratch_bytes(B);
}

int micloc_xylo_lif_sweep_i16(const int8_t *raster, int ternary_channels, int B, int T, int Cin, int N, int max_spikes, int32_t *rate,
                              void *ws, size_t ws_bytes, void *scratch, size_t scratch_bytes, int workers_per_cu, void *stream)
{
    if (!raster || !rate || bad_batch(B) || T < 1 || Cin < 1 || N < 1 || max_spikes < 1 || workers_per_cu < 0 || workers_per_cu > 8) return MICLOC_ERR_INVALID;
    if (Cin > 64 || ternary_channels < 1 || Cin != 2 * ternary_channels) return MICLOC_ERR_SHAPE;
    if (bad_ws(ws, ws_bytes, xylo_ws_bytes(Cin, N))) return MICLOC_ERR_WORKSPACE;
    if (bad_ws(scratch, scratch_bytes, xylo_sweep_scratch_bytes(B))) return MICLOC_ERR_WORKSPACE;
    DeviceGuard guard(device_of(ws));
    HIP_TRY(launch_xylo_sweep(raster, ternary_channels, B, T, Cin, N, max_spikes, rate, ws, scratch, workers_per_cu, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_xylo_sweep_status(const void *scratch, int *status2, void *stream)
{
    if (!scratch || !status2) return MICLOC_ERR_INVALID;
    int ctl[2];
    HIP_TRY(hipMemcpyAsync(ctl, scratch, sizeof(ctl), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    status2[0] = ctl[0];
    status2[1] = ctl[1];
    return MICLOC_OK;
}

int micloc_design_vectors_f64(const double *cov, int n_doa, int C, int bipolar, double rel_prec, double *bf_mat, int G, int g0,
                              void *stream)
{
    if (!cov || !bf_mat || n_doa < 1 || C < 2 || G < 1 || g0 < 0 || g0 + n_doa > G || !(rel_prec > 0.0)) return MICLOC_ERR_INVALID;
    if (C > 128 || (C > 32 && (C & 1)) || (bipolar && (C & 1))) return MICLOC_ERR_SHAPE;  // (the one-sided kernel pairs all columns: even C)
    DeviceGuard guard(device_of(bf_mat));
    HIP_TRY(launch_design_vec(cov, n_doa, C, bipolar ? 1 : 0, rel_prec, bf_mat, G, g0, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_pack_events_u8(const int8_t *raster, int B, int T, int C, int band, int bands, int mode, uint8_t *out, void *stream)
{
    if (!raster || !out || bad_batch(B) || T < 1 || C < 1 || bands < 1 || band < 0 || band >= bands) return MICLOC_ERR_INVALID;
    if (mode != MICLOC_PACK_TERNARY && mode != MICLOC_PACK_UNIPOLAR && mode != MICLOC_PACK_BIPOLAR) return MICLOC_ERR_INVALID;
    DeviceGuard guard(device_of(out));
    const int stride = bands * C * (mode == MICLOC_PACK_BIPOLAR ? 2 : 1);
    HIP_TRY(launch_pack_events(raster, (size_t)B * T, C, out, stride, band * C, bands * C + band * C, mode, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_rate_from_counts_f64(const int32_t *counts, int B, int G, int bands, int T, double fs, double *rate, void *stream)
{
    if (!counts || !rate || bad_batch(B) || G < 1 || bands < 1 || T < 1 || !(fs > 0.0) || (long long)B * G > 0x7fffffffll) return MICLOC_ERR_INVALID;
    DeviceGuard guard(device_of(rate));
    HIP_TRY(launch_rate_from_counts(counts, B, G, bands, T, fs, rate, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_envelope_track_f64(const double *y, int B, int T, int G, double a_rise, double i_rise, double a_fall, double *env, int32_t *index,
                              void *stream)
{
    if (!y || !env || bad_batch(B) || T < 1 || G < 1 || (long long)B * T > 0x7fffffffll * 4) return MICLOC_ERR_INVALID;
    // 1 - 1/w and 1/w of window lengths w >= 1 (utils.py:34): finite, in [0, 1]
    if (!(a_rise >= 0.0 && a_rise <= 1.0 && i_rise >= 0.0 && i_rise <= 1.0 && a_fall >= 0.0 && a_fall <= 1.0)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(device_of(env));
    HIP_TRY(launch_envelope_track(y, MICLOC_ENV_F64, B, T, G, a_rise, i_rise, a_fall, env, index, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_envelope_track_any(const void *y, int kind, int B, int T, int G, double a_rise, double i_rise, double a_fall, double *env, int32_t *index,
                              void *stream)
{
    if (!y || !env || bad_batch(B) || T < 1 || G < 1 || (long long)B * T > 0x7fffffffll * 4) return MICLOC_ERR_INVALID;
    if (kind < MICLOC_ENV_F64 || kind > MICLOC_ENV_I64) return MICLOC_ERR_INVALID;
    if (!(a_rise >= 0.0 && a_rise <= 1.0 && i_rise >= 0.0 && i_rise <= 1.0 && a_fall >= 0.0 && a_fall <= 1.0)) return MICLOC_ERR_INVALID;
    DeviceGuard guard(device_of(env));
    HIP_TRY(launch_envelope_track(y, kind, B, T, G, a_rise, i_rise, a_fall, env, index, (hipStream_t)stream));
    return MICLOC_OK;
}

int micloc_peak_location_i32(const int32_t *rate, int B, int G, int bands, int win_size, int32_t *index, void *stream)
{
    if (!rate || !index || bad_batch(B) || G < 1 || bands < 1) return MICLOC_ERR_INVALID;
    if (win_size < 1 || win_size % 2 != 1 || win_size > G / 2 || G > 16384) return MICLOC_ERR_INVALID;  // utils.py:96-107
    DeviceGuard guard(device_of(index));
    HIP_TRY(launch_peak_location(rate, B, G, bands, win_size, index, (hipStream_t)stream));
    return MICLOC_OK;
}

}  // extern "C"
