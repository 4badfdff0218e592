// fp32-MFMA variant of the LIF + beamforming + power kernel (a separately reported VARIANT, never the default).
//
// BASELINE.json's north star allows the beamforming stage to reproduce the reference "within 1e-5 float32"; everything
// up to the spikes must stay fp64 (fp32 before the encoder flips spikes), but from the (exact, ternary)
// spikes onwards fp32 is sufficient for the power pattern: v_mfma_f32_16x16x4_f32 issues in 32 cycles against 64 for
// the fp64 form, i.e. 157 TFLOP/s peak.  Same structure as beamform.hip (Toeplitz LIF whose accumulators are the
// A fragments of the next product), with the fp32 C/D layout of gfx950: lane l, register r -> row 4*(l>>4)+r,
// column l&15 (the fp64 form has row (l>>4)+4r).  The channel order inside a k-step is therefore c = 4q + r
// (q = l>>4) instead of 4r + q; W is fetched accordingly.  Squares are accumulated in fp32 per lane over 16 values and
// combined in fp64.  Parity: power within ~1e-6 relative of the fp64 path (tests: 1e-5), arg-max agreement measured and
// reported by bench.py, never assumed.
#include "micloc_internal.h"

namespace micloc {

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t32 __attribute__((ext_vector_type(2)));

constexpr int F32_THREADS = BF_WAVES * 64;

__device__ __forceinline__ double row_sum4_d(double x)
{
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    uint2_t32 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    uint2_t32 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(s);
    hi = __double2hiint(s);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

template <int CT>
__global__ __launch_bounds__(F32_THREADS) void beamform_f32_kernel(const int8_t *__restrict__ spikes,
                                                                    const double *__restrict__ ntab_g, int NK,
                                                                    const double *__restrict__ Wp, int GT, int C, int T,
                                                                    double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int Cs = 16 * CT;
    constexpr int KS = 4 * CT;
    const int Gp = 16 * GT;
    const int tid = threadIdx.x;
    const int wv = tid >> 6;
    const int l = tid & 63;
    const int lc = l & 15;
    const int q = l >> 4;
    const int chunk = blockIdx.x;
    const int nchunks = gridDim.x;
    const int b = blockIdx.y;
    const int cs = chunk * BF_CHUNK;

    // [ W fp32: C rows + one zero row ][ nir table fp32 ][ union{ red fp64 [8][Gp] , spike tile } ]
    float *Wl = reinterpret_cast<float *>(smem);
    float *ntab = Wl + (size_t)(C + 1) * Gp;
    const int ntab_len = 4 * NK + 16;
    double *red = reinterpret_cast<double *>(ntab + ((ntab_len + 3) & ~3));
    int8_t *spk = reinterpret_cast<int8_t *>(red);
    const int R = BF_CHUNK + 4 * NK - 16;

    for (int e = tid; e < C * Gp; e += F32_THREADS) Wl[e] = (float)Wp[e];
    for (int e = tid; e < Gp; e += F32_THREADS) Wl[(size_t)C * Gp + e] = 0.0f;
    for (int e = tid; e < ntab_len; e += F32_THREADS) ntab[e] = (float)ntab_g[e];
    {
        const int8_t *sb = spikes + (size_t)b * T * C;
        const int tau0 = cs + 16 - 4 * NK;
        for (int e0 = tid; e0 < R * Cs; e0 += F32_THREADS * 8) {
            int8_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * F32_THREADS;
                const int rho = e / Cs, c = e % Cs;
                int tau = tau0 + rho;
                tau = tau < 0 ? 0 : (tau >= T ? T - 1 : tau);
                v[i] = sb[(size_t)tau * C + (c < C ? c : C - 1)];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * F32_THREADS;
                const int rho = e / Cs, c = e % Cs;
                const int tau = tau0 + rho;
                if (e < R * Cs) spk[e] = (c < C && tau >= 0 && tau < T) ? v[i] : (int8_t)0;
            }
        }
    }
    __syncthreads();

    const int tb0 = cs + wv * BF_NT * 16;
    float4_t Vf[BF_NT][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        float4_t acc[BF_NT];
#pragma unroll
        for (int tt = 0; tt < BF_NT; ++tt) acc[tt] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (tb0 < T) {
            const int8_t *sp = spk + (size_t)(tb0 - cs + q) * Cs + 16 * ct + lc;
            const float *np_ = ntab + (lc - q + 4 * NK - 16 + 15);
            float bn_n = np_[0];
            int an[BF_NT];
#pragma unroll
            for (int tt = 0; tt < BF_NT; ++tt) an[tt] = sp[(size_t)(16 * tt) * Cs];
            for (int ks = 0; ks < NK; ++ks) {
                const float bn = bn_n;
                float a[BF_NT];
#pragma unroll
                for (int tt = 0; tt < BF_NT; ++tt) a[tt] = (float)an[tt];
                if (ks + 1 < NK) {
                    bn_n = np_[-4 * (ks + 1)];
#pragma unroll
                    for (int tt = 0; tt < BF_NT; ++tt) an[tt] = sp[(size_t)(16 * tt + 4 * (ks + 1)) * Cs];
                }
#pragma unroll
                for (int tt = 0; tt < BF_NT; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], bn, acc[tt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tt = 0; tt < BF_NT; ++tt) {
            const bool tvalid = (tb0 + 16 * tt + lc) < T;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tt][r] = tvalid ? acc[tt][r] : 0.f;
            Vf[tt][ct] = acc[tt];  // lane: t = l&15, channel 16ct + 4q + r
        }
    }
    __syncthreads();  // spike tile is dead: red may overwrite it

    if (tb0 >= T) {
        for (int g = l; g < Gp; g += 64) red[(size_t)wv * Gp + g] = 0.0;
    } else {
        // k-step (ct, r) contracts the channels 16ct + 4q + r, q = 0..3
        int woff[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int row = 16 * (k >> 2) + 4 * q + (k & 3);
            woff[k] = (row < C ? row : C) * Gp;
        }
        for (int gt = 0; gt < GT; ++gt) {
            const float *wp = Wl + 16 * gt + lc;
            float Wf[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) Wf[k] = wp[woff[k]];
            float4_t acc[BF_NT];
#pragma unroll
            for (int tt = 0; tt < BF_NT; ++tt) acc[tt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int tt = 0; tt < BF_NT; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vf[tt][k >> 2][k & 3], Wf[k], acc[tt], 0, 0, 0);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int tt = 0; tt < BF_NT; tt += 2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s0 = __builtin_fmaf(acc[tt][r], acc[tt][r], s0);
                    s1 = __builtin_fmaf(acc[tt + 1][r], acc[tt + 1][r], s1);
                }
            const double sq = row_sum4_d((double)s0 + (double)s1);
            if (l < 16) red[(size_t)wv * Gp + 16 * gt + l] = sq;
        }
    }
    __syncthreads();
    double *pout = partial + ((size_t)b * nchunks + chunk) * Gp;
    for (int g = tid; g < Gp; g += F32_THREADS) {
        double s = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < BF_WAVES; ++w8) s += red[(size_t)w8 * Gp + g];
        pout[g] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf_mat-stationary form (same ownership as beamform_ws_kernel in beamform.hip) for up to 16 channels:
// 256-frame workgroups, spikes staged as fp32, membrane fragments parked in LDS in fragment order, a wave keeps the
// bf_mat fragments of its own DoA tiles in registers and walks over all 16 time tiles.  Per (time tile, DoA tile): 4
// MFMAs + 2 v_pk_fma_f32 (running per-lane sum of squares in fp32 over the 64 values of a chunk, combined in fp64).
// ---------------------------------------------------------------------------------------------------------------
typedef float float2_t __attribute__((ext_vector_type(2)));
constexpr int WF_NT = 2;                       // 16-frame tiles per wave
constexpr int WF_CH = BF_WAVES * WF_NT * 16;   // 256 frames per workgroup
constexpr int WF_TILES = WF_CH / 16;

template <int NG>
__device__ __forceinline__ void wsf_stage2(const float *Vl, const double *__restrict__ Wp, int Gp, int wv, int l, int ntile,
                                           double *__restrict__ pout)
{
    if constexpr (NG > 0) {
        const int lc = l & 15;
        const int q = l >> 4;
        float Wf[NG][4];  // k-step r contracts the channels 4q + r
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const double *wp = Wp + 16 * (wv + BF_WAVES * j) + lc;
#pragma unroll
            for (int k = 0; k < 4; ++k) Wf[j][k] = (float)wp[(size_t)(4 * q + k) * Gp];
        }
        float2_t sq[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) sq[j] = float2_t{0.f, 0.f};
        auto ldv = [&](int tile, float (&V)[4]) {
            const float *p = Vl + (size_t)(tile < WF_TILES ? tile : WF_TILES - 1) * 256 + l;
#pragma unroll
            for (int k = 0; k < 4; ++k) V[k] = p[64 * k];
        };
        auto mm_sq = [&](const float (&V)[4]) {
            float4_t acc[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < NG; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[k], Wf[j][k], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const float2_t lo = {acc[j][0], acc[j][1]}, hi = {acc[j][2], acc[j][3]};
                sq[j] = __builtin_elementwise_fma(lo, lo, sq[j]);
                sq[j] = __builtin_elementwise_fma(hi, hi, sq[j]);
            }
        };
        float VA[4], VB[4];
        if (ntile > 0) ldv(0, VA);
        int t = 0;
        for (; t + 1 < ntile; t += 2) {
            ldv(t + 1, VB);
            mm_sq(VA);
            ldv(t + 2, VA);
            mm_sq(VB);
        }
        if (t < ntile) mm_sq(VA);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const double s = row_sum4_d((double)sq[j][0] + (double)sq[j][1]);
            if (l < 16) pout[16 * (wv + BF_WAVES * j) + l] = s;
        }
    }
}

template <int NGW>
__global__ __launch_bounds__(F32_THREADS, 6) void beamform_wsf32_kernel(const int8_t *__restrict__ spikes,
                                                                         const double *__restrict__ ntab_g, int NK,
                                                                         const double *__restrict__ Wp, int GT, int C, int T,
                                                                         double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Gp = 16 * GT;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int lc = l & 15;
    const int q = l >> 4;
    const int chunk = blockIdx.x;
    const int nchunks = gridDim.x;
    const int b = blockIdx.y;
    const int cs = chunk * WF_CH;

    // [ union{ spike tile as fp32 [R][16] , V fragments [16 tiles][4 k-steps][64 lanes] } ][ nir table fp32 ]
    const int R = WF_CH + 4 * NK - 16;
    float *S = reinterpret_cast<float *>(smem);
    float *Vl = S;
    float *ntab = S + (R * 16 > WF_TILES * 256 ? R * 16 : WF_TILES * 256);
    const int ntab_len = 4 * NK + 16;
    for (int e = tid; e < ntab_len; e += F32_THREADS) ntab[e] = (float)ntab_g[e];
    {
        const int8_t *sb = spikes + (size_t)b * T * C;
        const int tau0 = cs + 16 - 4 * NK;
        const int c = tid & 15;
        const int cc = c < C ? c : C - 1;
        constexpr int RP = F32_THREADS / 16;
        for (int r0 = tid >> 4; r0 < R; r0 += RP * 5) {
            int8_t v[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                int tau = tau0 + r0 + RP * i;
                tau = tau < 0 ? 0 : (tau >= T ? T - 1 : tau);
                v[i] = sb[(size_t)tau * C + cc];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int rho = r0 + RP * i;
                const int tau = tau0 + rho;
                if (rho < R) S[rho * 16 + c] = (c < C && tau >= 0 && tau < T) ? (float)v[i] : 0.f;
            }
        }
    }
    __syncthreads();

    const int tb0 = cs + wv * WF_NT * 16;
    const bool active = tb0 < T;
    float4_t vacc[WF_NT];
#pragma unroll
    for (int tt = 0; tt < WF_NT; ++tt) vacc[tt] = float4_t{0.f, 0.f, 0.f, 0.f};
    if (active) {
        const float *sp = S + (size_t)(tb0 - cs + q) * 16 + lc;
        const float *np_ = ntab + (lc - q + 4 * NK - 16 + 15);
        for (int ks = 0; ks < NK; ++ks) {
            const float bn = np_[-4 * ks];
            float a[WF_NT];
#pragma unroll
            for (int tt = 0; tt < WF_NT; ++tt) a[tt] = sp[(16 * tt + 4 * ks) * 16];
#pragma unroll
            for (int tt = 0; tt < WF_NT; ++tt) vacc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], bn, vacc[tt], 0, 0, 0);
        }
    }
    __syncthreads();  // the spike tile is dead: V fragments may overwrite it
    if (active) {
#pragma unroll
        for (int tt = 0; tt < WF_NT; ++tt) {
            const bool tvalid = (tb0 + 16 * tt + lc) < T;
            float *vp = Vl + (size_t)(wv * WF_NT + tt) * 256 + l;
#pragma unroll
            for (int r = 0; r < 4; ++r) vp[64 * r] = tvalid ? vacc[tt][r] : 0.f;  // lane: t = lc, channel 4q + r
        }
    }
    __syncthreads();

    int ntile = (T - cs + 15) >> 4;
    ntile = ntile > WF_TILES ? WF_TILES : ntile;
    double *pout = partial + ((size_t)b * nchunks + chunk) * Gp;
    if (wv + BF_WAVES * (NGW - 1) < GT)
        wsf_stage2<NGW>(Vl, Wp, Gp, wv, l, ntile, pout);
    else
        wsf_stage2<NGW - 1>(Vl, Wp, Gp, wv, l, ntile, pout);
}

static size_t wsf_lds_bytes(const NeuronTab &nt)
{
    const size_t tile = (size_t)(WF_CH + 4 * nt.NK - 16) * 16, vfrag = (size_t)WF_TILES * 256;
    return ((tile > vfrag ? tile : vfrag) + (size_t)(4 * nt.NK + 16)) * sizeof(float);
}

template <int NGW>
static hipError_t launch_wsf_n(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T, double *partial,
                               hipStream_t stream)
{
    auto k = &beamform_wsf32_kernel<NGW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    dim3 grid((T + WF_CH - 1) / WF_CH, B), block(F32_THREADS);
    hipLaunchKernelGGL(k, grid, block, wsf_lds_bytes(nt), stream, spikes, nt.tab, nt.NK, W.Wp, W.GT, W.C, T, partial);
    return hipGetLastError();
}

template <int CT>
static hipError_t launch_f32_ct(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T,
                                double *partial, hipStream_t stream)
{
    const int Gp = 16 * W.GT;
    size_t uni = (size_t)BF_WAVES * Gp * sizeof(double);
    const size_t tile = (size_t)(BF_CHUNK + 4 * nt.NK - 16) * 16 * CT;
    uni = uni > tile ? uni : tile;
    size_t lds = (size_t)(W.C + 1) * Gp * sizeof(float) + (size_t)((4 * nt.NK + 16 + 3) & ~3) * sizeof(float) + uni;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto k = &beamform_f32_kernel<CT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    dim3 grid(beamform_nchunks(T), B), block(F32_THREADS);
    hipLaunchKernelGGL(k, grid, block, lds, stream, spikes, nt.tab, nt.NK, W.Wp, W.GT, W.C, T, partial);
    return hipGetLastError();
}

hipError_t launch_lif_beamform_f32(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T,
                                   double *partial, hipStream_t stream, int *nchunks)
{
    if (W.CT == 1 && W.GT <= 4 * BF_WAVES && wsf_lds_bytes(nt) <= 160 * 1024) {
        *nchunks = (T + WF_CH - 1) / WF_CH;
        switch ((W.GT + BF_WAVES - 1) / BF_WAVES) {
            case 1: return launch_wsf_n<1>(W, nt, spikes, B, T, partial, stream);
            case 2: return launch_wsf_n<2>(W, nt, spikes, B, T, partial, stream);
            case 3: return launch_wsf_n<3>(W, nt, spikes, B, T, partial, stream);
            default: return launch_wsf_n<4>(W, nt, spikes, B, T, partial, stream);
        }
    }
    *nchunks = beamform_nchunks(T);
    switch (W.CT) {
        case 1: return launch_f32_ct<1>(W, nt, spikes, B, T, partial, stream);
        case 2: return launch_f32_ct<2>(W, nt, spikes, B, T, partial, stream);
        case 3: return launch_f32_ct<3>(W, nt, spikes, B, T, partial, stream);
        case 4: return launch_f32_ct<4>(W, nt, spikes, B, T, partial, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace micloc
