// Covariance-form power: power[g] = mean_t (sum_c vmem[t][c] W[c][g])^2 = w_g^T (V^T V / T) w_g.
// Algebraically identical to beamforming followed by mean |y|^2 (snn_beamformer.py:368 +
// target_snn_localization.py:462), but the T x G product is never formed: per frame 2 C^2 flops instead of
// 2 C G (25x fewer at C = 14, G = 360).  Reported as a separate algorithmic variant, never as the headline number.
// The same kernel yields the membrane covariance C = V^T V / T' over the last 3/4 of the signal that
// design_from_template needs (snn_beamformer.py:176-191).
//
// lif_cov_kernel: LIF in the *non-transposed* orientation  V = N S  (A = Toeplitz nir lookup, B = int8 spikes):
// the accumulator of k-step r, lane l holds V[t = (l>>4)+4r][c = l&15], which is simultaneously the A fragment
// (A[i=c][k=t]) and the B fragment (B[k=t][j=c']) of the Gram product  R += V^T V  -- one register feeds both
// operands of v_mfma_f64_16x16x4_f64.  The LIF summation order (chronological) is unchanged, so V is bit-identical
// to the beamforming kernel's.
#include "micloc_internal.h"

namespace micloc {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int CV_THREADS = BF_WAVES * 64;

template <int CT>
__global__ __launch_bounds__(CV_THREADS) void lif_cov_kernel(const int8_t *__restrict__ spikes,
                                                              const double *__restrict__ ntab_g, int NK, int C, int T,
                                                              int t_start, double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int Cs = 16 * CT;
    constexpr int NP = CT * (CT + 1) / 2;  // upper-triangular 16x16 tiles of R
    const int tid = threadIdx.x;
    const int wv = tid >> 6;
    const int l = tid & 63;
    const int lc = l & 15;
    const int q = l >> 4;
    const int chunk = blockIdx.x;
    const int nchunks = gridDim.x;
    const int b = blockIdx.y;
    const int cs = chunk * BF_CHUNK;

    double *ntab = reinterpret_cast<double *>(smem);
    const int ntab_len = 4 * NK + 16;
    double *red = ntab + ntab_len;                       // [BF_WAVES][NP][4][64]
    int8_t *spk = reinterpret_cast<int8_t *>(red);        // aliases red (barrier in between)
    const int R = BF_CHUNK + 4 * NK - 16;

    for (int e = tid; e < ntab_len; e += CV_THREADS) ntab[e] = ntab_g[e];
    {
        const int8_t *sb = spikes + (size_t)b * T * C;
        const int tau0 = cs + 16 - 4 * NK;
        // all loads of a batch are issued (clamped addresses, hence unconditional) before the first LDS write, so
        // the workgroup pays the memory latency once per batch instead of once per element
        for (int e0 = tid; e0 < R * Cs; e0 += CV_THREADS * 8) {
            int8_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * CV_THREADS;
                const int rho = e / Cs, c = e % Cs;
                int tau = tau0 + rho;
                tau = tau < 0 ? 0 : (tau >= T ? T - 1 : tau);
                v[i] = sb[(size_t)tau * C + (c < C ? c : C - 1)];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i * CV_THREADS;
                const int rho = e / Cs, c = e % Cs;
                const int tau = tau0 + rho;
                if (e < R * Cs) spk[e] = (c < C && tau >= 0 && tau < T) ? v[i] : (int8_t)0;
            }
        }
    }
    __syncthreads();

    const int tb0 = cs + wv * BF_NT * 16;
    double4_t Racc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) Racc[p] = double4_t{0.0, 0.0, 0.0, 0.0};

    if (tb0 < T) {
        // ---- LIF for the 4 time tiles of this wave, all channel tiles ----
        double4_t V[BF_NT][CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            double4_t acc[BF_NT];
#pragma unroll
            for (int tt = 0; tt < BF_NT; ++tt) acc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
            const int8_t *sp = spk + (size_t)(tb0 - cs + q) * Cs + 16 * ct + lc;
            const double *np_ = ntab + (lc - q + 4 * NK - 16 + 15);
            double bn_n = np_[0];
            int an[BF_NT];
#pragma unroll
            for (int tt = 0; tt < BF_NT; ++tt) an[tt] = sp[(size_t)(16 * tt) * Cs];
            for (int ks = 0; ks < NK; ++ks) {
                const double bn = bn_n;
                double a[BF_NT];
#pragma unroll
                for (int tt = 0; tt < BF_NT; ++tt) a[tt] = (double)an[tt];
                if (ks + 1 < NK) {
                    bn_n = np_[-4 * (ks + 1)];
#pragma unroll
                    for (int tt = 0; tt < BF_NT; ++tt) an[tt] = sp[(size_t)(16 * tt + 4 * (ks + 1)) * Cs];
                }
#pragma unroll
                for (int tt = 0; tt < BF_NT; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bn, a[tt], acc[tt], 0, 0, 0);  // A = nir, B = spikes
            }
#pragma unroll
            for (int tt = 0; tt < BF_NT; ++tt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = tb0 + 16 * tt + q + 4 * r;  // row of this accumulator element
                    acc[tt][r] = (t < T && t >= t_start) ? acc[tt][r] : 0.0;
                }
                V[tt][ct] = acc[tt];
            }
        }
        // ---- Gram accumulation: R(ct, ct') += V_ct^T V_ct'  (upper triangle) ----
        int p = 0;
#pragma unroll
        for (int c1 = 0; c1 < CT; ++c1)
#pragma unroll
            for (int c2 = c1; c2 < CT; ++c2) {
#pragma unroll
                for (int tt = 0; tt < BF_NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Racc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[tt][c1][r], V[tt][c2][r], Racc[p], 0, 0, 0);
                ++p;
            }
    }
    __syncthreads();  // spike tile is dead: red may overwrite it
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(((size_t)wv * NP + p) * 4 + r) * 64 + l] = Racc[p][r];
    __syncthreads();
    double *pout = partial + ((size_t)b * nchunks + chunk) * (NP * 256);
    for (int e = tid; e < NP * 256; e += CV_THREADS) {
        double s = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < BF_WAVES; ++w8) s += red[(size_t)w8 * NP * 256 + e];
        pout[e] = s;
    }
}

// ---- more than 64 channels (BASELINE config 5: 64 microphones, C = 128) -------------------------------------------
// The register-resident form above needs CT membrane fragments per time tile and CT (CT + 1) / 2 Gram accumulators per
// wave: 36 x 4 doubles at CT = 8 do not fit.  Here the workgroup walks its 512-frame chunk in sub-chunks of 64 frames:
//   LIF     wave w computes the membrane fragments of time tile (w & 3), channel tiles [HALF (w >> 2), HALF (w >> 2) + HALF)
//           (the same chronological fma chain as everywhere else) and parks them in LDS in fragment order;
//   Gram    the CT (CT + 1) / 2 output tiles are dealt round-robin to the 8 waves (<= 5 each: 20 accumulator registers);
//           a wave reads both operands of every product as 8-byte fragments from LDS (conflict-free: lane-contiguous).
// 2 C^2 instead of 2 C G flops per frame: 11x fewer than the direct form at C = 128, G = 1440.
template <int CT>
__global__ __launch_bounds__(CV_THREADS) void lif_cov_wide_kernel(const int8_t *__restrict__ spikes,
                                                                   const double *__restrict__ ntab_g, int NK, int C, int T,
                                                                   int t_start, double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int Cs = 16 * CT;
    constexpr int NP = CT * (CT + 1) / 2;
    constexpr int HALF = (CT + 1) / 2;
    constexpr int SUB = 64;                       // frames per sub-chunk (4 time tiles)
    constexpr int MYP = (NP + BF_WAVES - 1) / BF_WAVES;
    const int tid = threadIdx.x;
    const int wv = tid >> 6;
    const int l = tid & 63;
    const int lc = l & 15;
    const int q = l >> 4;
    const int chunk = blockIdx.x;
    const int nchunks = gridDim.x;
    const int b = blockIdx.y;
    const int cs = chunk * BF_CHUNK;

    double *ntab = reinterpret_cast<double *>(smem);
    const int ntab_len = 4 * NK + 16;
    double *Vs = ntab + ntab_len;                                 // [4 tt][CT][4 r][64 lanes]
    int8_t *spk = reinterpret_cast<int8_t *>(Vs + 4 * CT * 256);  // [SUB + 4 NK - 16][Cs]
    const int R = SUB + 4 * NK - 16;
    for (int e = tid; e < ntab_len; e += CV_THREADS) ntab[e] = ntab_g[e];

    // the Gram tiles of this wave: p = wv, wv + 8, ...
    int c1s[MYP], c2s[MYP];
#pragma unroll
    for (int i = 0; i < MYP; ++i) {
        const int p = wv + BF_WAVES * i;
        int c1 = 0, rem = p < NP ? p : 0;
        while (rem >= CT - c1) {
            rem -= CT - c1;
            ++c1;
        }
        c1s[i] = c1;
        c2s[i] = c1 + rem;
    }
    double4_t Racc[MYP];
#pragma unroll
    for (int i = 0; i < MYP; ++i) Racc[i] = double4_t{0.0, 0.0, 0.0, 0.0};

    const int8_t *sb = spikes + (size_t)b * T * C;
    const int tt = wv & 3;
    const int ct0 = HALF * (wv >> 2);
    for (int sub = 0; sub < BF_CHUNK / SUB; ++sub) {
        const int s0 = cs + sub * SUB;
        if (s0 >= T) break;  // uniform
        __syncthreads();     // the previous sub-chunk's Gram phase has read Vs; its LIF phase has read spk
        {
            const int tau0 = s0 + 16 - 4 * NK;
            for (int e = tid; e < R * Cs; e += CV_THREADS) {
                const int rho = e / Cs, c = e % Cs;
                const int tau = tau0 + rho;
                spk[e] = (c < C && tau >= 0 && tau < T) ? sb[(size_t)tau * C + c] : (int8_t)0;
            }
        }
        __syncthreads();
        // ---- LIF: time tile tt, channel tiles ct0 .. ct0 + HALF ----
        const int tb0 = s0 + 16 * tt;
#pragma unroll
        for (int ci = 0; ci < HALF; ++ci) {
            const int ct = ct0 + ci;
            if (ct < CT) {  // uniform
                double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
                const int8_t *sp = spk + (size_t)(16 * tt + q) * Cs + 16 * ct + lc;
                const double *np_ = ntab + (lc - q + 4 * NK - 16 + 15);
                for (int ks = 0; ks < NK; ++ks)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(np_[-4 * ks], (double)sp[(size_t)(4 * ks) * Cs], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = tb0 + q + 4 * r;
                    Vs[((size_t)(tt * CT + ct) * 4 + r) * 64 + l] = (t < T && t >= t_start) ? acc[r] : 0.0;
                }
            }
        }
        __syncthreads();
        // ---- Gram: R(c1, c2) += V_c1^T V_c2 over the 4 time tiles ----
#pragma unroll
        for (int i = 0; i < MYP; ++i) {
            if (wv + BF_WAVES * i < NP) {  // uniform
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Racc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[((size_t)(t4 * CT + c1s[i]) * 4 + r) * 64 + l],
                                                                       Vs[((size_t)(t4 * CT + c2s[i]) * 4 + r) * 64 + l], Racc[i], 0, 0, 0);
            }
        }
    }
    double *pout = partial + ((size_t)b * nchunks + chunk) * (NP * 256);
#pragma unroll
    for (int i = 0; i < MYP; ++i) {
        const int p = wv + BF_WAVES * i;
        if (p < NP) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pout[((size_t)p * 4 + r) * 64 + l] = Racc[i][r];
        }
    }
}

// One workgroup per trial: R = sum over chunks (fixed order), symmetric fill, optional normalised output, and
// power[g] = w_g^T R w_g / Tn for every DoA; arg-max.
__global__ __launch_bounds__(256) void cov_power_kernel(const double *__restrict__ partial, int nchunks, int CT, int C,
                                                         int Tn, const double *__restrict__ Wp, int Gp, int G,
                                                         double *__restrict__ cov_out, double *__restrict__ power,
                                                         int32_t *__restrict__ argmax)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    double *Rs = reinterpret_cast<double *>(smem2);  // [Cp][Cp]
    __shared__ double sv[256];
    __shared__ int si[256];
    const int Cp = 16 * CT;
    const int NP = CT * (CT + 1) / 2;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const double *pb = partial + (size_t)b * nchunks * NP * 256;
    for (int e = tid; e < NP * 256; e += 256) {
        double s = 0.0;
        for (int ch = 0; ch < nchunks; ++ch) s += pb[(size_t)ch * NP * 256 + e];
        // decode: e = (p*4 + r)*64 + l ; tile p = (c1, c2), row i = (l>>4) + 4r, col j = l&15
        const int l = e & 63, r = (e >> 6) & 3, p = e >> 8;
        int c1 = 0, rem = p;
        while (rem >= CT - c1) {
            rem -= CT - c1;
            ++c1;
        }
        const int c2 = c1 + rem;
        const int i = 16 * c1 + (l >> 4) + 4 * r, j = 16 * c2 + (l & 15);
        Rs[(size_t)i * Cp + j] = s;
        if (c1 != c2) Rs[(size_t)j * Cp + i] = s;
    }
    __syncthreads();
    const double inv = 1.0 / (double)Tn;
    if (cov_out) {
        for (int e = tid; e < C * C; e += 256) {
            const int i = e / C, j = e % C;
            cov_out[(size_t)b * C * C + e] = Rs[(size_t)i * Cp + j] * inv;
        }
    }
    double best = -1.0;
    int bi = 0x7fffffff;
    if (power || argmax) {
        for (int g = tid; g < G; g += 256) {
            double p = 0.0;
            // rows of R in blocks of 16: a column value of bf_mat is loaded once per block instead of once per row
            // (the summation order per row i is unchanged: u_i = sum_j R_ij w_j with j ascending, then p += w_i u_i)
            for (int i0 = 0; i0 < C; i0 += 16) {
                double u[16];
#pragma unroll
                for (int ii = 0; ii < 16; ++ii) u[ii] = 0.0;
                for (int j = 0; j < C; ++j) {
                    const double wj = Wp[(size_t)j * Gp + g];
#pragma unroll
                    for (int ii = 0; ii < 16; ++ii) u[ii] = __builtin_fma(Rs[(size_t)(i0 + ii) * Cp + j], wj, u[ii]);
                }
#pragma unroll
                for (int ii = 0; ii < 16; ++ii)
                    if (i0 + ii < C) p = __builtin_fma(Wp[(size_t)(i0 + ii) * Gp + g], u[ii], p);
            }
            p = p * inv;
            if (power) power[(size_t)b * G + g] = p;
            if (p > best) {
                best = p;
                bi = g;
            }
        }
    }
    sv[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const double ov = sv[tid + s];
            const int oi = si[tid + s];
            if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) {
                sv[tid] = ov;
                si[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && argmax) argmax[b] = si[0] == 0x7fffffff ? 0 : si[0];
}

size_t cov_partial_bytes(int B, int T, int CT)
{
    const size_t NP = (size_t)CT * (CT + 1) / 2;
    return ((size_t)B * beamform_nchunks(T) * NP * 256 * sizeof(double) + 255) & ~(size_t)255;
}

template <int CT>
static hipError_t launch_cov_ct(const NeuronTab &nt, const int8_t *spikes, int B, int T, int C, int t_start,
                                double *partial, hipStream_t stream)
{
    constexpr int NP = CT * (CT + 1) / 2;
    size_t red = (size_t)BF_WAVES * NP * 256 * sizeof(double);
    const size_t tile = (size_t)(BF_CHUNK + 4 * nt.NK - 16) * 16 * CT;
    size_t lds = (red > tile ? red : tile) + (size_t)(4 * nt.NK + 16) * sizeof(double);
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto k = &lif_cov_kernel<CT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    dim3 grid(beamform_nchunks(T), B), block(CV_THREADS);
    hipLaunchKernelGGL(k, grid, block, lds, stream, spikes, nt.tab, nt.NK, C, T, t_start, partial);
    return hipGetLastError();
}

template <int CT>
static hipError_t launch_cov_wide(const NeuronTab &nt, const int8_t *spikes, int B, int T, int C, int t_start, double *partial,
                                  hipStream_t stream)
{
    size_t lds = (size_t)(4 * nt.NK + 16) * sizeof(double) + (size_t)4 * CT * 256 * sizeof(double) + (size_t)(64 + 4 * nt.NK - 16) * 16 * CT;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto k = &lif_cov_wide_kernel<CT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    dim3 grid(beamform_nchunks(T), B), block(CV_THREADS);
    hipLaunchKernelGGL(k, grid, block, lds, stream, spikes, nt.tab, nt.NK, C, T, t_start, partial);
    return hipGetLastError();
}

hipError_t launch_lif_cov(const NeuronTab &nt, const int8_t *spikes, int B, int T, int C, int CT, int t_start,
                          double *partial, hipStream_t stream)
{
    switch (CT) {
        case 1: return launch_cov_ct<1>(nt, spikes, B, T, C, t_start, partial, stream);
        case 2: return launch_cov_ct<2>(nt, spikes, B, T, C, t_start, partial, stream);
        case 3: return launch_cov_ct<3>(nt, spikes, B, T, C, t_start, partial, stream);
        case 4: return launch_cov_wide<4>(nt, spikes, B, T, C, t_start, partial, stream);  // (ten Gram tiles per wave: the register form's reduction buffer alone is 160 KB)
        case 5: return launch_cov_wide<5>(nt, spikes, B, T, C, t_start, partial, stream);
        case 6: return launch_cov_wide<6>(nt, spikes, B, T, C, t_start, partial, stream);
        case 7: return launch_cov_wide<7>(nt, spikes, B, T, C, t_start, partial, stream);
        case 8: return launch_cov_wide<8>(nt, spikes, B, T, C, t_start, partial, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_cov_power(const double *partial, int B, int T, int CT, int C, int Tn, const double *Wp, int Gp, int G,
                            double *cov_out, double *power, int32_t *argmax, hipStream_t stream)
{
    const size_t lds = (size_t)(16 * CT) * (16 * CT) * sizeof(double);
    if (lds > 48 * 1024) {
        // (the kernel also has 3 KB of static LDS: ask for what is needed, not for the whole 160 KB)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&cov_power_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // do not leave the error behind for the next launch's check
            return e;
        }
    }
    hipLaunchKernelGGL(cov_power_kernel, dim3(B), dim3(256), lds, stream, partial, beamform_nchunks(T), CT, C, Tn, Wp, Gp,
                       G, cov_out, power, argmax);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Gram matrix of a planar signal:  R[b] = sum_{t >= t_start} x[b][:, t] x[b][:, t]^T  (/ (T - t_start)),  x planar [B][C][Ts].
//
// Beamformer.design_from_template (micloc/beamformer.py:142-150) needs the complex covariance conj(h)^T h / T' of the STHT
// output h = re + j im over the stable part of each delayed template; with the planar rows [re_0.. re_{M-1}, im_0.. im_{M-1}]
// it is a fold of the REAL 2M x 2M Gram matrix:  cov = (R_rr + R_ii) + j (R_ri - R_ir).
//
// One wave per (trial, time chunk, 16 x 16 tile pair).  In v_mfma_f64_16x16x4_f64 the A fragment of lane l is A[i = l & 15][k = l >> 4]
// and the B fragment B[k = l >> 4][j = l & 15]: for a Gram product both are x[channel = l & 15][time k], so a diagonal tile needs
// ONE register per step.  Which four time steps share an instruction only fixes the (deterministic) order of the sum, so lane
// group q = l >> 4 streams its own contiguous quarter of the chunk.  Chunk sums are combined in a fixed order by the second kernel.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PG_CHUNK = 2048;  // time steps per wave: 4 lane groups x 512

__global__ __launch_bounds__(64) void planar_gram_kernel(const double *__restrict__ x, int C, int CT, int T, int Ts, int t_start,
                                                          double *__restrict__ partial)
{
    const int l = threadIdx.x, lc = l & 15, q = l >> 4;
    const int chunk = blockIdx.x, pair = blockIdx.y, b = blockIdx.z;
    // pair -> (ti <= tj), row-major over the upper triangle
    int ti = 0, rem = pair;
    while (rem >= CT - ti) {
        rem -= CT - ti;
        ++ti;
    }
    const int tj = ti + rem;
    const int ca = 16 * ti + lc, cb = 16 * tj + lc;
    const double *xa = x + ((size_t)b * C + (ca < C ? ca : 0)) * Ts;
    const double *xb = x + ((size_t)b * C + (cb < C ? cb : 0)) * Ts;
    const bool va = ca < C, vb = cb < C;
    const int t0 = t_start + chunk * PG_CHUNK + q * (PG_CHUNK / 4);
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int s = 0; s < PG_CHUNK / 4; ++s) {
        const int t = t0 + s;
        const bool in = t < T;
        const double a = (in && va) ? xa[t] : 0.0;
        const double bb = ti == tj ? a : ((in && vb) ? xb[t] : 0.0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
    }
    // accumulator layout of the fp64 16 x 16 tile: lane l, register r -> R[i = (l >> 4) + 4 r][j = l & 15]
    double *out = partial + (((size_t)b * gridDim.x + chunk) * gridDim.y + pair) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(q + 4 * r) * 16 + lc] = acc[r];
}

__global__ __launch_bounds__(256) void planar_gram_reduce_kernel(const double *__restrict__ partial, int nchunks, int C, int CT, double inv_n,
                                                                  double *__restrict__ gram)
{
    const int b = blockIdx.y;
    const int np = CT * (CT + 1) / 2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < C * C; e += gridDim.x * 256) {
        const int i = e / C, j = e % C;
        const int lo = i < j ? i : j, hi = i < j ? j : i;  // symmetric: read the upper triangle
        const int ti = lo >> 4, tj = hi >> 4;
        const int pair = ti * CT - ti * (ti - 1) / 2 + (tj - ti);
        double acc = 0.0;
        for (int c = 0; c < nchunks; ++c) acc += partial[(((size_t)b * nchunks + c) * np + pair) * 256 + (lo & 15) * 16 + (hi & 15)];
        gram[((size_t)b * C + i) * C + j] = acc * inv_n;
    }
}

size_t planar_gram_partial_bytes(int B, int T, int C, int t_start)
{
    const int CT = (C + 15) / 16;
    const int nchunks = (T - t_start + PG_CHUNK - 1) / PG_CHUNK;
    return (size_t)B * nchunks * (CT * (CT + 1) / 2) * 256 * sizeof(double);
}

hipError_t launch_planar_gram(const double *x, int B, int C, int T, int Ts, int t_start, int normalise, double *gram, double *partial,
                              hipStream_t stream)
{
    const int CT = (C + 15) / 16;
    const int nchunks = (T - t_start + PG_CHUNK - 1) / PG_CHUNK;
    const int np = CT * (CT + 1) / 2;
    hipLaunchKernelGGL(planar_gram_kernel, dim3(nchunks, np, B), dim3(64), 0, stream, x, C, CT, T, Ts, t_start, partial);
    hipLaunchKernelGGL(planar_gram_reduce_kernel, dim3((C * C + 255) / 256, B), dim3(256), 0, stream, partial, nchunks, C, CT,
                       normalise ? 1.0 / (double)(T - t_start) : 1.0, gram);
    return hipGetLastError();
}

}  // namespace micloc
