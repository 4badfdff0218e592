// Per-trial results of the Monte-Carlo loop on the device.
// Reference: paper_plots/target_snn_localization.py:464-467 (DoA estimate from the arg-max, pi-periodic error) and :520
// (mean absolute error per SNR).  Keeps the timed sweep free of framework element-wise launches: one tiny kernel after
// power_argmax_kernel instead of index / sin / abs / asin / mean.
#include "micloc_internal.h"

namespace micloc {

// One workgroup per SNR group (contiguous runs of `per_group` trials).  The per-group mean is a fixed-order two-level
// sum (strided per-thread partial sums, then a binary tree in LDS): deterministic, independent of the launch.
__global__ __launch_bounds__(256) void doa_error_kernel(const int32_t *__restrict__ argmax, const double *__restrict__ doa_list,
                                                         int G, const double *__restrict__ doa_true, int per_group,
                                                         double *__restrict__ err, double *__restrict__ mae)
{
    __shared__ double red[256];
    const int s = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < per_group; i += 256) {
        const int b = s * per_group + i;
        int a = argmax[b];
        a = a < 0 ? 0 : (a >= G ? G - 1 : a);
        const double e = asin(fabs(sin(doa_list[a] - doa_true[b])));
        if (err) err[b] = e;
        acc += e;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0 && mae) mae[s] = red[0] / (double)per_group;
}

hipError_t launch_doa_error(const int32_t *argmax, const double *doa_list, int G, const double *doa_true, int B, int groups,
                            double *err, double *mae, hipStream_t stream)
{
    hipLaunchKernelGGL(doa_error_kernel, dim3(groups), dim3(256), 0, stream, argmax, doa_list, G, doa_true, B / groups, err, mae);
    return hipGetLastError();
}

// find_peak_location (micloc/utils.py:84-121) of the per-DoA spike counts of the Xylo sweep
// (paper_plots/target_xylo_localization.py:594-604): power[g] = mean over bands of rate[f * G + g], smoothed with a
// box-car of `win` samples by a FULL, non-circular convolution, arg-max, minus win // 2, modulo G.  The reference
// normalises power by its maximum and by T / fs first -- positive factors that do not move the arg-max -- and sums in
// floating point; here the window sums are exact integers (first maximum wins, as np.argmax), so the two can differ
// only where the float sums of the reference break an exact integer tie by rounding.
__global__ __launch_bounds__(256) void peak_location_kernel(const int32_t *__restrict__ rate, int G, int F, int win,
                                                             int32_t *__restrict__ index)
{
    extern __shared__ long long pw[];  // [G]
    __shared__ long long bestv[256];
    __shared__ int besti[256];
    const int b = blockIdx.x;
    const int32_t *r = rate + (size_t)b * F * G;
    for (int g = threadIdx.x; g < G; g += 256) {
        long long s = 0;
        for (int f = 0; f < F; ++f) s += r[(size_t)f * G + g];
        pw[g] = s;
    }
    __syncthreads();
    long long bv = -1;
    int bi = 0x7fffffff;
    for (int n = threadIdx.x; n < G + win - 1; n += 256) {
        long long s = 0;
        for (int k = 0; k < win; ++k) {
            const int g = n - k;
            if (g >= 0 && g < G) s += pw[g];
        }
        if (s > bv) {  // n ascending per thread: the first maximum of this thread's subsequence
            bv = s;
            bi = n;
        }
    }
    bestv[threadIdx.x] = bv;
    besti[threadIdx.x] = bi;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            const long long ov = bestv[threadIdx.x + h];
            const int oi = besti[threadIdx.x + h];
            if (ov > bestv[threadIdx.x] || (ov == bestv[threadIdx.x] && oi < besti[threadIdx.x])) {
                bestv[threadIdx.x] = ov;
                besti[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int idx = besti[0] - win / 2;
        idx %= G;
        index[b] = idx < 0 ? idx + G : idx;
    }
}

hipError_t launch_peak_location(const int32_t *rate, int B, int G, int F, int win, int32_t *index, hipStream_t stream)
{
    const size_t dyn = (size_t)G * sizeof(long long);
    if (dyn > 48 * 1024) {
        // beyond the default dynamic-LDS limit (the API accepts G <= 16384 = 128 KB + 3 KB static of the CU's 160 KB)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(peak_location_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(peak_location_kernel, dim3(B), dim3(256), dyn, stream, rate, G, F, win, index);
    return hipGetLastError();
}

// ---- moving-target tracking: Envelope.evolve + arg-max over the DoA grid per time step ---------------------------------------------
// micloc/utils.py:36-81 (the rise / fall envelope of every beamformer output column) and
// paper_plots/target_snn_localization.py:599-622 (`doa_index = np.argmax(sig_bf_env, axis=1)`): G independent serial chains over T.
//   state_0 = |y_0|;   rise_t = |y_t| >= state_{t-1};
//   state_t = (1 - 1/w) * state_{t-1} + 1/w * |y_t| * rise_t,   w = rise_t ? int(fs rise_time) : int(fs fall_time);   env[t] = state_t
// in NumPy's order of operations: two roundings of products, one of the sum, nothing fused (`__dmul_rn` / `__dadd_rn`); with rise_t = 0
// the second term is +0 and the sum is the first product.  a_* = 1 - 1/w and i_rise = 1/w_rise are computed on the host by NumPy itself.
// A lane = one DoA column of one trial (a wave's row segment: 512 contiguous bytes per step); per step the chain is two multiplies side by
// side, one add, one select.
// Round 6, second form: the chain is latency, so the rows must be waiting in LDS when the chain gets to them.  A workgroup = 64 DoA columns
// of one trial = FOUR waves: wave 0 walks the recurrence, waves 1-3 are loaders that take the 32-row tiles in turn -- a loader requests its
// tile (one 8-byte load per lane and row: 32 in flight), lets two iterations of the chain pass, writes the tile into a four-slot LDS ring and
// requests its next one; one barrier per 32 rows, every wave runs the same number of iterations.  Tile k: loads issued in iteration k, LDS
// write in iteration k + 2, walked in iteration k + 3; slot k % 4 is rewritten in iteration k + 6.  (The first form -- one wave per workgroup,
// the next 32 rows prefetched into registers while the current 32 are walked -- ran at 73 ns per step on the script's 240 000 x 449 array:
// a row load's latency is longer than 32 steps of the chain.)
constexpr int ENV_TILE = 32;
constexpr int ENV_SLOTS = 4;
constexpr int ENV_LOADERS = 3;

// What the envelope is taken of (np.abs of the caller's array): the real SNN beamformer output, the complex Beamformer's output
// (ref:paper_plots/target_localization.py:597-600: |z| = hypot(re, im), NumPy's npy_cabs), or an integer spike raster
// (ref:paper_plots/target_xylo_localization.py:757-768: `sig_bf = spikes_out`).  Integers and real doubles are exact; hypot is the device
// library's (< 1 ulp; the host libm's last bit is not a contract either).
template <int KIND>
struct EnvIn;
template <>
struct EnvIn<MICLOC_ENV_F64> {
    typedef double T;
    static __device__ __forceinline__ double mag(double v) { return fabs(v); }
};
template <>
struct EnvIn<MICLOC_ENV_C128> {
    typedef double2 T;
    static __device__ __forceinline__ double mag(double2 v) { return hypot(v.x, v.y); }
};
template <>
struct EnvIn<MICLOC_ENV_U8> {
    typedef uint8_t T;
    static __device__ __forceinline__ double mag(uint8_t v) { return (double)v; }
};
template <>
struct EnvIn<MICLOC_ENV_I32> {
    typedef int32_t T;
    static __device__ __forceinline__ double mag(int32_t v) { return fabs((double)v); }
};
template <>
struct EnvIn<MICLOC_ENV_I64> {
    typedef long long T;
    static __device__ __forceinline__ double mag(long long v) { return fabs((double)v); }
};

template <int KIND>
__global__ __launch_bounds__(64 * (1 + ENV_LOADERS)) void envelope_kernel(const void *__restrict__ y_, int T, int G, double a_rise, double i_rise,
                                                                          double a_fall, double *__restrict__ env)
{
    typedef typename EnvIn<KIND>::T In;
    const In *__restrict__ y = static_cast<const In *>(y_);
    extern __shared__ __attribute__((aligned(16))) double env_ring[];  // [ENV_SLOTS][ENV_TILE][64]: the MAGNITUDES of a tile's rows
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 64 + lane;
    const bool live = g < G;
    const In *p = y + (size_t)blockIdx.y * T * G + (live ? g : 0);
    double *o = env + (size_t)blockIdx.y * T * G + (live ? g : 0);
    const int nt = (T - 1 + ENV_TILE - 1) / ENV_TILE;  // tiles of rows 1 .. T - 1 (row 0 is the initial state)
    double state = 0.0;
    if (wave == 0) {
        state = live ? EnvIn<KIND>::mag(p[0]) : 0.0;
        if (live) o[0] = state;
    }
    // a barrier WITHOUT __syncthreads()'s memory fence: the fence would wait for every outstanding global load and store (vmcnt(0)) -- the
    // loaders' rows in flight, the chain's 32 stores -- in every iteration; only the LDS traffic has to be complete at the barrier.
    // The two roles run SEPARATE loops with the same trip count (nt + 3 barriers each): their register sets never meet in a phi.
    auto tile_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (wave == 0) {
        for (int it = 0; it < nt + 3; ++it) {
            const int k = it - 3;
            if (k >= 0) {
                double v[ENV_TILE];
                const double *slot = env_ring + ((size_t)(k % ENV_SLOTS) * ENV_TILE) * 64 + lane;
#pragma unroll
                for (int j = 0; j < ENV_TILE; ++j) v[j] = slot[j * 64];
                const int t0 = 1 + k * ENV_TILE;
                const int nrow = T - t0 < ENV_TILE ? T - t0 : ENV_TILE;  // (uniform; only a trial's last tile is ragged)
                // a lone wave issues one instruction of any kind per ~4.4 cycles, so a step costs its instruction COUNT: the lanes beyond
                // the grid are masked once per tile (not a branch around every store), the row's address is a scalar base that steps by
                // one row + this lane's 32-bit column offset, and the full tile carries no per-step row test
                if (live) {
                    const char *rowbase = reinterpret_cast<const char *>(env + ((size_t)blockIdx.y * T + t0) * G);  // (uniform)
                    const size_t rowbytes = (size_t)G * sizeof(double);
                    const unsigned voff = (unsigned)g * (unsigned)sizeof(double);
                    auto step = [&](int j) {
                        const double m = v[j];  // (a magnitude: the loaders took |.| of what they read)
                        const double up = __dadd_rn(__dmul_rn(a_rise, state), __dmul_rn(i_rise, m));
                        const double down = __dmul_rn(a_fall, state);
                        state = (m >= state) ? up : down;
                        *reinterpret_cast<double *>(const_cast<char *>(rowbase + (size_t)j * rowbytes) + voff) = state;
                    };
                    if (nrow == ENV_TILE) {
#pragma unroll
                        for (int j = 0; j < ENV_TILE; ++j) step(j);  // (compile-time indices: the tile stays in registers)
                    } else {
#pragma unroll
                        for (int j = 0; j < ENV_TILE; ++j)
                            if (j < nrow) step(j);
                    }
                }
            }
            tile_barrier();
        }
    } else {
        // this loader's tiles are k % 3 == wave - 1; an iteration is in exactly ONE of three phases (exclusive branches: the rows requested
        // two iterations ago stay in the registers they were loaded into -- no copies, no wait before they are written)
        In v[ENV_TILE];
#pragma unroll
        for (int j = 0; j < ENV_TILE; ++j) v[j] = In{};
        for (int it = 0; it < nt + 3; ++it) {
            const int ph = (it + ENV_LOADERS - (wave - 1)) % ENV_LOADERS;
            if (ph == 0) {
                if (it < nt) {  // request tile `it`: UNCONDITIONAL loads (a predicated load is a branch around it and a wait for everything
                    // in flight in front of the zero that replaces it): rows beyond the recording read row T - 1 again, lanes beyond the
                    // grid read column 0 (`p`) -- valid memory, values nobody uses
                    const int t0 = 1 + it * ENV_TILE;
#pragma unroll
                    for (int j = 0; j < ENV_TILE; ++j) {
                        const int row = t0 + j < T ? t0 + j : T - 1;  // (uniform)
                        v[j] = p[(size_t)row * G];
                    }
                }
            } else if (ph == 2) {
                const int ks = it - 2;  // the tile requested two iterations ago
                if (ks >= 0 && ks < nt) {
                    double *slot = env_ring + ((size_t)(ks % ENV_SLOTS) * ENV_TILE) * 64 + lane;
#pragma unroll
                    for (int j = 0; j < ENV_TILE; ++j) slot[j * 64] = EnvIn<KIND>::mag(v[j]);
                }
            }
            tile_barrier();
        }
    }
}

// index[row] = np.argmax(env[row, :]): the FIRST maximum (one wave per row; env >= 0, NaN never wins)
__global__ __launch_bounds__(256) void rows_argmax_kernel(const double *__restrict__ env, size_t rows, int G, int32_t *__restrict__ index)
{
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int l = threadIdx.x & 63;
    const double *r = env + row * G;
    double best = -1.0;
    int bi = 0x7fffffff;
    for (int g = l; g < G; g += 64) {  // ascending per lane: its first maximum
        const double v = r[g];
        if (v > best) {
            best = v;
            bi = g;
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const double ov = __shfl_xor(best, s, 64);
        const int oi = __shfl_xor(bi, s, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (l == 0) index[row] = bi == 0x7fffffff ? 0 : bi;
}

template <int KIND>
static hipError_t launch_envelope_kind(const void *y, int B, int T, int G, double a_rise, double i_rise, double a_fall, double *env, hipStream_t stream)
{
    const size_t ring = (size_t)ENV_SLOTS * ENV_TILE * 64 * sizeof(double);  // 64 KB
    auto k = &envelope_kernel<KIND>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((G + 63) / 64, B), dim3(64 * (1 + ENV_LOADERS)), ring, stream, y, T, G, a_rise, i_rise, a_fall, env);
    return hipGetLastError();
}

hipError_t launch_envelope_track(const void *y, int kind, int B, int T, int G, double a_rise, double i_rise, double a_fall, double *env,
                                 int32_t *index, hipStream_t stream)
{
    hipError_t e;
    switch (kind) {
        case MICLOC_ENV_F64: e = launch_envelope_kind<MICLOC_ENV_F64>(y, B, T, G, a_rise, i_rise, a_fall, env, stream); break;
        case MICLOC_ENV_C128: e = launch_envelope_kind<MICLOC_ENV_C128>(y, B, T, G, a_rise, i_rise, a_fall, env, stream); break;
        case MICLOC_ENV_U8: e = launch_envelope_kind<MICLOC_ENV_U8>(y, B, T, G, a_rise, i_rise, a_fall, env, stream); break;
        case MICLOC_ENV_I32: e = launch_envelope_kind<MICLOC_ENV_I32>(y, B, T, G, a_rise, i_rise, a_fall, env, stream); break;
        case MICLOC_ENV_I64: e = launch_envelope_kind<MICLOC_ENV_I64>(y, B, T, G, a_rise, i_rise, a_fall, env, stream); break;
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess || !index) return e;
    const size_t rows = (size_t)B * T;
    hipLaunchKernelGGL(rows_argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, env, rows, G, index);
    return hipGetLastError();
}

// ---- Demo.spike_encoding's channel bookkeeping (xylo_snn_localization.py:339-354) and extract_rate (:379-398) -----------------
// One band's int8 raster [rows][C] -> its channel block of the assembled tensor [rows][stride]:
//   mode 0  the ternary value itself          out[row][pos_off + c] = s            (bands concatenated on the channel axis)
//   mode 1  unipolar events                   out[row][pos_off + c] = s > 0
//   mode 2  bipolar events, the +/- split     out[row][pos_off + c] = s > 0,  out[row][neg_off + c] = s < 0
// (the reference: np.hstack over bands, astype(int64), then [spikes > 0, spikes < 0] side by side)
__global__ __launch_bounds__(256) void pack_events_kernel(const int8_t *__restrict__ raster, size_t rows, int C, uint8_t *__restrict__ out,
                                                           int stride, int pos_off, int neg_off, int mode)
{
    const size_t n = rows * (size_t)C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const size_t row = e / C;
        const int c = (int)(e - row * C);
        const int8_t s = raster[e];
        uint8_t *o = out + row * stride;
        if (mode == 0) {
            o[pos_off + c] = (uint8_t)s;
        } else {
            o[pos_off + c] = (uint8_t)(s > 0);
            if (mode == 2) o[neg_off + c] = (uint8_t)(s < 0);
        }
    }
}

hipError_t launch_pack_events(const int8_t *raster, size_t rows, int C, uint8_t *out, int stride, int pos_off, int neg_off, int mode,
                              hipStream_t stream)
{
    const size_t n = rows * (size_t)C;
    const unsigned grid = (unsigned)((n + 255) / 256 > 65535 * 4 ? 65535 * 4 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_events_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, raster, rows, C, out, stride, pos_off, neg_off, mode);
    return hipGetLastError();
}

// rate[b][g] = mean over the F bands of (counts[b][f G + g] / T) * fs -- np.mean(spikes_out, 0) * fs, then .reshape(-1, G).mean(0),
// in that order of operations (the counts are exact integers, so the time mean is one division)
__global__ __launch_bounds__(256) void rate_from_counts_kernel(const int32_t *__restrict__ counts, int B, int G, int F, double T, double fs,
                                                                double *__restrict__ rate)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G;
    double s = 0.0;
    for (int f = 0; f < F; ++f) s += (double)counts[((size_t)b * F + f) * G + g] / T * fs;
    rate[i] = s / (double)F;
}

hipError_t launch_rate_from_counts(const int32_t *counts, int B, int G, int F, int T, double fs, double *rate, hipStream_t stream)
{
    hipLaunchKernelGGL(rate_from_counts_kernel, dim3((B * G + 255) / 256), dim3(256), 0, stream, counts, B, G, F, (double)T, fs, rate);
    return hipGetLastError();
}

}  // namespace micloc
