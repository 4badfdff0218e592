// Band-pass IIR + robust zero-crossing (RZCC) spike encoder for gfx950.
// Reference: micloc/snn_beamformer.py:330-338 (lfilter(b, a, .) then spk_encoder.evolve) and
// micloc/spike_encoder.py:115-137 (cumsum -> scipy.signal.find_peaks(distance=robust_width)).
//
// The encoder contract is bit-exact, and both the DF2T recurrence and np.cumsum are strictly
// sequential in time, so the time axis is NOT parallelised: one lane owns one (trial, channel) stream
// and walks it with the exact operation order of oracle/micloc_oracle.c; parallelism comes from the
// B x 2M independent streams.  The kernel is therefore bound by the instruction-issue latency of one
// wave per 64 streams, and everything is arranged to keep that wave's loop short and stall-free:
//
//   * workgroup = 2 waves.  Wave 1 is a LOADER: it streams 32-step x 64-stream input tiles from the
//     planar [stream][Ts] layout (256-byte coalesced row segments) into a double-buffered, transposed
//     LDS tile and (for the band-pass-only variants) stores the filtered tile back.  Wave 0 is the
//     COMPUTE wave: it reads its sample from LDS, so no global load ever sits in front of its global
//     stores in the in-order vmcnt queue (measured: a conditional global store in the serial loop cost
//     a vmcnt(0) drain per chunk and 2.5x the loop time).
//   * detection is branch-free: per step two fp64 compares maintain (dir, left) = direction and index
//     of the last strict change of the cumulative sum; a maximum completes when c falls after a rise
//     (scipy _local_maxima_1d: plateau -> midpoint, edges never peaks), minima mirror it.  Candidates
//     are written unconditionally into a per-lane LDS ring; the write index only advances on an event.
//   * scipy's _select_by_peak_distance (greedy by descending priority, later peak wins ties) only
//     couples peaks closer than `distance`, so the candidate list splits into independent clusters at
//     every same-polarity gap >= distance.  After each 32-step tile the compute wave walks the NEW
//     events (not the time steps), closes finished clusters, resolves them in LDS and scatters the kept
//     peaks as +1 / -1 bytes into the zero-initialised [B][T][C] spike tensor.
//   * a cluster that outgrows the LDS ring (pathological inputs: long runs of close peaks) flags its
//     stream; flagged streams are redone by a slow list-based kernel with unbounded capacity, so the
//     result is exact for every input.
#include "micloc_internal.h"

namespace micloc {

constexpr int RZ_RING = 64;    // candidate ring entries per stream (power of two)
constexpr int RZ_ROW = 65;     // padded row of the transposed input tile (doubles)

template <int N>
struct Iir {
    double z[N > 1 ? N - 1 : 1];

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < (N > 1 ? N - 1 : 1); ++i) z[i] = 0.0;
    }

    __device__ __forceinline__ double step(const IirCoef &coef, double xin)
    {
        double y;
        if (N == 1) {
            y = __builtin_fma(coef.b[0], xin, 0.0);
        } else {
            y = __builtin_fma(coef.b[0], xin, z[0]);
#pragma unroll
            for (int i = 0; i < N - 2; ++i)
                z[i] = __builtin_fma(-coef.a[i + 1], y, __builtin_fma(coef.b[i + 1], xin, z[i + 1]));
            z[N - 2] = __builtin_fma(-coef.a[N - 1], y, coef.b[N - 1] * xin);
        }
        return y;
    }
};

// Greedy min-distance selection inside one cluster.  Entries live at list indices s, s+stride, ... < e;
// word = (position << 1) | polarity, complemented once decided.  `at(i)` maps a list index to storage.
template <typename WordAt, typename ValAt>
__device__ __forceinline__ void resolve_cluster(int s, int e, int stride, int w, int8_t mark, int8_t *sp, int C,
                                                WordAt word_at, ValAt val_at)
{
    int remaining = (e - s + stride - 1) / stride;
    while (remaining > 0) {
        int best = -1;
        double bv = 0.0;
        for (int k = s; k < e; k += stride) {
            const int wk = *word_at(k);
            if (wk < 0) continue;
            const double vk = *val_at(k);
            if (best < 0 || vk >= bv) {  // >= : equal priority -> the later peak wins (stable sort order)
                best = k;
                bv = vk;
            }
        }
        const int wb = *word_at(best);
        const int pb = wb >> 1;
        sp[(size_t)pb * C] = mark;
        *word_at(best) = ~wb;
        --remaining;
        for (int k = best - stride; k >= s; k -= stride) {
            const int wk = *word_at(k);
            const int pk = (wk < 0 ? ~wk : wk) >> 1;
            if (pb - pk >= w) break;
            if (wk >= 0) {
                *word_at(k) = ~wk;
                --remaining;
            }
        }
        for (int k = best + stride; k < e; k += stride) {
            const int wk = *word_at(k);
            const int pk = (wk < 0 ? ~wk : wk) >> 1;
            if (pk - pb >= w) break;
            if (wk >= 0) {
                *word_at(k) = ~wk;
                --remaining;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Fast path: a 4-stage wave pipeline over 16-step tiles (one barrier per tile).
//   wave L (loader)  tile k+1 -> LDS (loads issued one tile earlier, into registers), stores filtered tiles
//   wave F (filter)  tile k  : x -> band-pass -> cumulative sum, in place in LDS
//   wave D (detect)  tile k-1: local maxima / minima of the cumulative sum -> candidate ring
//   wave S (select)  tile k-2: walks the new candidates, closes clusters, min-distance greedy, scatters spikes
// Each stage is a different wave of the workgroup, i.e. a different SIMD of the CU, so the serial chain of
// one stream costs max(stage) instructions per time step instead of their sum.
// ---------------------------------------------------------------------------------------------------
constexpr int RZ_MT = 16;

template <int N, bool WANT_PRE, bool WANT_SPIKES>
__global__ __launch_bounds__(256) void bandpass_rzcc_fast_kernel(const double *__restrict__ h,
                                                                  double *__restrict__ pre,
                                                                  int8_t *__restrict__ spikes,
                                                                  int *__restrict__ flag_count,
                                                                  int *__restrict__ flag_list, IirCoef coef,
                                                                  int nlanes, int C, int T, int Ts, int w, int bipolar,
                                                                  const double *__restrict__ xin, int M, int shift)
{
    __shared__ __attribute__((aligned(16))) double X[3][RZ_MT][RZ_ROW];
    __shared__ double Y[WANT_PRE ? 2 : 1][WANT_PRE ? RZ_MT : 1][WANT_PRE ? RZ_ROW : 1];
    __shared__ double ringV[WANT_SPIKES ? RZ_RING : 1][64];
    __shared__ int ringP[WANT_SPIKES ? RZ_RING : 1][64];
    __shared__ int nPub[64];

    const int wave = threadIdx.x >> 6;  // 0: filter, 1: loader, 2: detect, 3: select
    const int lane = threadIdx.x & 63;
    const int base = blockIdx.x * 64;
    const int NM = (T + RZ_MT - 1) / RZ_MT;
    const int NSTEP = WANT_SPIKES ? NM + 2 : NM + 1;
    const int lane_g = base + lane;
    const bool active = lane_g < nlanes;

    if (wave == 1) {
        // ------------------------------------ loader ---------------------------------------------------
        const int tl = lane & 15;  // time offset inside the tile
        const int sq = lane >> 4;  // stream slot 0..3 of each group of four
        const bool full_block = base + 64 <= nlanes;
        double v[16];
        // Per-stream source.  Quadrature channels (and everything when xin == nullptr) come from the planar STHT
        // buffer h; with xin != nullptr the in-phase channels c < M are read straight from the input frames,
        // x[b][(t - L/2) mod T][c]  (np.roll, snn_beamformer.py:325), so the STHT kernel need not write them.
        const double *pb[16];
        bool rolled[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int g = base + 4 * j + sq;
            g = g < nlanes ? g : nlanes - 1;
            const int bb = g / C, cc = g - bb * C;
            rolled[j] = xin != nullptr && cc < M;
            pb[j] = rolled[j] ? xin + (size_t)bb * T * M + cc : h + (size_t)g * Ts;
        }
        const int sh = shift % T;
        auto issue_loads = [&](int m) {
            const int t = m * RZ_MT + tl;
            const int tc = t < Ts ? t : Ts - 1;  // clamp: samples past T are never used
            int tr = (t < T ? t : T - 1) - sh;
            tr = tr < 0 ? tr + T : tr;
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = rolled[j] ? pb[j][(size_t)tr * M] : pb[j][tc];
        };
        auto write_tile = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 16; ++j) X[buf][tl][4 * j + sq] = v[j];
        };
        auto store_tile = [&](int m) {
            const int t = m * RZ_MT + tl;
            const int yb = m & 1;
            if (full_block && (m + 1) * RZ_MT <= Ts) {
                double *p = pre + (size_t)(base + sq) * Ts + t;
#pragma unroll
                for (int j = 0; j < 16; ++j) p[(size_t)(4 * j) * Ts] = Y[WANT_PRE ? yb : 0][WANT_PRE ? tl : 0][WANT_PRE ? 4 * j + sq : 0];
            } else {
#pragma unroll 4
                for (int j = 0; j < 16; ++j) {
                    const int g = base + 4 * j + sq;
                    if (g < nlanes && t < Ts) pre[(size_t)g * Ts + t] = Y[WANT_PRE ? yb : 0][WANT_PRE ? tl : 0][WANT_PRE ? 4 * j + sq : 0];
                }
            }
        };
        issue_loads(0);
        write_tile(0);
        if (NM > 1) issue_loads(1);
        __syncthreads();
        for (int k = 0; k < NSTEP; ++k) {
            if (k + 1 < NM) write_tile((k + 1) % 3);
            if (k + 2 < NM) issue_loads(k + 2);
            if (WANT_PRE && k >= 1 && k <= NM) store_tile(k - 1);
            __syncthreads();
        }
        return;
    }

    if (wave == 0) {
        // ------------------------------------ filter ---------------------------------------------------
        Iir<N> iir;
        iir.init();
        double cs = 0.0;
        __syncthreads();
        for (int k = 0; k < NSTEP; ++k) {
            if (k < NM) {
                const int buf = k % 3;
                const int steps = (T - k * RZ_MT) < RZ_MT ? (T - k * RZ_MT) : RZ_MT;
                auto one = [&](int j) {
                    const double y = iir.step(coef, X[buf][j][lane]);
                    if (WANT_PRE) Y[WANT_PRE ? (k & 1) : 0][WANT_PRE ? j : 0][WANT_PRE ? lane : 0] = y;
                    if (WANT_SPIKES) {
                        cs = cs + y;
                        X[buf][j][lane] = cs;
                    }
                };
                if (steps == RZ_MT) {
#pragma unroll
                    for (int j = 0; j < RZ_MT; ++j) one(j);
                } else {
                    for (int j = 0; j < steps; ++j) one(j);
                }
            }
            __syncthreads();
        }
        return;
    }

    if (!WANT_SPIKES) return;  // (band-pass-only launches have just the two waves above)

    if (wave == 2) {
        // ------------------------------------ detect ---------------------------------------------------
        // dir  = direction of the last strict change of c (+1 rise, -1 fall, 0 none yet); left = its time index.
        // A maximum completes when c falls after a rise: plateau [left, t-1] -> position (left+t-1)>>1, priority =
        // plateau value = prev.  Minima mirror this (priority -prev).  Maxima and minima alternate strictly, so both
        // share one candidate list, tagged by the low bit of the stored word.
        double prev = __builtin_nan("");  // comparisons with NaN are false: no event at t = 0
        int left = 0, dir = 0, n = 0;
        nPub[lane] = 0;
        __syncthreads();
        for (int k = 0; k < NSTEP; ++k) {
            if (k >= 1 && k <= NM) {
                const int m = k - 1;
                const int buf = m % 3;
                const int tbase = m * RZ_MT;
                const int steps = (T - tbase) < RZ_MT ? (T - tbase) : RZ_MT;
                auto one = [&](int j) {
                    const double c = X[buf][j][lane];
                    const int t = tbase + j;
                    const bool rise = c > prev;
                    const bool fall = c < prev;
                    const bool ev = (fall && dir > 0) || (bipolar && rise && dir < 0);
                    const int slot = n & (RZ_RING - 1);  // unconditional store; consumed only if n advances
                    ringP[slot][lane] = ((left + t - 1) & ~1) | (rise ? 1 : 0);
                    ringV[slot][lane] = rise ? -prev : prev;
                    n += ev ? 1 : 0;
                    left = (rise || fall) ? t : left;
                    dir = rise ? 1 : (fall ? -1 : dir);
                    prev = c;
                };
                if (steps == RZ_MT) {
#pragma unroll
                    for (int j = 0; j < RZ_MT; ++j) one(j);
                } else {
                    for (int j = 0; j < steps; ++j) one(j);
                }
                nPub[lane] = n;
            }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------- select -------------------------------------------------------
    int n_done = 0;
    int s0 = -1, s1 = -1;  // first list index of the open cluster per polarity (-1: none)
    int l0 = 0, l1 = 0;    // position of the last candidate per polarity
    bool dead = !active;   // ring overflow (or lane out of range): stop selecting; redone by the fallback kernel
    const int stride = bipolar ? 2 : 1;
    const int b = active ? lane_g / C : 0;
    const int ch = active ? lane_g - b * C : 0;
    int8_t *sp = spikes + (size_t)b * T * C + ch;
    auto word_at = [&](int i) { return &ringP[i & (RZ_RING - 1)][lane]; };
    auto val_at = [&](int i) { return &ringV[i & (RZ_RING - 1)][lane]; };
    auto close_cluster = [&](int s, int e, int pol, int lastpos) {
        const int8_t mark = pol ? -1 : 1;
        if (e - s <= stride)
            sp[(size_t)lastpos * C] = mark;
        else
            resolve_cluster(s, e, stride, w, mark, sp, C, word_at, val_at);
    };
    __syncthreads();
    for (int k = 0; k < NSTEP; ++k) {
        if (k >= 2) {
            const int n = nPub[lane];  // candidates published by the detect wave before the last barrier
            for (int i = n_done; __any(!dead && i < n); ++i) {
                if (!dead && i < n) {
                    const int word = *word_at(i);
                    const int pol = word & 1;
                    const int pos = word >> 1;
                    const int sp_i = pol ? s1 : s0;
                    const int lp = pol ? l1 : l0;
                    int snew = sp_i;
                    if (sp_i >= 0 && pos - lp >= w) {
                        close_cluster(sp_i, i, pol, lp);
                        snew = i;
                    }
                    if (sp_i < 0) snew = i;
                    if (pol) {
                        s1 = snew;
                        l1 = pos;
                    } else {
                        s0 = snew;
                        l0 = pos;
                    }
                }
            }
            n_done = n;
            // the detect wave runs up to two tiles ahead of what has been selected: everything still open must
            // survive 2 * RZ_MT more appends
            const int oldest = (s0 >= 0 && (s1 < 0 || s0 < s1)) ? s0 : (s1 >= 0 ? s1 : n);
            if (!dead && n - oldest >= RZ_RING - 2 * RZ_MT) dead = true;
        }
        __syncthreads();
    }
    if (active) {
        if (!dead) {
            if (s0 >= 0) close_cluster(s0, n_done, 0, l0);
            if (s1 >= 0) close_cluster(s1, n_done, 1, l1);
        } else {
            const int kk = atomicAdd(flag_count, 1);
            flag_list[kk] = lane_g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Fallback for flagged streams: unbounded candidate lists in global scratch (slot-major), selection
// after the stream.  Slow (conditional global stores in the serial loop) but exact for any input.
// ---------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void rzcc_fallback_kernel(const double *__restrict__ h, int8_t *__restrict__ spikes,
                                                            const int *__restrict__ flag_count,
                                                            const int *__restrict__ flag_list, int *__restrict__ plist,
                                                            double *__restrict__ vlist, IirCoef coef, int nlanes,
                                                            int C, int T, int Ts, int w, int bipolar,
                                                            const double *__restrict__ xin, int M, int shift)
{
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= *flag_count) return;
    const int lane_g = flag_list[idx];
    const int b = lane_g / C;
    const int ch = lane_g - b * C;
    const bool rolled = xin != nullptr && ch < M;
    const double *src = rolled ? xin + (size_t)b * T * M + ch : h + (size_t)lane_g * Ts;
    const int sh = shift % T;
    const size_t NL = (size_t)nlanes;
    int *P = plist + idx;
    double *V = vlist + idx;

    Iir<N> iir;
    iir.init();
    double c = 0.0, prev = __builtin_nan("");
    int left = 0, dir = 0, n = 0;
    for (int t = 0; t < T; ++t) {
        int tr = t - sh;
        tr = tr < 0 ? tr + T : tr;
        const double y = iir.step(coef, rolled ? src[(size_t)tr * M] : src[t]);
        c = c + y;
        const bool rise = c > prev;
        const bool fall = c < prev;
        if ((fall && dir > 0) || (bipolar && rise && dir < 0)) {
            P[(size_t)n * NL] = ((left + t - 1) & ~1) | (rise ? 1 : 0);
            V[(size_t)n * NL] = rise ? -prev : prev;
            ++n;  // n <= T - 1 < capacity T
        }
        left = (rise || fall) ? t : left;
        dir = rise ? 1 : (fall ? -1 : dir);
        prev = c;
    }

    int8_t *sp = spikes + (size_t)b * T * C + ch;
    const int stride = bipolar ? 2 : 1;
    auto word_at = [&](int i) { return P + (size_t)i * NL; };
    auto val_at = [&](int i) { return V + (size_t)i * NL; };
    const int first_pol = n > 0 ? (P[0] & 1) : 0;
    for (int pol = 0; pol < (bipolar ? 2 : 1); ++pol) {
        const int8_t mark = pol ? -1 : 1;
        const int i0 = bipolar ? (first_pol == pol ? 0 : 1) : 0;
        if (i0 >= n) continue;
        int s = i0;
        int plast = *word_at(i0) >> 1;
        for (int i = i0 + stride;; i += stride) {
            const bool has = i < n;
            int pi = 0;
            bool closes = true;
            if (has) {
                pi = *word_at(i) >> 1;
                closes = (pi - plast) >= w;
            }
            if (closes) {
                const int e = has ? i : n;
                if (e - s <= stride)
                    sp[(size_t)plast * C] = mark;
                else
                    resolve_cluster(s, e, stride, w, mark, sp, C, word_at, val_at);
                s = i;
            }
            if (!has) break;
            plast = pi;
        }
    }
}

// Zero fill as an ordinary kernel.  hipMemsetAsync is avoided on purpose: captured into a HIP graph (ROCm 7.x) the
// memset NODE was observed not to be ordered before the kernel nodes that follow it -- the fallback kernel then read a
// stale flagged-stream counter and faulted on the third replay -- whereas kernel -> kernel edges are honoured.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4 *__restrict__ p16, size_t n16, unsigned char *__restrict__ tail,
                                                         int ntail)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p16[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

static hipError_t zero_fill(void *ptr, size_t bytes, hipStream_t stream)
{
    // torch / hipMalloc buffers are at least 16-byte aligned; the scatter target [B][T][C] int8 may have any size
    const size_t n16 = bytes / 16;
    const int ntail = (int)(bytes - n16 * 16);
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<uint4 *>(ptr), n16,
                       reinterpret_cast<unsigned char *>(ptr) + n16 * 16, ntail);
    return hipGetLastError();
}

size_t rzcc_scratch_bytes(int nlanes, int T)
{
    size_t bytes = 256;                                         // flagged-stream counter
    bytes += ((size_t)nlanes * sizeof(int) + 255) & ~(size_t)255;  // flagged-stream list
    bytes += (size_t)T * nlanes * sizeof(double);               // fallback priority lists
    bytes += (size_t)T * nlanes * sizeof(int);                  // fallback position lists
    return (bytes + 255) & ~(size_t)255;
}

template <int N>
static void launch_rz(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts, int w, int bipolar,
                      double *pre, int8_t *spikes, int *flag_count, int *flag_list, int *plist, double *vlist,
                      const double *xin, int M, int shift, hipStream_t stream)
{
    dim3 grid((nlanes + 63) / 64), block(spikes ? 256 : 128);
    if (pre && spikes)
        hipLaunchKernelGGL((bandpass_rzcc_fast_kernel<N, true, true>), grid, block, 0, stream, h, pre, spikes,
                           flag_count, flag_list, coef, nlanes, C, T, Ts, w, bipolar, xin, M, shift);
    else if (spikes)
        hipLaunchKernelGGL((bandpass_rzcc_fast_kernel<N, false, true>), grid, block, 0, stream, h, pre, spikes,
                           flag_count, flag_list, coef, nlanes, C, T, Ts, w, bipolar, xin, M, shift);
    else
        hipLaunchKernelGGL((bandpass_rzcc_fast_kernel<N, true, false>), grid, block, 0, stream, h, pre, spikes,
                           flag_count, flag_list, coef, nlanes, C, T, Ts, w, bipolar, xin, M, shift);
    if (spikes)
        hipLaunchKernelGGL((rzcc_fallback_kernel<N>), grid, dim3(64), 0, stream, h, spikes, flag_count, flag_list,
                           plist, vlist, coef, nlanes, C, T, Ts, w, bipolar, xin, M, shift);
}

hipError_t launch_bandpass_rzcc(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts,
                                int robust_width, int bipolar, double *pre, int8_t *spikes, void *scratch,
                                hipStream_t stream, const double *xin, int M, int shift)
{
    if (!pre && !spikes) return hipErrorInvalidValue;
    int *flag_count = nullptr, *flag_list = nullptr, *plist = nullptr;
    double *vlist = nullptr;
    if (spikes) {
        unsigned char *base = reinterpret_cast<unsigned char *>(scratch);
        flag_count = reinterpret_cast<int *>(base);
        flag_list = reinterpret_cast<int *>(base + 256);
        vlist = reinterpret_cast<double *>(base + 256 + (((size_t)nlanes * sizeof(int) + 255) & ~(size_t)255));
        plist = reinterpret_cast<int *>(vlist + (size_t)T * nlanes);
        hipError_t e = zero_fill(flag_count, 256, stream);
        if (e != hipSuccess) return e;
        e = zero_fill(spikes, (size_t)nlanes * T, stream);
        if (e != hipSuccess) return e;
    }
#define RZ_CASE(NN)                                                                                       \
    case NN:                                                                                              \
        launch_rz<NN>(coef, h, nlanes, C, T, Ts, robust_width, bipolar, pre, spikes, flag_count, flag_list, \
                      plist, vlist, xin, M, shift, stream);                                               \
        break;
    switch (coef.n) {
        RZ_CASE(1)
        RZ_CASE(2)
        RZ_CASE(3)
        RZ_CASE(4)
        RZ_CASE(5)
        RZ_CASE(6)
        RZ_CASE(7)
        RZ_CASE(8)
        RZ_CASE(9)
        default:
            return hipErrorInvalidValue;
    }
#undef RZ_CASE
    return hipGetLastError();
}

// ---- row-major [B][T][C] <-> planar [B][C][Ts] (LDS tile transpose, both sides coalesced) ----------
__global__ __launch_bounds__(256) void pack_planar_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                           int T, int C, int Ts, int to_planar)
{
    __shared__ double tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const double *rm = src + (to_planar ? (size_t)b * T * C : (size_t)b * C * Ts);
    double *out = dst + (to_planar ? (size_t)b * C * Ts : (size_t)b * T * C);
    if (to_planar) {
        for (int r = ty; r < 32; r += 8) {
            const int t = t0 + r, c = c0 + tx;
            tile[r][tx] = (t < T && c < C) ? rm[(size_t)t * C + c] : 0.0;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, t = t0 + tx;
            if (c < C && t < Ts) out[(size_t)c * Ts + t] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, t = t0 + tx;
            tile[r][tx] = (c < C && t < T) ? rm[(size_t)c * Ts + t] : 0.0;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int t = t0 + r, c = c0 + tx;
            if (t < T && c < C) out[(size_t)t * C + c] = tile[tx][r];
        }
    }
}

hipError_t launch_pack_planar(const double *src, double *dst, int B, int T, int C, int Ts, hipStream_t stream)
{
    dim3 grid((Ts + 31) / 32, (C + 31) / 32, B), block(256);
    hipLaunchKernelGGL(pack_planar_kernel, grid, block, 0, stream, src, dst, T, C, Ts, 1);
    return hipGetLastError();
}

hipError_t launch_unpack_planar(const double *src, double *dst, int B, int T, int C, int Ts, hipStream_t stream)
{
    dim3 grid((Ts + 31) / 32, (C + 31) / 32, B), block(256);
    hipLaunchKernelGGL(pack_planar_kernel, grid, block, 0, stream, src, dst, T, C, Ts, 0);
    return hipGetLastError();
}

}  // namespace micloc
