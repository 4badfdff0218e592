// Band-pass IIR + robust zero-crossing (RZCC) spike encoder for gfx950.
// Reference: micloc/snn_beamformer.py:330-338 (lfilter(b, a, .) then spk_encoder.evolve) and
// micloc/spike_encoder.py:115-137 (cumsum -> scipy.signal.find_peaks(distance=robust_width)).
//
// The encoder contract is bit-exact, and both the DF2T recurrence and np.cumsum are strictly
// sequential in time, so the time axis is NOT parallelised: one lane owns one (trial, channel) stream
// and walks it with the exact operation order of oracle/micloc_oracle.c; parallelism comes from the
// B x 2M independent streams.  The kernel is bound by the instruction-issue latency of single waves
// (a wave issues at most one instruction of any kind every ~4 cycles), so the work of a time step is
// spread over a 5-stage wave pipeline (see bandpass_rzcc_fast_kernel) and every stage is written for a
// minimal instruction count:
//
//   * LOADER wave: 16-step x 64-stream input tiles from the planar [stream][Ts] layout (in-phase channels
//     straight from the rolled input frames) into a transposed LDS tile; two register sets keep the loads
//     of two tiles in flight.  No global load ever sits in front of a global store in another wave's
//     in-order vmcnt queue (measured: a conditional global store in the serial loop cost a vmcnt(0) drain
//     per chunk and 2.5x the loop time).
//   * FILTER wave: DF2T recurrence + cumulative sum, in place in LDS.
//   * DETECT wave: branch-free local-extremum detector.  Per step two fp64 compares give wave-level
//     lane masks (rise / fall); the direction of the last strict change lives in lane masks too and is
//     updated on the SALU; a maximum completes when c falls after a rise (scipy _local_maxima_1d:
//     plateau -> midpoint, edges never peaks), minima mirror it.  Candidates are written unconditionally
//     into a per-lane LDS ring; the write index only advances on an event (add-with-carry of the mask).
//   * SELECT waves (one per polarity): scipy's _select_by_peak_distance (greedy by descending priority,
//     later peak wins ties) only couples peaks closer than `distance`, so the candidate list splits into
//     independent clusters at every same-polarity gap >= distance.  After each tile the wave walks the NEW
//     events (not the time steps), closes finished clusters, resolves them in LDS and scatters the kept
//     peaks as +1 / -1 bytes into the zero-initialised [B][T][C] spike tensor.
//   * a cluster that outgrows the LDS ring (pathological inputs: long runs of close peaks) flags its
//     stream; flagged streams are redone by a slow list-based kernel with unbounded capacity, so the
//     result is exact for every input.
#include "micloc_internal.h"

#include <type_traits>

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
// word >> 1 = position (the low bit is free for the caller), complemented once decided; priority = sgn * value.
// `at(i)` maps a list index to storage.
template <typename WordAt, typename ValAt>
__device__ __forceinline__ void resolve_cluster(int s, int e, int stride, int w, int8_t mark, int8_t *sp, int C,
                                                WordAt word_at, ValAt val_at, double sgn)
{
    int remaining = (e - s + stride - 1) / stride;
    while (remaining > 0) {
        int best = -1;
        double bv = 0.0;
        for (int k = s; k < e; k += stride) {
            const int wk = *word_at(k);
            if (wk < 0) continue;
            const double vk = *val_at(k) * sgn;
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
// Fast path: a wave pipeline over 16-step tiles (one barrier per tile).
//   wave L (loader)  tile k+1 -> LDS (loads issued two tiles earlier, into registers), stores filtered tiles
//   wave F (filter)  tile k  : x -> band-pass -> cumulative sum, in place in LDS
//   wave D (detect)  tile k-1: local maxima / minima of the cumulative sum -> candidate ring
//   waves S0, S1 (select) tile k-2: walk the new candidates of one polarity each, close clusters, min-distance
//                    greedy, scatter spikes
// Each stage is a different wave of the workgroup, i.e. a different SIMD of the CU, so the serial chain of
// one stream costs max(stage) instructions per time step instead of their sum.
// ---------------------------------------------------------------------------------------------------
constexpr int RZ_MT = 16;

// ---- detect-stage helpers: per-lane updates driven by wave-level lane masks (SGPR pairs) ------------------------
// v += 1 in the lanes of m (one VALU instruction: add with carry-in)
__device__ __forceinline__ int add_lane_mask(int v, uint64_t m)
{
    uint64_t carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %0, 0, %2" : "+v"(v), "=s"(carry_out) : "s"(m));
    return v;
}
// v = J in the lanes of m (J an inline constant: the lane mask is the only scalar operand)
template <int J>
__device__ __forceinline__ int set_lane_mask(int v, uint64_t m)
{
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v) : "n"(J), "s"(m));
    return v;
}

struct DetectState {
    double prev;    // previous value of the cumulative sum (NaN before the first sample: no event at t = 0)
    int lrel;       // time of the last strict change, relative to the current tile
    int n;          // candidates appended so far
    uint64_t dpos;  // lanes whose last strict change was a rise
    uint64_t dneg;  // ... a fall
    uint64_t ffall; // lanes whose FIRST strict change was a fall (their first candidate is a minimum)
};

// One time step of the branch-free local-extremum detector (scipy _local_maxima_1d semantics, see the file header).
// 8 VALU instructions: two fp64 compares, the append counter, the plateau edge, the candidate word and three for the
// ring address; everything that concerns the direction of the last change is SALU work on lane masks.
template <int J, typename RingP, typename RingV>
__device__ __forceinline__ void detect_step(DetectState &d, double c, int word_base, uint64_t bip, int lane, RingP &ringP,
                                            RingV &ringV)
{
    const uint64_t rise = __builtin_amdgcn_fcmp(c, d.prev, 2);  // ordered >
    const uint64_t fall = __builtin_amdgcn_fcmp(c, d.prev, 4);  // ordered <
    const uint64_t ev = (fall & d.dpos) | (bip & rise & d.dneg);
    const int slot = d.n & (RZ_RING - 1);  // unconditional store; consumed only if n advances
    ringP[slot][lane] = d.lrel + word_base + J;  // left + t - 1; position = word >> 1 (plateau midpoint)
    ringV[slot][lane] = d.prev;                  // plateau value; minima negate it when they compare
    d.n = add_lane_mask(d.n, ev);
    d.ffall |= fall & ~(d.dpos | d.dneg);
    d.lrel = set_lane_mask<J>(d.lrel, rise | fall);
    d.dpos = rise | (d.dpos & ~fall);
    d.dneg = fall | (d.dneg & ~rise);
    d.prev = c;
}

template <int J0, int U, typename RingP, typename RingV>
__device__ __forceinline__ void detect_append(DetectState &d, const double (&c)[RZ_MT], const uint64_t (&chg)[8],
                                              const uint64_t (&ev)[8], int word_base, int lane, RingP &ringP, RingV &ringV)
{
    if constexpr (U < 8) {
        constexpr int J = J0 + U;
        const int slot = d.n & (RZ_RING - 1);  // unconditional store; consumed only if n advances
        ringP[slot][lane] = d.lrel + word_base + J;  // left + t - 1; position = word >> 1 (plateau midpoint)
        ringV[slot][lane] = J ? c[J ? J - 1 : 0] : d.prev;  // plateau value; minima negate it when they compare
        d.n = add_lane_mask(d.n, ev[U]);
        d.lrel = set_lane_mask<J>(d.lrel, chg[U]);
        detect_append<J0, U + 1>(d, c, chg, ev, word_base, lane, ringP, ringV);
    }
}

// Full tile in three phases per half of 8 steps, so that the VALU -> SALU -> VALU round trip of a step (compare, mask
// logic, per-lane update) is paid once per phase instead of once per step:
//   A  16 fp64 compares (independent: rise / fall of step j need only c[j-1], c[j])            -> lane masks
//   B  the direction recurrences and the event masks, pure SALU
//   C  per-lane appends: ring address, candidate word + value, counter, plateau edge
// MODE 0: general (some lane has not seen a strict change yet; polarity switch is a run-time mask)
// MODE 1 / 2: every lane has a direction, so "last change was a fall" is simply ~dpos; bipolar / unipolar fixed at
//             compile time.  6 instead of 12 SALU instructions per step -- they count: a single wave issues at most one
//             instruction of ANY kind every ~4 cycles, and this wave is one stage of a latency-bound pipeline.
template <int J0, int MODE, typename RingP, typename RingV>
__device__ __forceinline__ void detect_half(DetectState &d, const double (&c)[RZ_MT], int word_base, uint64_t bip, int lane,
                                            RingP &ringP, RingV &ringV)
{
    uint64_t rise[8], fall[8], ev[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const double p = (J0 + u) ? c[(J0 + u ? J0 + u : 1) - 1] : d.prev;
        rise[u] = __builtin_amdgcn_fcmp(c[J0 + u], p, 2);  // ordered >
        fall[u] = __builtin_amdgcn_fcmp(c[J0 + u], p, 4);  // ordered <
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (MODE == 0) {
            ev[u] = (fall[u] & d.dpos) | (bip & rise[u] & d.dneg);
            d.ffall |= fall[u] & ~(d.dpos | d.dneg);
            d.dpos = rise[u] | (d.dpos & ~fall[u]);
            d.dneg = fall[u] | (d.dneg & ~rise[u]);
        } else {
            ev[u] = MODE == 1 ? ((fall[u] & d.dpos) | (rise[u] & ~d.dpos)) : (fall[u] & d.dpos);
            d.dpos = rise[u] | (d.dpos & ~fall[u]);
        }
        rise[u] |= fall[u];  // strict change
    }
    __builtin_amdgcn_sched_barrier(0);
    detect_append<J0, 0>(d, c, rise, ev, word_base, lane, ringP, ringV);
}

template <int MODE, typename RingP, typename RingV>
__device__ __forceinline__ void detect_full(DetectState &d, const double (&c)[RZ_MT], int word_base, uint64_t bip, int lane,
                                            RingP &ringP, RingV &ringV)
{
    detect_half<0, MODE>(d, c, word_base, bip, lane, ringP, ringV);
    detect_half<8, MODE>(d, c, word_base, bip, lane, ringP, ringV);
    if (MODE != 0) d.dneg = ~d.dpos;
    d.prev = c[RZ_MT - 1];
}

// last, partial tile of a stream
template <int J, typename Tile, typename RingP, typename RingV>
__device__ __forceinline__ void detect_partial(DetectState &d, const Tile &tile, int steps, int word_base, uint64_t bip,
                                               int lane, RingP &ringP, RingV &ringV)
{
    if constexpr (J < RZ_MT) {
        if (J < steps) {  // uniform
            detect_step<J>(d, tile[J][lane], word_base, bip, lane, ringP, ringV);
            detect_partial<J + 1>(d, tile, steps, word_base, bip, lane, ringP, ringV);
        }
    }
}

template <int N, bool WANT_PRE, bool WANT_SPIKES>
__global__ __launch_bounds__(320) void bandpass_rzcc_fast_kernel(const double *__restrict__ h,
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
    __shared__ int polPub[64];  // 1: the stream's first candidate is a minimum (candidates alternate from there)
    __shared__ int deadPub[64];

    // 0: loader, 1: filter, 2: detect, 3: select maxima, 4: select minima (wave 4 shares its SIMD with the loader)
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int base = blockIdx.x * 64;
    const int NM = (T + RZ_MT - 1) / RZ_MT;
    const int NSTEP = WANT_SPIKES ? NM + 2 : NM + 1;
    const int lane_g = base + lane;
    const bool active = lane_g < nlanes;

    if (wave == 0) {
        // ------------------------------------ loader ---------------------------------------------------
        const int tl = lane & 15;  // time offset inside the tile
        const int sq = lane >> 4;  // stream slot 0..3 of each group of four
        const bool full_block = base + 64 <= nlanes;
        double v[2][16];  // two register sets: the loads of tile m are issued two tiles before they are written to LDS
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
        auto issue_loads = [&](int m, auto set) {
            constexpr int S = decltype(set)::value;
            const int t = m * RZ_MT + tl;
            const int tc = t < Ts ? t : Ts - 1;  // clamp: samples past T are never used
            int tr = (t < T ? t : T - 1) - sh;
            tr = tr < 0 ? tr + T : tr;
#pragma unroll
            for (int j = 0; j < 16; ++j) v[S][j] = rolled[j] ? pb[j][(size_t)tr * M] : pb[j][tc];
        };
        auto write_tile = [&](int buf, auto set) {
            constexpr int S = decltype(set)::value;
#pragma unroll
            for (int j = 0; j < 16; ++j) X[buf][tl][4 * j + sq] = v[S][j];
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
        using set0 = std::integral_constant<int, 0>;
        using set1 = std::integral_constant<int, 1>;
        // Loads and LDS writes are unconditional (tile index clamped; a tile past the end lands in a buffer nobody
        // reads): with straight-line code the compiler's s_waitcnt vmcnt(N) leaves the younger register set in flight.
        auto clampm = [&](int m) { return m < NM ? m : NM - 1; };
        issue_loads(0, set0{});
        write_tile(0, set0{});
        issue_loads(clampm(1), set1{});
        issue_loads(clampm(2), set0{});
        __syncthreads();
        // iteration k writes tile k+1 (register set (k+1) & 1) and refills that set with tile k+3
        auto iter = [&](int k, auto set) {
            write_tile((k + 1) % 3, set);
            issue_loads(clampm(k + 3), set);
            if (WANT_PRE && k >= 1 && k <= NM) store_tile(k - 1);
            __syncthreads();
        };
        int k = 0;
        for (; k + 1 < NSTEP; k += 2) {
            iter(k, set1{});
            iter(k + 1, set0{});
        }
        if (k < NSTEP) iter(k, set1{});
        return;
    }

    if (wave == 1) {
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
        // A maximum completes when c falls after a rise: plateau [left, t-1] -> position (left+t-1)>>1, priority =
        // plateau value = prev.  Minima mirror this (priority -prev).  Maxima and minima alternate strictly, so both
        // share one candidate list and the polarity of candidate i is (first polarity) ^ (i & 1).
        DetectState d;
        d.prev = __builtin_nan("");
        d.lrel = 0;
        d.n = 0;
        d.dpos = d.dneg = d.ffall = 0;
        const uint64_t bip = bipolar ? ~0ull : 0ull;
        nPub[lane] = 0;
        polPub[lane] = 0;
        __syncthreads();
        for (int k = 0; k < NSTEP; ++k) {
            if (k >= 1 && k <= NM) {
                const int m = k - 1;
                const int tbase = m * RZ_MT;
                const int steps = (T - tbase) < RZ_MT ? (T - tbase) : RZ_MT;
                // left + t - 1 = (tbase + lrel) + (tbase + j) - 1
                if (steps == RZ_MT) {
                    double c[RZ_MT];
#pragma unroll
                    for (int j = 0; j < RZ_MT; ++j) c[j] = X[m % 3][j][lane];
                    if ((d.dpos | d.dneg) != ~0ull)  // uniform
                        detect_full<0>(d, c, 2 * tbase - 1, bip, lane, ringP, ringV);
                    else if (bipolar)
                        detect_full<1>(d, c, 2 * tbase - 1, bip, lane, ringP, ringV);
                    else
                        detect_full<2>(d, c, 2 * tbase - 1, bip, lane, ringP, ringV);
                } else {
                    detect_partial<0>(d, X[m % 3], steps, 2 * tbase - 1, bip, lane, ringP, ringV);
                }
                d.lrel -= RZ_MT;
                nPub[lane] = d.n;
                polPub[lane] = (int)((d.ffall >> lane) & 1);
            }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------- select -------------------------------------------------------
    // One wave per polarity (wave 3: maxima, wave 4: minima).  Candidates alternate strictly, so a polarity owns every
    // second ring entry from its first one on; the two waves never touch the same entry.
    const int mypol = wave - 3;
    const bool mine = bipolar || mypol == 0;
    const int stride = bipolar ? 2 : 1;
    int i_next = -1;      // next own list index to examine (-1: the stream has no candidate yet)
    int s_open = -1;      // first list index of the open cluster (-1: none)
    int l_last = 0;       // position of the last own candidate
    bool dead = !active;  // ring overflow (or lane out of range): stop selecting; redone by the fallback kernel
    const int b = active ? lane_g / C : 0;
    const int ch = active ? lane_g - b * C : 0;
    int8_t *sp = spikes + (size_t)b * T * C + ch;
    const int8_t mark = mypol ? -1 : 1;
    const double sgn = mypol ? -1.0 : 1.0;
    auto word_at = [&](int i) { return &ringP[i & (RZ_RING - 1)][lane]; };
    auto val_at = [&](int i) { return &ringV[i & (RZ_RING - 1)][lane]; };
    auto close_cluster = [&](int s, int e, int lastpos) {
        if (e - s <= stride)
            sp[(size_t)lastpos * C] = mark;
        else
            resolve_cluster(s, e, stride, w, mark, sp, C, word_at, val_at, sgn);
    };
    if (mypol == 0) deadPub[lane] = 0;
    __syncthreads();
    for (int k = 0; k < NSTEP; ++k) {
        if (k >= 2 && mine) {
            const int n = nPub[lane];  // candidates published by the detect wave before the last barrier
            if (i_next < 0 && n > 0) i_next = bipolar ? (polPub[lane] ^ mypol) : 0;
            while (__any(!dead && i_next >= 0 && i_next < n)) {
                if (!dead && i_next >= 0 && i_next < n) {
                    const int i = i_next;
                    const int pos = *word_at(i) >> 1;
                    if (s_open >= 0 && pos - l_last >= w) {
                        close_cluster(s_open, i, l_last);
                        s_open = i;
                    }
                    if (s_open < 0) s_open = i;
                    l_last = pos;
                    i_next = i + stride;
                }
            }
            // the detect wave runs up to two tiles ahead of what has been selected: everything still open must
            // survive 2 * RZ_MT more appends
            const int oldest = s_open >= 0 ? s_open : n;
            if (!dead && n - oldest >= RZ_RING - 2 * RZ_MT) dead = true;
        }
        __syncthreads();
    }
    if (active && mine) {
        if (!dead) {
            if (s_open >= 0) close_cluster(s_open, nPub[lane], l_last);
        } else if (atomicExch(&deadPub[lane], 1) == 0) {  // flag the stream once, whichever polarity overflowed
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
                    resolve_cluster(s, e, stride, w, mark, sp, C, word_at, val_at, 1.0);
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
    dim3 grid((nlanes + 63) / 64), block(spikes ? 320 : 128);
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
