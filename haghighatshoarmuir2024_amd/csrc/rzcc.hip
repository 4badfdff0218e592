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
//   * DETECT wave: bit-parallel local-extremum detector.  Two fp64 compares per step, each folded into a per-lane 16-bit
//     word (rise / fall bits of the tile) by one add-with-carry; the direction of the last strict change before every
//     step is the carry chain of ONE addition on those words, a maximum completes where the sum falls after a rise
//     (scipy _local_maxima_1d: plateau -> midpoint, edges never peaks), minima mirror it; the two or so events of a
//     tile are appended to a per-lane LDS ring in a short loop.
//   * SELECT waves (one per polarity): scipy's _select_by_peak_distance (greedy by descending priority,
//     later peak wins ties) only couples peaks closer than `distance`, so the candidate list splits into
//     independent clusters at every same-polarity gap >= distance.  After each tile the wave walks the NEW
//     events (not the time steps), closes finished clusters, resolves them in LDS and scatters the kept
//     peaks as +1 / -1 bytes into the zero-initialised [B][T][C] spike tensor.
//   * a cluster that outgrows the LDS ring (long runs of close peaks) flags its (stream, chunk) unit; flagged
//     units are redone by a slow list-based kernel with unbounded capacity, so the result is exact for
//     every input.
//   * long streams are cut into time chunks that run side by side, each restarted from the EXACT state a
//     cheap serial scan stored at its boundary (see "Time chunking" below): bit-exact by construction.
#include "micloc_internal.h"


#include <type_traits>

// (the instantiations that also store the filtered signal are limited by LDS, not by the waves-per-SIMD hint below)
#pragma clang diagnostic ignored "-Wpass-failed"

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

    // The same step when the numerator coefficients b[i], i in ZMASK, are EXACTLY zero (every Butterworth band-pass: its zeros sit at
    // z = +-1, b = g [1, 0, -2, 0, 1]): fma(0, x, z) is z for every finite x -- as a number; only a zero's sign can differ (+-0 + -0)
    // and no comparison, sum or spike ever sees it -- so the term is skipped: 9 instead of 11 operations per step at order 4.  The
    // serial checkpoint scan of long recordings is bound by exactly this count (one wave issues one instruction per ~4.4 cycles).
    // FINITE INPUT ONLY (include/micloc_hip.h, "PRECONDITION"): for x = +-Inf the skipped product is NaN and the encode kernels, which
    // use step(), would diverge from the scan's checkpoints; tests/test_hip_chunked.py::test_chunked_signed_zeros_and_subnormals pins
    // the cases that ARE in the contract (states and samples that are -0, subnormal, exactly cancelling).
    template <unsigned ZMASK>
    __device__ __forceinline__ double step_zb(const IirCoef &coef, double xin)
    {
        static_assert(N >= 2, "needs a state");
        const double y = __builtin_fma(coef.b[0], xin, z[0]);
#pragma unroll
        for (int i = 0; i < N - 2; ++i) {
            const double t = ((ZMASK >> (i + 1)) & 1u) ? z[i + 1] : __builtin_fma(coef.b[i + 1], xin, z[i + 1]);
            z[i] = __builtin_fma(-coef.a[i + 1], y, t);
        }
        z[N - 2] = ((ZMASK >> (N - 1)) & 1u) ? -coef.a[N - 1] * y : __builtin_fma(-coef.a[N - 1], y, coef.b[N - 1] * xin);
        return y;
    }
};

// The filter coefficients arrive as kernel arguments (scalar loads).  The compiler waits for a scalar load at its first
// use; when that use sits inside the serial loop, the wait (s_waitcnt lgkmcnt(0): LDS and scalar loads share the counter)
// is re-executed every tile and drains the LDS reads issued just before it.  Touching the values once in front of the
// loop moves the wait there.
template <int N>
__device__ __forceinline__ void pin_coef(const IirCoef &coef)
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("; filter coefficient resident" ::"s"(coef.b[i]), "s"(coef.a[i]));
}

// Short clusters (the common case) are resolved in registers: all entries after one round of LDS reads, the rounds of the
// selection as unrolled compare / mask sequences of width W.  The list walk of resolve_cluster pays one LDS latency per
// entry it looks at, several times per round, and a wave waits for its slowest lane: dense candidate trains (order-1
// band-pass) were bound by it.  Three widths, because every round costs W compares whatever the cluster holds.
template <int W, typename WordAt, typename ValAt, typename Emit>
__device__ __forceinline__ void resolve_cluster_regs(int s, int stride, int count, int w, Emit emit, WordAt word_at, ValAt val_at,
                                                     double sgn)
{
    int P[W];
    double V[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const int kc = u < count ? s + u * stride : s;
        P[u] = *word_at(kc) >> 1;
        V[u] = *val_at(kc) * sgn;
    }
    unsigned alive = (1u << count) - 1u;
    while (alive) {
        int best = -1, pb = 0;
        double bv = 0.0;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (((alive >> u) & 1u) && (best < 0 || V[u] >= bv)) {  // >= : equal priority -> the later peak wins
                best = u;
                bv = V[u];
                pb = P[u];
            }
        }
        emit(pb);
        alive &= ~(1u << best);
        // positions ascend with u: "everything closer than w" is what the outward walks with their early exit remove
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int d = P[u] - pb;
            if ((d < 0 ? -d : d) < w) alive &= ~(1u << u);
        }
    }
}

// Greedy min-distance selection inside one cluster.  Entries live at list indices s, s+stride, ... < e;
// word >> 1 = position (the low bit is free for the caller), complemented once decided; priority = sgn * value.
// `at(i)` maps a list index to storage.
template <typename WordAt, typename ValAt, typename Emit>
__device__ __forceinline__ void resolve_cluster(int s, int e, int stride, int w, Emit emit, WordAt word_at, ValAt val_at,
                                                double sgn)
{
    int remaining = (e - s + stride - 1) / stride;
    // (measured on one box, encoder launch of config 2 / config 4: list walk only 0.42 / 14.7 ms; one width of 8: 0.39 / 9.2;
    // 4 + 8: 0.34 / 9.7; 2 + 4 + 8: 0.32 / 10.0; 4 + 8 + 16: 0.33 / 9.1)
    if (remaining <= 4) {
        resolve_cluster_regs<4>(s, stride, remaining, w, emit, word_at, val_at, sgn);
        return;
    }
    if (remaining <= 8) {
        resolve_cluster_regs<8>(s, stride, remaining, w, emit, word_at, val_at, sgn);
        return;
    }
    if (remaining <= 16) {
        resolve_cluster_regs<16>(s, stride, remaining, w, emit, word_at, val_at, sgn);
        return;
    }
    while (remaining > 0) {
        int best = -1;
        double bv = 0.0;
        // four list entries per trip: the LDS reads of a trip do not depend on each other, so their latencies overlap
        // (one read at a time left this loop latency bound on dense candidate trains: order-1 band-pass, config 4)
        for (int k = s; k < e; k += 4 * stride) {
            int wk[4];
            double vk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + u * stride;
                const int kc = kk < e ? kk : s;
                wk[u] = *word_at(kc);
                vk[u] = *val_at(kc) * sgn;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + u * stride;
                if (kk < e && wk[u] >= 0 && (best < 0 || vk[u] >= bv)) {  // >= : equal priority -> the later peak wins (stable sort order)
                    best = kk;
                    bv = vk[u];
                }
            }
        }
        const int wb = *word_at(best);
        const int pb = wb >> 1;
        emit(pb);
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

constexpr int RZ_MT = 16;  // time steps per tile of the wave pipeline (one barrier per tile)

// ---- detect-stage helper ------------------------------------------------------------------------------------------
// One step of BOTH words of the detect wave: r = 2 r + (a > b), f = 2 f + (a < b), ordered compares, through VCC inside one asm
// block.  With the compare outside (a builtin writing an SGPR pair that the add-with-carry reads and overwrites) the compiler put an
// s_nop behind every add-with-carry -- 30 of the wave's ~440 issue slots per tile; a compare writing VCC straight behind an
// add-with-carry that wrote it is an ordinary write after write.
__device__ __forceinline__ void rise_fall_step(unsigned &r, unsigned &f, double a, double b)
{
    asm("v_cmp_gt_f64 vcc, %2, %3\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\tv_cmp_lt_f64 vcc, %2, %3\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc"
        : "+v"(r), "+v"(f)
        : "v"(a), "v"(b)
        : "vcc");
}

// ---------------------------------------------------------------------------------------------------
// Time chunking with exact state hand-off.
//
// The serial chain of a stream (DF2T state, running sum, detector state) cannot be re-associated without changing
// bits, but it can be CHECKPOINTED: rzcc_scan_kernel walks every stream once with nothing but the band-pass and the
// running sum (the cheapest possible serial pass: two waves per 64 streams) and stores the exact state at every
// chunk boundary.  bandpass_rzcc_fast_kernel then runs one workgroup per (64 streams, chunk): it restarts the
// recurrences from the stored state -- the same operations on the same operands, hence the same bits -- and does
// the expensive part (detect + select + scatter) for its chunk, P chunks side by side.  With few, long streams
// (speech: 1750 streams x 332 157 steps) this turns a 28-workgroup latency chain into a launch that fills the chip.
//
// Chunk p owns the clusters whose FIRST candidate lies in [own_lo, own_hi).  Clusters are separated by
// same-polarity gaps >= w, so a chunk starts Vt tiles (>= w steps) early: every candidate that can chain into an
// owned cluster is seen, and a cluster whose first seen member lies before own_lo belongs to the predecessor.
// After own_hi it runs V2t more tiles to see its last cluster close; a cluster still open then, a ring overflow, or a
// detector state the scan could not pin down (a plateau longer than a tile right at the boundary) flags the
// (stream, chunk) unit, which rzcc_unit_fallback_kernel redoes serially from the same checkpoint with unbounded lists.
// ---------------------------------------------------------------------------------------------------
struct RzGeom {
    int P;    // chunks per stream (1: one pass over the whole stream, no scan kernel)
    int Lt;   // RZ_MT-step tiles owned by a chunk: chunk p owns the times [RZ_MT p Lt, RZ_MT (p+1) Lt)
    int Vt;   // look-back tiles (RZ_MT Vt >= w)
    int V2t;  // tail tiles
};

struct RzSpan {
    int m_lo, m_hi;      // tiles processed: [m_lo, m_hi)
    int own_lo, own_hi;  // owned cluster starts: [own_lo, own_hi)
    bool at_end;         // m_hi is the end of the stream
};

__host__ __device__ inline RzSpan rz_span(const RzGeom &g, int p, int NM)
{
    RzSpan s;
    s.m_lo = p == 0 ? 0 : p * g.Lt - g.Vt;
    const int hi = (p + 1) * g.Lt + g.V2t;
    s.m_hi = (p == g.P - 1 || hi > NM) ? NM : hi;
    s.own_lo = p == 0 ? 0 : p * g.Lt * RZ_MT;
    s.own_hi = p == g.P - 1 ? 0x7fffffff : (p + 1) * g.Lt * RZ_MT;
    s.at_end = s.m_hi == NM;
    return s;
}

// checkpoint q = p - 1 (state entering tile m_lo(p)):  doubles [q][N][nlanes]: DF2T state z_0..z_{N-2}, running sum;
// ints [q][3][nlanes]: direction of the last strict change (RZ_DIR_*), its time, last tile that contained one
constexpr int RZ_DIR_NONE = 0, RZ_DIR_RISE = 1, RZ_DIR_FALL = 2, RZ_DIR_UNKNOWN = 3;

// ---- loader wave (shared by the scan and the encoder kernel) ----------------------------------------------------
// 16-step x 64-stream input tiles from the planar [stream][Ts] layout (in-phase channels straight from the rolled
// input frames) into a transposed LDS tile; two register sets keep the loads of two tiles in flight.
template <bool WANT_PRE, int SW, typename XT, typename YT>
__device__ __forceinline__ void rz_loader(XT &X, YT &Y, const double *__restrict__ h, double *__restrict__ pre,
                                          const double *__restrict__ xin, int base, int nlanes, int C, int T, int Ts, int M,
                                          int shift, int m_lo, int m_hi, int nstep, int lane)
{
    const int tl = lane & 15;  // time offset inside the tile
    const int sq = lane >> 4;  // stream slot 0..3 of each group of four
    constexpr int NJ = SW / 4;  // streams per lane: stream slot 4 j + sq
    const bool full_block = base + SW <= nlanes;
    double v[2][NJ];  // two register sets: the loads of tile m are issued two tiles before they are written to LDS
    // Per-stream source.  Quadrature channels (and everything when xin == nullptr) come from the planar STHT
    // buffer h; with xin != nullptr the in-phase channels c < M are read straight from the input frames,
    // x[b][(t - L/2) mod T][c]  (np.roll, snn_beamformer.py:325), so the STHT kernel need not write them.
    const double *pb[NJ];
    bool rolled[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
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
        for (int j = 0; j < NJ; ++j) v[S][j] = rolled[j] ? pb[j][(size_t)tr * M] : pb[j][tc];
    };
    auto write_tile = [&](int buf, auto set) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j) X[buf][tl][4 * j + sq] = v[S][j];
    };
    auto store_tile = [&](int m) {  // (WANT_PRE launches are never chunked: m is also the buffer parity)
        const int t = m * RZ_MT + tl;
        const int yb = m & 1;
        if (full_block && (m + 1) * RZ_MT <= Ts) {
            double *p = pre + (size_t)(base + sq) * Ts + t;
#pragma unroll
            for (int j = 0; j < NJ; ++j) p[(size_t)(4 * j) * Ts] = Y[WANT_PRE ? yb : 0][WANT_PRE ? tl : 0][WANT_PRE ? 4 * j + sq : 0];
        } else {
#pragma unroll 4
            for (int j = 0; j < NJ; ++j) {
                const int g = base + 4 * j + sq;
                if (g < nlanes && t < Ts) pre[(size_t)g * Ts + t] = Y[WANT_PRE ? yb : 0][WANT_PRE ? tl : 0][WANT_PRE ? 4 * j + sq : 0];
            }
        }
    };
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    // Loads and LDS writes are unconditional (tile index clamped; a tile past the end lands in a buffer nobody
    // reads): with straight-line code the compiler's s_waitcnt vmcnt(N) leaves the younger register set in flight.
    const int NMc = m_hi - m_lo;
    auto clampm = [&](int kk) { return m_lo + (kk < NMc ? kk : NMc - 1); };
    issue_loads(clampm(0), set0{});
    write_tile(0, set0{});
    issue_loads(clampm(1), set1{});
    issue_loads(clampm(2), set0{});
    __syncthreads();
    // iteration k writes tile k+1 (register set (k+1) & 1) and refills that set with tile k+3
    auto iter = [&](int k, auto set) {
        write_tile((k + 1) % 3, set);
        issue_loads(clampm(k + 3), set);
        if (WANT_PRE && k >= 1 && k <= NMc) store_tile(k - 1);
        __syncthreads();
    };
    int k = 0;
    for (; k + 1 < nstep; k += 2) {
        iter(k, set1{});
        iter(k + 1, set0{});
    }
    if (k < nstep) iter(k, set1{});
}

// ---------------------------------------------------------------------------------------------------
// Scan: band-pass + running sum only, exact state at every chunk boundary.  This serial walk is what bounds long
// streams (speech: 332 157 dependent steps), so it is written for the issue floor of ONE wave: the DF2T step is
// 9 fma + 1 mul, the running sum 1 add, the plateau test 1 compare + 1 scalar or = 14 issue slots of ~4 cycles
// (tools/valu_f64_issue.hip: a lone wave sustains 4.4 cycles per fp64 instruction of this exact chain, and a second
// wave on the SIMD adds nothing -- fp64 VALU is saturated by one wave).  Everything else is kept off that wave:
//   * two LOADER waves (even / odd tiles, two register sets each = four tiles of global loads in flight; with one
//     loader and two tiles the walk was bound by HBM latency, 94 cycles per step);
//   * the FILTER wave reads tile k+1 from LDS into registers while it computes tile k from registers, so no
//     s_waitcnt lgkmcnt sits inside the arithmetic.
// ---------------------------------------------------------------------------------------------------
// loader wave `phase` of NP: owns the tiles m_lo + phase, m_lo + phase + NP, ...; tile m is written to X[m % 3] during
// iteration m - 1 (tile 0 before the first barrier); `nstep` barriers after the first one.
template <int NP, int PHASE, int SW, typename XT>
__device__ __forceinline__ void rz_loader_np(XT &X, const double *__restrict__ h, const double *__restrict__ xin, int base,
                                             int nlanes, int C, int T, int Ts, int M, int shift, int m_lo, int m_hi, int nstep,
                                             int lane)
{
    constexpr int phase = PHASE;  // compile time: the s_waitcnt vmcnt(N) in front of a tile write must be able to leave
                                  // the younger register set's loads in flight, which needs straight-line knowledge
    const int tl = lane & 15;  // time offset inside the tile
    const int sq = lane >> 4;  // stream slot 0..3 of each group of four
    constexpr int NJ = SW / 4;  // streams per lane: stream slot 4 j + sq
    double v[2][NJ];
    const double *pb[NJ];
    bool rolled[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int g = base + 4 * j + sq;
        g = g < nlanes ? g : nlanes - 1;
        const int bb = g / C, cc = g - bb * C;
        rolled[j] = xin != nullptr && cc < M;
        pb[j] = rolled[j] ? xin + (size_t)bb * T * M + cc : h + (size_t)g * Ts;
    }
    const int sh = shift % T;
    const int NMc = m_hi - m_lo;
    auto issue_loads = [&](int q, auto set) {  // own tile q (clamped: a tile past the end lands in a buffer nobody reads)
        constexpr int S = decltype(set)::value;
        int mm = phase + NP * q;
        mm = m_lo + (mm < NMc ? mm : NMc - 1);
        const int t = mm * RZ_MT + tl;
        const int tc = t < Ts ? t : Ts - 1;
        int tr = (t < T ? t : T - 1) - sh;
        tr = tr < 0 ? tr + T : tr;
        // one 32-bit element index per stream (T * M and Ts are far below 2^31): selecting between two 64-bit offsets per stream
        // costs a register pair each and pushed this wave role over the 128-VGPR budget of four waves per SIMD (32 B of scratch)
        const unsigned i_roll = (unsigned)tr * (unsigned)M, i_lin = (unsigned)tc;
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[S][j] = pb[j][rolled[j] ? i_roll : i_lin];
    };
    auto write_tile = [&](int buf, auto set) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j) X[buf][tl][4 * j + sq] = v[S][j];
    };
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    issue_loads(0, set0{});
    issue_loads(1, set1{});
    if (phase == 0) {
        write_tile(0, set0{});
        issue_loads(2, set0{});
    }
    __syncthreads();
    // iteration k writes tile m = k + 1 if it is ours: own index q = (m - phase) / NP, register set q & 1, refilled with q + 2
    // k = 4a + i writes tile m = k + 1 if it is ours ((i + 1) % NP == phase: known at compile time)
    auto iter = [&](int k, auto pos, auto set) {
        constexpr int I = decltype(pos)::value;
        if constexpr (((I + 1) % NP) == phase) {
            const int m = k + 1;
            write_tile(m % 3, set);
            issue_loads((m - phase) / NP + 2, set);
        }
        __syncthreads();
    };
    // with NP = 2 the set index of tile m is ((m - phase) / 2) & 1: period 4 in k, fixed per (k & 3, phase)
    static_assert(NP == 2, "unrolled for two loader waves");
    using p0 = std::integral_constant<int, 0>;
    using p1 = std::integral_constant<int, 1>;
    using p2 = std::integral_constant<int, 2>;
    using p3 = std::integral_constant<int, 3>;
    int k = 0;
    for (; k + 3 < nstep; k += 4) {
        iter(k, p0{}, set0{});      // m = 4a + 1: phase 1, q = 2a     -> set 0
        iter(k + 1, p1{}, set1{});  // m = 4a + 2: phase 0, q = 2a + 1 -> set 1
        iter(k + 2, p2{}, set1{});  // m = 4a + 3: phase 1, q = 2a + 1 -> set 1
        iter(k + 3, p3{}, set0{});  // m = 4a + 4: phase 0, q = 2a + 2 -> set 0
    }
    if (k < nstep) iter(k++, p0{}, set0{});
    if (k < nstep) iter(k++, p1{}, set1{});
    if (k < nstep) iter(k++, p2{}, set1{});
}

template <int N, unsigned ZMASK = 0u>
__global__ __launch_bounds__(192) void rzcc_scan_kernel(const double *__restrict__ h, IirCoef coef, int nlanes, int C, int T,
                                                         int Ts, const double *__restrict__ xin, int M, int shift, RzGeom g,
                                                         double *__restrict__ ckd, int *__restrict__ cki)
{
    __shared__ __attribute__((aligned(16))) double X[3][RZ_MT][RZ_ROW];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int base = xcd_walk(blockIdx.x, gridDim.x) * 64;
    const int m_end = (g.P - 1) * g.Lt - g.Vt;  // tile of the last checkpoint: nothing to do beyond it
    const int nstep = m_end + 2;                // the filter wave runs one tile behind the LDS ring
    if (wave == 0) {
        rz_loader_np<2, 0, 64>(X, h, xin, base, nlanes, C, T, Ts, M, shift, 0, m_end, nstep, lane);
        return;
    }
    if (wave == 1) {
        rz_loader_np<2, 1, 64>(X, h, xin, base, nlanes, C, T, Ts, M, shift, 0, m_end, nstep, lane);
        return;
    }
    // ------------------------------------ filter + checkpoints ----------------------------------------
    const int lane_g = base + lane;
    const bool active = lane_g < nlanes;
    Iir<N> iir;
    iir.init();
    double cs = 0.0;
    int chg_tile = -1;        // last tile with a strict change of the running sum
    int dir = RZ_DIR_NONE;    // direction / time of the last strict change, valid when chg_tile is the previous tile
    int left = 0;
    int next_ck = g.Lt - g.Vt;
    int q = 0;
    // tile kk: checkpoint (if due), then the walk over its 16 steps from registers
    auto istep = [&](double xv) {
        if constexpr (ZMASK != 0u && N >= 2)
            return iir.template step_zb<ZMASK>(coef, xv);
        else
            return iir.step(coef, xv);
    };
    auto tile = [&](int kk, const double (&x)[RZ_MT]) {
        if (kk == next_ck) {
            if (active) {
                const size_t nl = (size_t)nlanes;
#pragma unroll
                for (int i = 0; i < N - 1; ++i) ckd[((size_t)q * N + i) * nl + lane_g] = iir.z[i];
                ckd[((size_t)q * N + (N - 1)) * nl + lane_g] = cs;
                // the detector state is exact if the last strict change happened in the tile just walked with full
                // tracking (or never); otherwise only the tile of the last change is known
                const int d = chg_tile < 0 ? RZ_DIR_NONE : (chg_tile == kk - 1 ? dir : RZ_DIR_UNKNOWN);
                cki[((size_t)q * 3 + 0) * nl + lane_g] = d;
                cki[((size_t)q * 3 + 1) * nl + lane_g] = chg_tile < 0 ? 0 : left;
                cki[((size_t)q * 3 + 2) * nl + lane_g] = chg_tile;
            }
            next_ck += g.Lt;
            ++q;
        }
        if (kk >= m_end) return;
        uint64_t chg = 0;
        const bool t0 = kk == 0;  // the very first sample has no predecessor: never a strict change (detector: prev = NaN)
        if (kk + 1 == next_ck) {
            // tile in front of a checkpoint: track direction and time of the last strict change
#pragma unroll
            for (int j = 0; j < RZ_MT; ++j) {
                const double y = istep(x[j]);
                const double c1 = cs + y;
                const bool first = j == 0 && t0;
                const bool rise = c1 > cs && !first, fall = c1 < cs && !first;
                dir = rise ? RZ_DIR_RISE : (fall ? RZ_DIR_FALL : dir);
                left = (rise || fall) ? kk * RZ_MT + j : left;
                chg |= __ballot(rise || fall);  // exactly the detector's rise | fall
                cs = c1;
            }
        } else {
            // Any other tile only needs "did the running sum move at all in this tile".  sum(end) != sum(start) says yes
            // for one compare per TILE; equality is ambiguous (no change, or a change that came back), so a tile in which
            // some lane ends where it started is walked again from the saved state with the per-step compare -- the same
            // operations on the same operands, hence the same state.  Only plateaus (digital silence) ever take it.
            const Iir<N> iir0 = iir;
            const double cs0 = cs;
#pragma unroll
            for (int j = 0; j < RZ_MT; ++j) cs = cs + istep(x[j]);
            const uint64_t moved = __builtin_amdgcn_fcmp(cs, cs0, 6);  // ordered !=
            chg = moved;
            if (t0 || ~moved != 0ull) {  // uniform, rare (and the first tile: its first sample is never a change)
                iir = iir0;
                cs = cs0;
                chg = 0;
#pragma unroll
                for (int j = 0; j < RZ_MT; ++j) {
                    const double c1 = cs + istep(x[j]);
                    const uint64_t ne = __builtin_amdgcn_fcmp(c1, cs, 6);  // ordered != : the detector's rise | fall
                    chg |= (j == 0 && t0) ? 0ull : ne;
                    cs = c1;
                }
            }
        }
        chg_tile = ((chg >> lane) & 1) ? kk : chg_tile;
    };
    // LDS -> registers, hand-issued: the reads of tile k are in flight while tile k - 1 is computed, and the one wait sits
    // at the end of the iteration (the compiler's own placement waited right behind the reads, once per tile)
    const unsigned lds_lane = (unsigned)(size_t)(&X[0][0][lane]);
    // The destination of an asynchronous LDS read is only valid behind the s_waitcnt, and the compiler does not know that the asm is
    // asynchronous: the very objects the reads write (eight 16-byte register pairs per tile) are the ones `landed` ties to the wait
    // with "+v", and scalars are taken out of them only behind it -- no copy of a destination can be scheduled in front of the wait
    // (the first form extracted x[j] = v0[0] right behind the read, and the compiler materialised those copies ahead of
    // the wait; tests/test_isa_hazards_cpu.py checks the emitted ISA for that pattern).
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    constexpr int RZ_MT2 = RZ_MT / 2;
    auto fetch = [&](int kk, dbl2 (&x2)[RZ_MT2]) {
        const int buf = (kk < m_end ? kk : 0) % 3;  // (past the end: any buffer, the values are not used)
        const unsigned addr = lds_lane + (unsigned)buf * (RZ_MT * RZ_ROW * 8);
        // two steps per LDS instruction (a lone wave issues ONE instruction of any kind per ~4.4 cycles: every read saved is a tenth of
        // a step's arithmetic).  ds_read2_b64 offsets are 8-bit counts of 8 bytes: one base address per four rows of the tile.
        static_assert(RZ_MT % 4 == 0 && 3 * RZ_ROW < 256, "ds_read2_b64 offsets");
#pragma unroll
        for (int j = 0; j < RZ_MT; j += 4) {
            const unsigned a4 = addr + (unsigned)(j * RZ_ROW * 8);
            asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(x2[j / 2]) : "v"(a4), "n"(0), "n"(RZ_ROW));
            asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(x2[j / 2 + 1]) : "v"(a4), "n"(2 * RZ_ROW), "n"(3 * RZ_ROW));
        }
    };
    static_assert(RZ_MT2 == 8, "landed() names eight register pairs");
    auto landed = [&](dbl2 (&x2)[RZ_MT2]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(x2[0]), "+v"(x2[1]), "+v"(x2[2]), "+v"(x2[3]), "+v"(x2[4]), "+v"(x2[5]), "+v"(x2[6]), "+v"(x2[7]));
    };
    auto run = [&](int kk, const dbl2 (&x2)[RZ_MT2]) {  // (x2 has passed through landed(): plain values from here on)
        double x[RZ_MT];
#pragma unroll
        for (int j = 0; j < RZ_MT; ++j) x[j] = x2[j >> 1][j & 1];
        tile(kk, x);
    };
    dbl2 xa[RZ_MT2], xb[RZ_MT2];
    pin_coef<N>(coef);
    __syncthreads();
    // iteration k: LDS -> registers for tile k (written before the last barrier), arithmetic on tile k - 1
    fetch(0, xa);
    landed(xa);
    __syncthreads();  // k = 0
    int k = 1;
    for (; k + 1 < nstep; k += 2) {
        fetch(k, xb);
        run(k - 1, xa);
        landed(xb);
        __syncthreads();
        fetch(k + 1, xa);
        run(k, xb);
        landed(xa);
        __syncthreads();
    }
    if (k < nstep) {
        run(k - 1, xa);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Encoder: a wave pipeline over 16-step tiles (one barrier per tile), one workgroup per (64 streams, chunk).
//   wave L (loader)  tile k+1 -> LDS (loads issued two tiles earlier, into registers), stores filtered tiles
//   wave F (filter)  tile k  : x -> band-pass -> cumulative sum, in place in LDS
//   wave D (detect)  tile k-1: local maxima / minima of the cumulative sum -> candidate ring
//   waves S0, S1 (select) tile k-2: walk the new candidates of one polarity each, close clusters, min-distance
//                    greedy, scatter spikes
//   chunked launches add wave W (writer, tile k-3: stores the spike bytes the select waves queued) and wave R (resolver,
//                    tile k-3: the greedy selection inside clusters of four to eight candidates, eight clusters at a time)
// Each stage is a different wave of the workgroup, i.e. a different SIMD of the CU, so the serial chain of
// one stream costs max(stage) instructions per time step instead of their sum.
// ---------------------------------------------------------------------------------------------------
// ---- streaming (tile by tile over launches) ---------------------------------------------------------------------
// The single-pass form of the kernel is a sequential machine per stream; at the end of a launch every stage has drained
// (the pipeline skew is inside the launch), so its complete state is small and well defined: DF2T state, running sum,
// the detector's previous value / last strict change / direction, the candidate ring with the open clusters, and the
// select waves' cursors.  Dumped at the end of one launch and reloaded at the start of the next, the stream continues
// with the very same operations -- exact hand-off by construction (reference semantics: the recording is ONE stream,
// micloc/spike_encoder.py:115-137; the reference's live demo, localization_demo_snn.py:125-193, restarts every 0.25 s
// frame instead).  Positions are absolute (t_base = frames consumed so far, a multiple of RZ_MT), spikes are scattered
// into the full-length raster, clusters still open at the end of a tile are emitted by a later launch.  A stream whose
// candidate ring overflows cannot be redone from its start here (the history is gone): it is counted in `overflow`.
struct RzStream {
    double *sd;      // [nblk][N + 1][64]: z_0..z_{N-2}, running sum, detector's previous value
    int *si;         // [nblk][12][64]: left (absolute), n, flag bits, 2 x (i_next, s_open, l_last, dead), first polarity
    double *ringV;   // [nblk][RZ_RING][64]
    int *ringP;      // [nblk][RZ_RING][64]
    int *overflow;   // number of streams lost to a ring overflow
    int resume;      // load the state (not the first tile)
    int final_;      // the stream ends with this tile: close the open clusters
    int t_base;      // absolute time of this launch's first frame
    int Ttot;        // frames per trial of the spike raster (row stride of the scatter)
    int on;
    int pos_lo;      // the raster starts at this absolute frame (a sliding window over the recording; 0: the whole recording)
    const int *clk;  // device clock of a clocked stream (loc_state control words, STREAM_CLK_*) or nullptr: t_base / pos_lo / resume
                     // are read from it, so the launch carries no absolute time by value and a captured graph of it can be replayed
};

// RING: candidate ring entries per stream (power of two).  A whole tile of appends (RZ_MT) is reserved before every tile, so
// the usable depth is RING - RZ_MT.  Measured on the speech workload: RING = 32 (three workgroups per CU instead of two)
// overflows so often that the unit fallback takes 3x the time saved -- 64 it is.
// WRITER: a seventh wave that stores the spikes (see spq below) and an eighth that resolves the longer clusters (see cq below).  It pays off where the launch is occupancy bound (chunked
// launches: speech 13.1 -> 12.2 ms for scan + chunks) and costs where one workgroup per CU runs at the pace of its slowest
// wave (sweep shape: 0.417 -> 0.445 ms), so only chunked launches carry it.
template <int N, bool WANT_PRE, bool WANT_SPIKES, int RING = RZ_RING, bool WRITER = false, int SW = 64>
__global__ __launch_bounds__(WRITER ? 512 : 448, (WANT_PRE && WANT_SPIKES) ? 2 : 4) void bandpass_rzcc_fast_kernel(const double *__restrict__ h,
                                                                  double *__restrict__ pre,
                                                                  int8_t *__restrict__ spikes,
                                                                  int *__restrict__ flag_count,
                                                                  int *__restrict__ flag_list, IirCoef coef,
                                                                  int nlanes, int C, int T, int Ts, int w, int bipolar,
                                                                  const double *__restrict__ xin, int M, int shift,
                                                                  RzGeom g, int nblk, const double *__restrict__ ckd,
                                                                  const int *__restrict__ cki, RzStream ss)
{
    if (ss.on && ss.clk) {  // (scalar loads, workgroup-uniform)
        ss.t_base = ss.clk[STREAM_CLK_T];
        ss.pos_lo = ss.clk[STREAM_CLK_BASE];
        ss.resume = ss.clk[STREAM_CLK_T] != 0;
    }
    static_assert(SW == 64 || SW == 32, "streams per workgroup");
    constexpr int ROW = SW + 1;    // padded row of the transposed input tile (doubles)
    __shared__ __attribute__((aligned(16))) double X[3][RZ_MT][ROW];
    __shared__ double Y[WANT_PRE ? 2 : 1][WANT_PRE ? RZ_MT : 1][WANT_PRE ? ROW : 1];
    __shared__ double ringV[WANT_SPIKES ? RING : 1][SW];
    __shared__ int ringP[WANT_SPIKES ? RING : 1][SW];
    __shared__ int nPub[SW];
    __shared__ int polPub[SW];     // 1: the stream's first candidate is a minimum (candidates alternate from there)
    __shared__ int deadPub[SW];
    __shared__ int ovPub[SW];      // the detect wave found the ring full (or the checkpoint unusable): unit flagged
    __shared__ int leftPub[SW];    // time of the last strict change when the detect wave stopped
    __shared__ int oldPub[2][SW];  // per polarity: oldest ring entry the select wave still needs
    // Spike positions on their way to the raster.  A kept peak becomes one byte store into [B][T][C]: 64 lanes, 64 different
    // cache lines per instruction, in the middle of the select waves' dependent walk -- measured 0.13 of the kernel's 0.41 ms
    // on the sweep shape.  The select waves only queue the positions; a WRITER wave stores them one tile later, off the
    // critical path.  A lane that finds its queue full (a long cluster resolved at once) stores directly.
    constexpr int QN = 8;
    __shared__ int spq[WRITER ? 2 : 1][WRITER ? QN : 1][SW];
    __shared__ int qwPub[2][SW];
    // Clusters of four to eight candidates on their way to the RESOLVER wave (chunked launches, like the writer).  The greedy selection
    // inside such a cluster is ~80 instructions per kept peak for ONE lane -- and the whole select wave sits through it whenever one of
    // its 64 streams closes one (config 4, order-1 band-pass: two thirds of the select waves' time).  The select waves only push a
    // descriptor (stream, polarity, first ring slot, count); one tile later the resolver wave takes eight descriptors at a time, one
    // cluster per group of eight lanes, one candidate per lane: arg-max by three exchange steps, one compare per lane to remove the
    // neighbours -- the same greedy rule, the same ties (later wins), every lane busy.  A full queue (or a longer cluster) is
    // resolved in place as before.
    constexpr bool RESOLVER = WRITER;
    constexpr int QC = 64;
    __shared__ int cq[RESOLVER ? 2 : 1][RESOLVER ? QC : 1];
    __shared__ int cqn[2];
    __shared__ size_t spOff[RESOLVER ? SW : 1];

    // 0: loader, 1: filter, 2: detect, 3: select maxima, 4: select minima.  Launches that do not store the filtered signal
    // have a second loader wave in front (even / odd tiles, four tiles of global loads in flight instead of two: with
    // one loader the pipeline ran at the pace of the memory latency).
    constexpr bool LD2 = WANT_SPIKES && !WANT_PRE;
    static_assert(!WRITER || LD2, "the writer wave exists in the spikes-only launches");
    const int wave_hw = threadIdx.x >> 6;
    const int wave = LD2 ? (wave_hw == 0 ? 0 : wave_hw - 1) : wave_hw;
    const int lane = threadIdx.x & 63;
    const int vid = xcd_walk(blockIdx.x, gridDim.x);
    const int blk = vid % nblk;
    const int p = vid / nblk;
    const int base = blk * SW;
    const RzSpan sp_ = rz_span(g, p, (T + RZ_MT - 1) / RZ_MT);
    const int m_lo = sp_.m_lo;
    const int NM = sp_.m_hi - sp_.m_lo;  // tiles of this chunk
    const int NSTEP = WANT_SPIKES ? NM + 2 : NM + 1;
    const int lane_g = base + lane;
    const bool active = lane_g < nlanes;              // (lanes >= SW of the compute waves leave right after the loaders' branch)
    const int lane_c = active ? lane_g : nlanes - 1;  // clamped: inactive lanes shadow the last stream, results unused
    const size_t nl = (size_t)nlanes;
    const int sblk = lane_g >> 6, sl = lane_g & 63;  // streaming state: blocks of 64 streams, whatever SW is

    if (LD2) {
        if (wave_hw == 0) {
            rz_loader_np<2, 0, SW>(X, h, xin, base, nlanes, C, T, Ts, M, shift, m_lo, sp_.m_hi, NSTEP, lane);
            return;
        }
        if (wave_hw == 1) {
            rz_loader_np<2, 1, SW>(X, h, xin, base, nlanes, C, T, Ts, M, shift, m_lo, sp_.m_hi, NSTEP, lane);
            return;
        }
    } else if (wave == 0) {
        rz_loader<WANT_PRE, SW>(X, Y, h, pre, xin, base, nlanes, C, T, Ts, M, shift, m_lo, sp_.m_hi, NSTEP, lane);
        return;
    }

    // Fewer than 64 streams per workgroup: a wave's step time does not depend on its live lanes (the chain is latency
    // bound), so a launch with too few 64-stream workgroups to give every CU two runs 32 or 16 streams per workgroup
    // instead -- twice / four times the workgroups, the same serial chain per stream, hence the same bits.  The upper lanes
    // of the compute waves have nothing to do; they still arrive at every barrier (a barrier counts waves).
    if (SW < 64 && lane >= SW) return;

    if (wave == 1) {
        // ------------------------------------ filter ---------------------------------------------------
        Iir<N> iir;
        iir.init();
        double cs = 0.0;
        if (p > 0) {
#pragma unroll
            for (int i = 0; i < N - 1; ++i) iir.z[i] = ckd[((size_t)(p - 1) * N + i) * nl + lane_c];
            cs = ckd[((size_t)(p - 1) * N + (N - 1)) * nl + lane_c];
        }
        double *const sdb = ss.on ? ss.sd + (size_t)sblk * (N + 1) * 64 : nullptr;
        if (ss.on && ss.resume) {
#pragma unroll
            for (int i = 0; i < N - 1; ++i) iir.z[i] = sdb[i * 64 + sl];
            cs = sdb[(N - 1) * 64 + sl];
        }
        pin_coef<N>(coef);
        __syncthreads();
        for (int k = 0; k < NSTEP; ++k) {
            if (k < NM) {
                const int buf = k % 3;
                const int tb = (m_lo + k) * RZ_MT;
                const int steps = (T - tb) < RZ_MT ? (T - tb) : RZ_MT;
                auto one = [&](int j) {
                    const double y = iir.step(coef, X[buf][j][lane]);
                    if (WANT_PRE) Y[WANT_PRE ? (k & 1) : 0][WANT_PRE ? j : 0][WANT_PRE ? lane : 0] = y;
                    if (WANT_SPIKES) {
                        cs = cs + y;
                        X[buf][j][lane] = cs;
                    }
                };
                if (steps == RZ_MT && WANT_SPIKES && !WANT_PRE) {
                    // the whole tile into registers with hand-issued reads and ONE wait (the compiler's own placement waits
                    // for every pair of reads in the middle of the dependent arithmetic), then arithmetic, then the stores
                    double xr[RZ_MT];
                    const unsigned addr = (unsigned)(size_t)(&X[buf][0][lane]);
#pragma unroll
                    for (int j = 0; j < RZ_MT; ++j)
                        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(xr[j]) : "v"(addr), "n"(j * ROW * 8));
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]),
                                   "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]),
                                   "+v"(xr[15]));
#pragma unroll
                    for (int j = 0; j < RZ_MT; ++j) {
                        cs = cs + iir.step(coef, xr[j]);
                        xr[j] = cs;
                    }
#pragma unroll
                    for (int j = 0; j < RZ_MT; ++j) X[buf][j][lane] = xr[j];
                } else if (steps == RZ_MT) {
#pragma unroll
                    for (int j = 0; j < RZ_MT; ++j) one(j);
                } else {
                    for (int j = 0; j < steps; ++j) one(j);
                }
            }
            __syncthreads();
        }
        if (ss.on) {
#pragma unroll
            for (int i = 0; i < N - 1; ++i) sdb[i * 64 + sl] = iir.z[i];
            sdb[(N - 1) * 64 + sl] = cs;
        }
        return;
    }

    if (!WANT_SPIKES) return;  // (band-pass-only launches have just the two waves above)
    int *const sib = ss.on ? ss.si + (size_t)sblk * 12 * 64 : nullptr;
    const int tb0 = ss.on ? ss.t_base : 0;  // absolute time of local frame 0

    if (WRITER && wave == 5) {
        // ------------------------------------ writer ---------------------------------------------------
        const int bw = lane_c / C;
        int8_t *spw = spikes + (size_t)bw * (ss.on ? ss.Ttot : T) * C + (lane_c - bw * C);
        int rd0 = 0, rd1 = 0;
        auto drain1 = [&](int &rd, auto polc) {
            constexpr int pol = decltype(polc)::value;
            const int wr = qwPub[pol][lane];
            while (__any(rd < wr)) {
                if (rd < wr) {
                    const int pos = spq[WRITER ? pol : 0][WRITER ? (rd & (QN - 1)) : 0][lane];
                    if (active) spw[(size_t)pos * C] = pol ? (int8_t)-1 : (int8_t)1;
                    ++rd;
                }
            }
        };
        auto drain = [&]() {
            drain1(rd0, std::integral_constant<int, 0>{});
            if (bipolar) drain1(rd1, std::integral_constant<int, 1>{});
        };
        qwPub[0][lane] = 0;
        qwPub[1][lane] = 0;
        __syncthreads();
        for (int k = 0; k < NSTEP; ++k) {
            if (k >= 3) drain();  // what the select waves published at the last barrier
            __syncthreads();
        }
        drain();
        return;
    }

    if (RESOLVER && wave == 6) {
        // ------------------------------------ resolver -------------------------------------------------
        const int grp = lane >> 3, j = lane & 7;
        const int stride = bipolar ? 2 : 1;
        auto drain = [&](int buf) {
            int n = cqn[RESOLVER ? buf : 0];
            n = n < QC ? n : QC;
            n = __builtin_amdgcn_readfirstlane(n);
            for (int it = 0; it * 8 < n; ++it) {
                const int di = it * 8 + grp;
                const bool have = di < n;
                const int desc = have ? cq[RESOLVER ? buf : 0][RESOLVER ? di : 0] : 0;
                const int st = desc & 63, pol = (desc >> 6) & 1, cnt = (desc >> 7) & 15, s0 = (desc >> 11) & (RING - 1);
                bool alive = have && j < cnt;
                const int slot = (s0 + j * stride) & (RING - 1);
                const int P = ringP[WANT_SPIKES ? slot : 0][st] >> 1;
                const double V = ringV[WANT_SPIKES ? slot : 0][st] * (pol ? -1.0 : 1.0);
                int8_t *spd = spikes + spOff[RESOLVER ? st : 0];  // (a pushed descriptor belongs to an active stream)
                const int8_t mk = pol ? (int8_t)-1 : (int8_t)1;
                while (__any(alive)) {
                    // arg-max over the group's live candidates, the later one on a tie; dead lanes carry index -1.  Three exchange
                    // steps inside the group of eight lanes by DPP (neighbour, other pair, mirrored half): every lane ends with the
                    // group's best
                    double bv = V;
                    int bi = alive ? j : -1, bp = P;
                    auto step = [&](auto ctrl) {
                        constexpr int CTRL = decltype(ctrl)::value;
                        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(bv), CTRL, 0xf, 0xf, false);
                        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(bv), CTRL, 0xf, 0xf, false);
                        const double ov = __hiloint2double(hi, lo);
                        const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, false);
                        const int op = __builtin_amdgcn_update_dpp(0, bp, CTRL, 0xf, 0xf, false);
                        const bool take = oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi > bi));
                        bv = take ? ov : bv;
                        bi = take ? oi : bi;
                        bp = take ? op : bp;
                    };
                    step(std::integral_constant<int, 0xB1>{});   // quad_perm [1, 0, 3, 2]
                    step(std::integral_constant<int, 0x4E>{});   // quad_perm [2, 3, 0, 1]
                    step(std::integral_constant<int, 0x141>{});  // row_half_mirror: lane i <-> 7 - i of each eight
                    if (bi >= 0) {
                        if (j == bi) spd[(size_t)bp * C] = mk;
                        const int d = P - bp;
                        if ((d < 0 ? -d : d) < w) alive = false;  // (the kept one included: distance 0)
                    }
                }
            }
            if (lane == 0) cqn[RESOLVER ? buf : 0] = 0;
        };
        if (lane < 2) cqn[lane] = 0;
        {
            const int bb = lane_c / C;  // where a stream's spikes go (one division per stream, here, not one per descriptor)
            spOff[RESOLVER ? lane : 0] = (size_t)bb * T * C + (lane_c - bb * C);
        }
        __syncthreads();
        for (int k = 0; k < NSTEP; ++k) {
            if (k >= 3) drain((k - 1) & 1);  // what the select waves pushed during the previous tile
            __syncthreads();
        }
        drain((NSTEP - 1) & 1);
        return;
    }

    if (wave == 2) {
        // ------------------------------------ detect ---------------------------------------------------
        // A maximum completes when c falls after a rise: plateau [left, t-1] -> position (left+t-1)>>1, priority =
        // plateau value = prev.  Minima mirror this (priority -prev).  Maxima and minima alternate strictly, so both
        // share one candidate list and the polarity of candidate i is (first polarity) ^ (i & 1).
        //
        // Bit-parallel per lane: the 16 steps of a tile give two 16-bit words per lane, R (the sum rose at step j) and F (it
        // fell) -- two fp64 compares per step, each folded into its word by one add-with-carry (word = 2 word + mask bit).
        // Everything sequential in the old formulation (direction of the last strict change, plateau start) is then a
        // handful of integer operations on whole words: the direction recurrence D_{j+1} = R_j | (D_j & ~(R_j | F_j)) is
        // the carry chain of ONE addition, events are F & D (maxima) and R & ~D & "moved before" (minima), and the few
        // events of a tile (about two) are appended in a short loop instead of sixteen unconditional ring writes.
        // About 10 instead of 17 instructions per step, no scalar-unit dependency chains.
        double prev = __builtin_nan("");  // previous value of the running sum (NaN: no event at t = 0)
        int left = tb0;                   // absolute time of the last strict change
        int n = 0;                        // candidates appended so far
        int dir = 0;                      // direction of the last strict change: 0 none yet, 1 rise, 2 fall
        int ffall = 0;                    // the stream's FIRST strict change was a fall (its first candidate is a minimum)
        bool livel = true;                // this lane may still append (ring space checked before every tile)
        if (p > 0) {
            // detector state at the chunk start, from the scan's checkpoint
            const int dcode = cki[((size_t)(p - 1) * 3 + 0) * nl + lane_c];
            left = cki[((size_t)(p - 1) * 3 + 1) * nl + lane_c];
            prev = ckd[((size_t)(p - 1) * N + (N - 1)) * nl + lane_c];
            dir = dcode == RZ_DIR_RISE ? 1 : (dcode == RZ_DIR_FALL ? 2 : 0);
            ffall = dir == 2;
            livel = dcode != RZ_DIR_UNKNOWN;
        }
        if (ss.on && ss.resume) {
            double *const sdb = ss.sd + (size_t)sblk * (N + 1) * 64;
            prev = sdb[N * 64 + sl];
            left = sib[0 * 64 + sl];
            n = sib[1 * 64 + sl];
            const int bits = sib[2 * 64 + sl];
            dir = (bits & 1) ? 1 : ((bits & 2) ? 2 : 0);
            ffall = (bits >> 2) & 1;
            livel = (bits >> 3) & 1;
            // the candidate ring with the clusters that were still open
            const double *rv = ss.ringV + (size_t)sblk * RZ_RING * 64;
            const int *rp = ss.ringP + (size_t)sblk * RZ_RING * 64;
            for (int e = 0; e < RING; ++e) {
                ringV[e][lane] = rv[e * 64 + sl];
                ringP[e][lane] = rp[e * 64 + sl];
            }
        }
        nPub[lane] = n;
        polPub[lane] = ffall;
        ovPub[lane] = livel ? 0 : 1;
        __syncthreads();
        for (int k = 0; k < NSTEP; ++k) {
            if (k >= 1 && k <= NM) {
                const int m = k - 1;
                const int tloc = (m_lo + m) * RZ_MT;  // local time of the tile: input indexing and the ragged end
                const int tbase = tloc + tb0;         // absolute time: candidate positions
                const int steps = (T - tloc) < RZ_MT ? (T - tloc) : RZ_MT;
                // Ring space for a whole tile of appends?  oldPub may be the value a select wave published before the last barrier or
                // the one it is publishing during this step (no ordering between the two waves inside a step): either is a lower bound
                // of what the select waves AND the resolver wave still read -- the select waves keep a queued cluster protected through
                // two publications (pend_prev below), i.e. until the barrier behind the resolver's pass over it.
                {
                    const int o0 = oldPub[0][lane], o1 = oldPub[1][lane];
                    const int oldest = bipolar ? (o0 < o1 ? o0 : o1) : o0;
                    if (livel && n + RZ_MT - oldest > RING - 1) {
                        livel = false;
                        ovPub[lane] = 1;
                    }
                }
                // ---- A: rise / fall words (bit j = step j of the tile) ----
                double c[RZ_MT];
#pragma unroll
                for (int jj = 0; jj < RZ_MT; ++jj) c[jj] = X[m % 3][jj][lane];  // (rows past a ragged end: masked below)
                unsigned Rw = 0, Fw = 0;
#pragma unroll
                for (int jj = RZ_MT - 1; jj >= 0; --jj) {
                    const double pj = jj ? c[jj ? jj - 1 : 0] : prev;
                    rise_fall_step(Rw, Fw, c[jj], pj);  // ordered > / ordered <
                }
                const unsigned vmask = (1u << steps) - 1u;
                Rw &= vmask;
                Fw &= vmask;
                // ---- B: direction before every step, events ----
                const unsigned Sw = Rw | Fw;                        // strict changes
                const unsigned Aw = (Rw | ~Sw) & 0xFFFFu;           // generate | propagate
                const unsigned Dw = (Aw + Rw + (dir == 1 ? 1u : 0u)) ^ Aw ^ Rw;  // bit j: the last change before step j was a rise
                const unsigned low = Sw & (0u - Sw);                // lowest strict change of the tile
                const unsigned Hw = dir != 0 ? 0xFFFFu : (Sw ? (~(low | (low - 1u)) & 0xFFFFu) : 0u);  // a change happened before step j
                unsigned Ew = (Fw & Dw) | (bipolar ? (Rw & ~Dw & Hw) : 0u);
                Ew = livel ? Ew : 0u;
                if (dir == 0 && Sw) ffall = (Fw & low) ? 1 : 0;
                // ---- C: append the events (time order; maxima and minima alternate) ----
                // The plateau value of an event is read from the LDS tile at an index only the event knows.  One event per
                // trip with that read in the middle cost a full LDS latency per event, and a wave makes as many trips as its
                // busiest lane has events: the first DET_PF events of every lane are located with integer operations, their
                // values fetched together, then appended; a lane with more events finishes in the loop below.
                constexpr int DET_PF = 3;
                {
                    int jev[DET_PF], lfv[DET_PF];
                    double vev[DET_PF];
                    const char *xt = reinterpret_cast<const char *>(&X[m % 3][0][lane]);
#pragma unroll
                    for (int u = 0; u < DET_PF; ++u) {
                        const bool has = Ew != 0u;
                        const int je = has ? __builtin_ctz(Ew) : 0;
                        Ew &= Ew - 1u;  // (0 stays 0)
                        const unsigned below = Sw & ((1u << je) - 1u);
                        lfv[u] = below ? tbase + (31 - __builtin_clz(below)) : left;
                        jev[u] = has ? je : -1;
                        // plateau value: the sum just before the step that completes the candidate
                        const double v = *reinterpret_cast<const double *>(xt + (size_t)(je > 0 ? je - 1 : 0) * ROW * 8);
                        vev[u] = je > 0 ? v : prev;
                    }
                    // unconditional stores: a lane without a (further) event writes into its next free slot without taking it (a whole
                    // tile of appends is reserved, see the ring-space check) -- no exec-mask bookkeeping around six LDS writes
#pragma unroll
                    for (int u = 0; u < DET_PF; ++u) {
                        const int slot = n & (RING - 1);
                        ringP[slot][lane] = lfv[u] + tbase + jev[u] - 1;  // left + t - 1; position = word >> 1 (plateau midpoint)
                        ringV[slot][lane] = vev[u];
                        n += jev[u] >= 0 ? 1 : 0;
                    }
                }
                while (__any(Ew != 0u)) {
                    if (Ew) {
                        const int je = __builtin_ctz(Ew);
                        Ew &= Ew - 1u;
                        const unsigned below = Sw & ((1u << je) - 1u);
                        const int lf = below ? tbase + (31 - __builtin_clz(below)) : left;
                        const double val = je ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(&X[m % 3][0][lane]) +
                                                                                 (size_t)(je - 1) * ROW * 8)
                                              : prev;
                        const int slot = n & (RING - 1);
                        ringP[slot][lane] = lf + tbase + je - 1;
                        ringV[slot][lane] = val;
                        ++n;
                    }
                }
                // ---- D: state after the tile ----
                if (Sw) {
                    left = tbase + (31 - __builtin_clz(Sw));
                    dir = (Dw >> RZ_MT) & 1u ? 1 : 2;
                }
                prev = steps == RZ_MT ? c[RZ_MT - 1] : X[m % 3][steps - 1][lane];
                if (k == NM) {
                    // time of the last strict change; a lane that has not moved at all yet cannot complete a candidate
                    // whose plateau starts before the first step after this chunk
                    leftPub[lane] = dir != 0 ? left : tbase + RZ_MT + 1;
                }
                if (k == NM && ss.on) {
                    double *const sdb = ss.sd + (size_t)sblk * (N + 1) * 64;
                    sdb[N * 64 + sl] = prev;
                    sib[0 * 64 + sl] = left;  // absolute time of the last strict change
                    sib[1 * 64 + sl] = n;
                    sib[2 * 64 + sl] = (dir == 1 ? 1 : 0) | (dir == 2 ? 2 : 0) | (ffall << 2) | ((livel ? 1 : 0) << 3);
                }
                nPub[lane] = n;
                polPub[lane] = ffall;
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
    int s_first = 0;      // position of the open cluster's first candidate (kept in a register: no LDS read when it closes)
    bool dead = !active;  // ring overflow (or lane out of range): stop selecting; redone by the fallback kernel
    bool was_dead = false;  // (streaming: already lost in an earlier tile)
    const int b = lane_c / C;
    const int ch = lane_c - b * C;
    int8_t *sp = spikes + (size_t)b * (ss.on ? ss.Ttot : T) * C + ch;
    const int8_t mark = mypol ? -1 : 1;
    const double sgn = mypol ? -1.0 : 1.0;
    const int own_lo = sp_.own_lo, own_hi = sp_.own_hi;
    auto word_at = [&](int i) { return &ringP[i & (RING - 1)][lane]; };
    auto val_at = [&](int i) { return &ringV[i & (RING - 1)][lane]; };
    int widx = 0, w1 = 0, w2 = 0;  // queue write index now / at the last barrier / at the one before
    int qbuf = -1;                 // resolver queue of the current tile (-1: resolve in place)
    int pend_lo = 0x7fffffff;      // first list index of the clusters pushed during the current tile: they stay in the ring one more tile
    auto emit = [&](int pos) {
        if (WRITER && widx - w2 < QN) {
            spq[WRITER ? mypol : 0][WRITER ? (widx & (QN - 1)) : 0][lane] = pos;
            ++widx;
        } else if (!ss.on) {
            sp[(size_t)pos * C] = mark;
        } else {
            // streaming: the raster may be a window [pos_lo, pos_lo + Ttot) over the recording; a spike that lies outside it
            // cannot be stored (the window slid past a cluster that was still open): counted, never written out of bounds
            const int rel = pos - ss.pos_lo;
            if (rel >= 0 && rel < ss.Ttot)
                sp[(size_t)rel * C] = mark;
            else
                atomicAdd(ss.overflow, 1);
        }
    };
    auto close_cluster = [&](int s, int e, int lastpos, int first) {
        if (first < own_lo || first >= own_hi) return;  // the cluster belongs to a neighbouring chunk
        if (e - s <= stride) {
            emit(lastpos);
        } else if (e - s <= 2 * stride) {
            // two candidates (the most frequent multi-candidate cluster): they lie within w of each other by construction, so the
            // better one -- the later one on a tie -- is the spike; both positions are in registers, two LDS reads for the priorities
            const double v0 = *val_at(s) * sgn, v1 = *val_at(s + stride) * sgn;
            emit(v1 >= v0 ? lastpos : first);
        } else if (e - s <= 3 * stride) {
            // three candidates p0 < p1 < p2, consecutive gaps < w (the order-1 band-pass of the Xylo path produces them by the
            // thousand): the greedy rule in closed form -- the best one (the later one on a tie) is a spike and removes its
            // neighbours; if that was an end and the other end is >= w away, the other end is a spike too.  Three priorities and
            // the middle position from LDS, no selection loop for the whole wave to sit through.
            const double v0 = *val_at(s) * sgn, v1 = *val_at(s + stride) * sgn, v2 = *val_at(s + 2 * stride) * sgn;
            const bool mid = v1 >= v0 && v2 < v1;  // arg-max with "later wins": 1 beats 0 on >=, 2 beats the best so far on >=
            if (mid) {
                emit(*word_at(s + stride) >> 1);
            } else {
                const bool last_best = v2 >= (v1 >= v0 ? v1 : v0);
                emit(last_best ? lastpos : first);
                if (lastpos - first >= w) emit(last_best ? first : lastpos);
            }
        } else {
            // (collecting the clusters of three or more candidates in a per-lane queue and resolving them once per tile instead of
            // once per trip -- in THIS wave -- was measured and rejected: select waves 3300 -> 4200 cycles per tile on config 4)
            const int cnt = (e - s + stride - 1) / stride;
            if (RESOLVER && qbuf >= 0 && cnt <= 8) {
                const int slot = atomicAdd(&cqn[qbuf], 1);
                if (slot < QC) {
                    cq[RESOLVER ? qbuf : 0][RESOLVER ? slot : 0] = lane | (mypol << 6) | (cnt << 7) | ((s & (RING - 1)) << 11);
                    pend_lo = s < pend_lo ? s : pend_lo;
                    return;
                }
            }
            resolve_cluster(s, e, stride, w, emit, word_at, val_at, sgn);
        }
    };
    if (mypol == 0) deadPub[lane] = 0;
    oldPub[mypol][lane] = 0;
    if (ss.on && ss.resume && mine) {
        i_next = sib[(3 + 4 * mypol) * 64 + sl];
        s_open = sib[(4 + 4 * mypol) * 64 + sl];
        l_last = sib[(5 + 4 * mypol) * 64 + sl];
        was_dead = sib[(6 + 4 * mypol) * 64 + sl] != 0;
        dead = dead || was_dead;
        s_first = s_open >= 0 ? ss.ringP[(size_t)sblk * RZ_RING * 64 + (s_open & (RING - 1)) * 64 + sl] >> 1 : 0;
        oldPub[mypol][lane] = dead ? 0x7fffffff : (s_open >= 0 ? s_open : (i_next >= 0 ? i_next : 0));
    }
    if (!mine) oldPub[mypol][lane] = 0x7fffffff;
    __syncthreads();
    for (int k = 0; k < NSTEP; ++k) {
        if (k >= 2 && mine) {
            const int n = nPub[lane];  // candidates published by the detect wave before the last barrier
            if (ovPub[lane]) dead = true;
            qbuf = RESOLVER ? (k & 1) : -1;
            // the clusters queued during the previous tile are read by the resolver wave during THIS one: they stay protected in this
            // step's publication too (the detect wave may already see it at the top of this very step)
            const int pend_prev = pend_lo;
            pend_lo = 0x7fffffff;
            if (i_next < 0 && n > 0) i_next = bipolar ? (polPub[lane] ^ mypol) : 0;
            // one candidate per lane and trip.  (Fetching the words of the next four candidates up front was measured and
            // rejected: most tiles bring one new candidate per lane, the three extra reads and selects cost more than the
            // latency they hide -- select waves 1450 -> 1720 cycles per tile on config 2.)
            while (__any(!dead && i_next >= 0 && i_next < n)) {
                if (!dead && i_next >= 0 && i_next < n) {
                    const int i = i_next;
                    const int pos = *word_at(i) >> 1;
                    if (s_open >= 0 && pos - l_last >= w) {
                        close_cluster(s_open, i, l_last, s_first);
                        s_open = -1;
                    }
                    if (s_open < 0) {
                        s_open = i;
                        s_first = pos;
                    }
                    l_last = pos;
                    i_next = i + stride;
                }
            }
            // everything from the open cluster on must survive in the ring; without one, everything not yet examined -- and what the
            // resolver wave reads during the next tile
            {
                int keep = s_open >= 0 ? s_open : (i_next >= 0 ? i_next : 0);
                keep = pend_lo < keep ? pend_lo : keep;
                oldPub[mypol][lane] = dead ? 0x7fffffff : (pend_prev < keep ? pend_prev : keep);
            }
            qbuf = -1;
        }
        if (WRITER && mine) {
            qwPub[mypol][lane] = widx;
            w2 = w1;
            w1 = widx;
        }
        __syncthreads();
    }
    // (clusters closed from here on -- end of the stream / chunk -- are stored directly: the writer has left its loop)
    w2 = widx - QN;
    if (ss.on) {
        if (ovPub[lane]) dead = true;
        if (mine) {
            sib[(3 + 4 * mypol) * 64 + sl] = i_next;
            sib[(4 + 4 * mypol) * 64 + sl] = s_open;
            sib[(5 + 4 * mypol) * 64 + sl] = l_last;
            sib[(6 + 4 * mypol) * 64 + sl] = dead ? 1 : 0;
        }
        // a stream that lost its ring cannot be redone from its start (the history is gone): count it once
        if (active && mine && dead && !was_dead && atomicExch(&deadPub[lane], 1) == 0) atomicAdd(ss.overflow, 1);
        __syncthreads();  // (both select waves: the ring is final)
        if (mypol == 0) {
            double *rv = ss.ringV + (size_t)sblk * RZ_RING * 64;
            int *rp = ss.ringP + (size_t)sblk * RZ_RING * 64;
            for (int e = 0; e < RING; ++e) {
                rv[e * 64 + sl] = ringV[e][lane];
                rp[e * 64 + sl] = ringP[e][lane];
            }
        }
        if (!ss.final_) return;
        if (active && mine && !dead && s_open >= 0) close_cluster(s_open, nPub[lane], l_last, s_first);  // end of the stream
        return;
    }
    if (active && mine) {
        if (ovPub[lane]) dead = true;
        // the next candidate the stream can still produce completes at t' >= t_end (the first step after this
        // chunk) with its plateau starting at left' >= left: position (left' + t' - 1) >> 1 >= (left + t_end - 1) >> 1.
        const int t_end = (m_lo + NM) * RZ_MT;
        const int nextpos = (leftPub[lane] + t_end - 1) >> 1;
        // A plateau longer than the tail tiles: a candidate that completes after this chunk can still lie in the owned
        // range (its position is the plateau midpoint) -- nobody else would claim it.
        if (!sp_.at_end && nextpos < own_hi) dead = true;
        if (!dead && s_open >= 0) {
            const int first = s_first;
            if (first >= own_lo && first < own_hi) {
                // the open cluster is closed iff the next candidate is >= w away (or the stream ends here)
                if (sp_.at_end || nextpos - l_last >= w)
                    close_cluster(s_open, nPub[lane], l_last, s_first);
                else
                    dead = true;
            }
        }
        if (dead && atomicExch(&deadPub[lane], 1) == 0) {  // flag the unit once, whichever polarity gave up
            const int kk = atomicAdd(flag_count, 1);
            flag_list[kk] = p * nlanes + lane_g;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Fallback for flagged (stream, chunk) units: serial walk from the chunk's checkpoint, unbounded candidate lists in
// global scratch (slot-major), selection afterwards.  Slow (conditional global stores in the serial loop) but exact
// for any input.  Slot s handles the flagged units s, s + nslots, ... one after the other.
// ---------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void rzcc_unit_fallback_kernel(const double *__restrict__ h, int8_t *__restrict__ spikes,
                                                                 const int *__restrict__ flag_count,
                                                                 const int *__restrict__ flag_list, int *__restrict__ plist,
                                                                 double *__restrict__ vlist, int nslots, IirCoef coef,
                                                                 int nlanes, int C, int T, int Ts, int w, int bipolar,
                                                                 const double *__restrict__ xin, int M, int shift, RzGeom g,
                                                                 const double *__restrict__ ckd, const int *__restrict__ cki)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= nslots) return;
    const int count = *flag_count;
    const int NMall = (T + RZ_MT - 1) / RZ_MT;
    const size_t NS = (size_t)nslots;
    const size_t nl = (size_t)nlanes;
    int *P = plist + slot;
    double *V = vlist + slot;
    const int sh = shift % T;
    auto word_at = [&](int i) { return P + (size_t)i * NS; };
    auto val_at = [&](int i) { return V + (size_t)i * NS; };

    for (int idx = slot; idx < count; idx += nslots) {
        const int unit = flag_list[idx];
        const int p = unit / nlanes;
        const int lane_g = unit - p * nlanes;
        const RzSpan span = rz_span(g, p, NMall);
        const int b = lane_g / C;
        const int ch = lane_g - b * C;
        const bool rolled = xin != nullptr && ch < M;
        const double *src = rolled ? xin + (size_t)b * T * M + ch : h + (size_t)lane_g * Ts;
        auto sample = [&](int t) {
            int tr = t - sh;
            tr = tr < 0 ? tr + T : tr;
            return rolled ? src[(size_t)tr * M] : src[t];
        };
        Iir<N> iir;
        auto load_ck = [&](int q, double &cs) {  // q < 0: the stream start
            iir.init();
            cs = 0.0;
            if (q >= 0) {
#pragma unroll
                for (int i = 0; i < N - 1; ++i) iir.z[i] = ckd[((size_t)q * N + i) * nl + lane_g];
                cs = ckd[((size_t)q * N + (N - 1)) * nl + lane_g];
            }
        };
        double c = 0.0;
        int left = 0, dir = 0;  // dir: +1 rise, -1 fall, 0 none yet
        if (p > 0) {
            const int dcode = cki[((size_t)(p - 1) * 3 + 0) * nl + lane_g];
            left = cki[((size_t)(p - 1) * 3 + 1) * nl + lane_g];
            dir = dcode == RZ_DIR_RISE ? 1 : (dcode == RZ_DIR_FALL ? -1 : 0);
            if (dcode == RZ_DIR_UNKNOWN) {
                // the last strict change lies in tile Lc, more than a tile before this chunk: walk that tile again from
                // the nearest checkpoint at or before it
                const int Lc = cki[((size_t)(p - 1) * 3 + 2) * nl + lane_g];
                int pq = (Lc + g.Vt) / g.Lt;  // largest chunk whose first tile is <= Lc
                pq = pq > p - 1 ? p - 1 : pq;
                load_ck(pq - 1, c);
                const int t0 = pq == 0 ? 0 : (pq * g.Lt - g.Vt) * RZ_MT;
                for (int t = t0; t < (Lc + 1) * RZ_MT; ++t) {
                    const double c1 = c + iir.step(coef, sample(t));
                    if (t == 0) {
                        // the first sample has no predecessor: never a strict change
                    } else if (c1 > c) {
                        dir = 1;
                        left = t;
                    } else if (c1 < c) {
                        dir = -1;
                        left = t;
                    }
                    c = c1;
                }
            }
        }
        load_ck(p - 1, c);
        double prev = p > 0 ? c : __builtin_nan("");

        // ---- candidates from the chunk start until every owned cluster has provably closed -------------------
        int n = 0;
        int lastpos[2] = {0, 0};
        bool seen[2] = {false, false};
        bool done[2] = {false, !bipolar};
        const int t_lo = span.m_lo * RZ_MT;
        auto walk = [&](int t, double xt) {
            const double y = iir.step(coef, xt);
            c = c + y;
            const bool rise = c > prev;
            const bool fall = c < prev;
            if ((fall && dir > 0) || (bipolar && rise && dir < 0)) {
                const int pol = rise ? 1 : 0;
                const int pos = (left + t - 1) >> 1;
                P[(size_t)n * NS] = ((left + t - 1) & ~1) | pol;
                V[(size_t)n * NS] = rise ? -prev : prev;
                ++n;  // n <= T - 1 < capacity T
                // a cluster that starts at or after own_hi: every owned cluster of this polarity is closed
                if ((!seen[pol] || pos - lastpos[pol] >= w) && pos >= span.own_hi) done[pol] = true;
                seen[pol] = true;
                lastpos[pol] = pos;
            }
            left = (rise || fall) ? t : left;
            dir = rise ? 1 : (fall ? -1 : dir);
            prev = c;
            // the next candidate completes at t' >= t + 1 with left' >= left: position (left' + t' - 1) >> 1 >= (left + t) >> 1
            // (a stream that has not moved at all yet: its first plateau starts at t + 1 at the earliest)
            const int nextpos = dir == 0 ? t + 1 : (left + t) >> 1;
            if (nextpos >= span.own_hi) {
                if (!seen[0] || nextpos - lastpos[0] >= w) done[0] = true;
                if (!seen[1] || nextpos - lastpos[1] >= w) done[1] = true;
            }
        };
        // The samples of FB_BLK steps are fetched together (independent loads, one wait).  With one load per step in front of
        // the conditional list stores the walk ran at one memory round trip per step (0.7 us: 8.7 ms for ONE flagged unit of a
        // 12 000-frame chunk -- more than the whole chunked pass of config 4).
        constexpr int FB_BLK = 16;
        for (int t0 = t_lo; t0 < T && !(done[0] && done[1]); t0 += FB_BLK) {
            double xs[FB_BLK];
#pragma unroll
            for (int u = 0; u < FB_BLK; ++u) xs[u] = sample(t0 + u < T ? t0 + u : T - 1);
#pragma unroll
            for (int u = 0; u < FB_BLK; ++u)
                if (t0 + u < T && !(done[0] && done[1])) walk(t0 + u, xs[u]);
        }

        // ---- clusters of each polarity; the owned ones are resolved and scattered -----------------------------
        int8_t *sp = spikes + (size_t)b * T * C + ch;
        const int stride = bipolar ? 2 : 1;
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
                    const int first = *word_at(s) >> 1;
                    if (first >= span.own_lo && first < span.own_hi) {
                        if (e - s <= stride)
                            sp[(size_t)plast * C] = mark;
                        else
                            resolve_cluster(s, e, stride, w, [&](int pos) { sp[(size_t)pos * C] = mark; }, word_at, val_at, 1.0);
                    }
                    s = i;
                }
                if (!has) break;
                plast = pi;
            }
        }
    }
}

// Zero fill as an ordinary kernel.  hipMemsetAsync is avoided on purpose: captured into a HIP graph (ROCm 7.x) the
// memset NODE was observed not to be ordered before the kernel nodes that follow it -- the fallback kernel then read a
// stale flagged-stream counter and faulted on the third replay -- whereas kernel -> kernel edges are honoured.
// `extra` (may be null): a second, 256-byte block zeroed by the same launch (the encoder's counters beside its spike tensor)
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4 *__restrict__ p16, size_t n16, unsigned char *__restrict__ tail,
                                                         int ntail, uint4 *__restrict__ extra)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p16[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
    if (extra && blockIdx.x == gridDim.x - 1 && threadIdx.x < 16) extra[threadIdx.x] = make_uint4(0, 0, 0, 0);
}

static hipError_t zero_fill(void *ptr, size_t bytes, hipStream_t stream, void *extra256 = nullptr)
{
    // torch / hipMalloc buffers are at least 16-byte aligned; the scatter target [B][T][C] int8 may have any size
    const size_t n16 = bytes / 16;
    const int ntail = (int)(bytes - n16 * 16);
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<uint4 *>(ptr), n16,
                       reinterpret_cast<unsigned char *>(ptr) + n16 * 16, ntail, reinterpret_cast<uint4 *>(extra256));
    return hipGetLastError();
}

// ---- launch geometry and scratch layout -------------------------------------------------------------------------
// chunk_frames: 0 = automatic, < 0 = never chunk, > 0 = owned frames per chunk (rounded up to whole tiles)
static RzGeom rz_geom(int nlanes, int T, int w, int chunk_frames)
{
    RzGeom g;
    const int NM = (T + RZ_MT - 1) / RZ_MT;
    const int nblk = (nlanes + 63) / 64;
    g.Vt = (w + RZ_MT - 1) / RZ_MT;
    if (g.Vt < 1) g.Vt = 1;
    g.V2t = 4;
    int Lt = NM;
    if (chunk_frames > 0) {
        Lt = (chunk_frames + RZ_MT - 1) / RZ_MT;
    } else if (chunk_frames == 0) {
        // A workgroup is a serial chain over its tiles and a CU holds two of them (LDS), so a launch runs in
        // ceil(workgroups / 512) rounds of chain length.  Few workgroups (speech: 28) or long streams (config 4: 3000
        // tiles) are split so that about 2048 workgroups exist -- four full rounds of chunk length, P = floor(2048 / nblk)
        // so that the last round is full -- with chunks of at least 2048 frames (look-back + tail tiles cost 4 %).  The
        // scan's serial walk costs 0.1 - 0.5 of the single pass (it has no detector / selection), so short streams on a
        // full chip (config 2: 300 tiles, 241 workgroups) stay one exact pass.
        if (nblk < 512 && ((nblk < 160 && NM >= 2 * 128) || NM >= 1024)) {
            int want = 2048 / nblk;
            if (want < 2) want = 2;
            Lt = (NM + want - 1) / want;
            if (Lt < 128) Lt = 128;
        }
    }
    // tail tiles: a cluster still open at the end of the tail sends its unit to the serial fallback, which walks the whole
    // chunk -- for a 12 000-frame chunk longer than the whole chunked pass takes.  Long chunks can afford a long tail (3 %).
    if (Lt / 32 > g.V2t) g.V2t = Lt / 32 > 24 ? 24 : Lt / 32;
    if (Lt < g.Vt + 1) Lt = g.Vt + 1;
    int P = (NM + Lt - 1) / Lt;
    // unit ids p * nlanes + lane must fit an int; keep the checkpoint tables small
    while (P > 1 && ((long long)P * nlanes >= (1ll << 30) || P > 4096)) {
        Lt *= 2;
        P = (NM + Lt - 1) / Lt;
    }
    if (P < 1) P = 1;
    g.P = P;
    g.Lt = P == 1 ? NM : Lt;
    return g;
}

struct RzScratch {
    size_t count, list, ckd, cki, vlist, plist, total;
    int nslots;
};

static RzScratch rz_scratch(int nlanes, int T, int P)
{
    RzScratch s;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    s.count = off;
    off += 256;
    s.list = off;
    off += al((size_t)P * nlanes * sizeof(int));
    s.ckd = off;
    off += al((size_t)(P > 1 ? P - 1 : 0) * MICLOC_MAX_IIR * nlanes * sizeof(double));
    s.cki = off;
    off += al((size_t)(P > 1 ? P - 1 : 0) * 3 * nlanes * sizeof(int));
    // fallback candidate lists: T entries per slot (a unit may have to walk to the end of its stream), at most 256 MiB
    long long slots = (256ll << 20) / (12ll * (T > 0 ? T : 1));
    if (slots < 64) slots = 64;
    if (slots > (long long)P * nlanes) slots = (long long)P * nlanes;
    if (slots > 4096) slots = 4096;
    s.nslots = (int)slots;
    s.vlist = off;
    off += al((size_t)T * s.nslots * sizeof(double));
    s.plist = off;
    off += al((size_t)T * s.nslots * sizeof(int));
    s.total = al(off);
    return s;
}

size_t rzcc_scratch_bytes(int nlanes, int T, int w, int chunk_frames)
{
    return rz_scratch(nlanes, T, rz_geom(nlanes, T, w, chunk_frames).P).total;
}

int rzcc_chunks(int nlanes, int T, int w, int chunk_frames) { return rz_geom(nlanes, T, w, chunk_frames).P; }

// Streams per workgroup of the spikes-only launches: 64.  (The kernel is templated on it; 32 and 16 were measured in round 3 --
// profiles/r3/rz_sw_sweep.txt -- and are not instantiated: a workgroup is a serial chain whose pace does not depend on its live
// lanes, so halving the streams per workgroup doubles the workgroups but not the speed of a launch that is resident in one round
// anyway, and what a CU can hold is bounded by registers and LDS per stream, which do not shrink.  The lever for long streams is
// time chunking, rz_geom.)
template <int N, int SW>
static void launch_rz_spikes(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts, int w, int bipolar,
                             int8_t *spikes, int *flag_count, int *flag_list, const RzGeom &g, const double *ckd, const int *cki,
                             const double *xin, int M, int shift, hipStream_t stream)
{
    const int nblk = (nlanes + SW - 1) / SW;
    dim3 grid(nblk * g.P);
    if (g.P > 1)
        hipLaunchKernelGGL((bandpass_rzcc_fast_kernel<N, false, true, RZ_RING, true, SW>), grid, dim3(512), 0, stream, h, nullptr,
                           spikes, flag_count, flag_list, coef, nlanes, C, T, Ts, w, bipolar, xin, M, shift, g, nblk, ckd, cki,
                           RzStream{});
    else
        hipLaunchKernelGGL((bandpass_rzcc_fast_kernel<N, false, true, RZ_RING, false, SW>), grid, dim3(384), 0, stream, h, nullptr,
                           spikes, flag_count, flag_list, coef, nlanes, C, T, Ts, w, bipolar, xin, M, shift, g, nblk, ckd, cki,
                           RzStream{});
}

template <int N>
static void launch_rz(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts, int w, int bipolar,
                      double *pre, int8_t *spikes, unsigned char *scratch, const RzGeom &g, const RzScratch &sc,
                      const double *xin, int M, int shift, hipStream_t stream, int phases)
{
    const int nblk = (nlanes + 63) / 64;
    int *flag_count = scratch ? reinterpret_cast<int *>(scratch + sc.count) : nullptr;
    int *flag_list = scratch ? reinterpret_cast<int *>(scratch + sc.list) : nullptr;
    double *ckd = scratch ? reinterpret_cast<double *>(scratch + sc.ckd) : nullptr;
    int *cki = scratch ? reinterpret_cast<int *>(scratch + sc.cki) : nullptr;
    if (g.P > 1 && (phases & RZ_PHASE_SCAN)) {
        // exactly-zero numerator coefficients (Butterworth band-passes: b = g [1, 0, -2, 0, 1] / g [1, 0, -1]) are skipped by the scan
        bool alt_zero = N == 5 || N == 3;
        for (int i = 1; i < N && alt_zero; i += 2) alt_zero = coef.b[i] == 0.0;
        if constexpr (N == 5 || N == 3) {
            if (alt_zero)
                hipLaunchKernelGGL((rzcc_scan_kernel<N, (N == 5 ? 0xAu : 0x2u)>), dim3(nblk), dim3(192), 0, stream, h, coef, nlanes, C, T, Ts,
                                   xin, M, shift, g, ckd, cki);
        }
        if (!alt_zero)
            hipLaunchKernelGGL((rzcc_scan_kernel<N>), dim3(nblk), dim3(192), 0, stream, h, coef, nlanes, C, T, Ts, xin, M, shift,
                               g, ckd, cki);
    }
    if (!(phases & RZ_PHASE_ENCODE)) return;
    dim3 grid(nblk * g.P), block(spikes ? 320 : 128);
    if (pre && spikes)
        hipLaunchKernelGGL((bandpass_rzcc_fast_kernel<N, true, true>), grid, block, 0, stream, h, pre, spikes,
                           flag_count, flag_list, coef, nlanes, C, T, Ts, w, bipolar, xin, M, shift, g, nblk, ckd, cki, RzStream{});
    else if (spikes) {
        launch_rz_spikes<N, 64>(coef, h, nlanes, C, T, Ts, w, bipolar, spikes, flag_count, flag_list, g, ckd, cki, xin, M, shift, stream);
    } else
        hipLaunchKernelGGL((bandpass_rzcc_fast_kernel<N, true, false>), grid, block, 0, stream, h, pre, spikes,
                           flag_count, flag_list, coef, nlanes, C, T, Ts, w, bipolar, xin, M, shift, g, nblk, ckd, cki, RzStream{});
    if (spikes)
        hipLaunchKernelGGL((rzcc_unit_fallback_kernel<N>), dim3((sc.nslots + 63) / 64), dim3(64), 0, stream, h, spikes,
                           flag_count, flag_list, reinterpret_cast<int *>(scratch + sc.plist),
                           reinterpret_cast<double *>(scratch + sc.vlist), sc.nslots, coef, nlanes, C, T, Ts, w, bipolar, xin,
                           M, shift, g, ckd, cki);
}

hipError_t launch_bandpass_rzcc(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts,
                                int robust_width, int bipolar, double *pre, int8_t *spikes, void *scratch,
                                hipStream_t stream, const double *xin, int M, int shift, int chunk_frames, int phases)
{
    if (!pre && !spikes) return hipErrorInvalidValue;
    if (!(phases & (RZ_PHASE_SCAN | RZ_PHASE_ENCODE))) return hipErrorInvalidValue;
    if ((unsigned long long)T * (unsigned long long)(M > 0 ? M : 1) > 0xffffffffull) return hipErrorInvalidValue;  // (the loaders index a trial with 32 bits)
    // launches that also store the filtered signal walk each stream once (the loader stores tile by tile)
    const RzGeom g = rz_geom(nlanes, T, robust_width, (pre || !spikes) ? -1 : chunk_frames);
    const RzScratch sc = rz_scratch(nlanes, T, g.P);
    unsigned char *base = reinterpret_cast<unsigned char *>(scratch);
    if (spikes && (phases & RZ_PHASE_ENCODE)) {
        hipError_t e = zero_fill(spikes, (size_t)nlanes * T, stream, base + sc.count);  // (+ the 256-byte counter block: one launch)
        if (e != hipSuccess) return e;
    }
#define RZ_CASE(NN)                                                                                                 \
    case NN:                                                                                                        \
        launch_rz<NN>(coef, h, nlanes, C, T, Ts, robust_width, bipolar, pre, spikes, spikes ? base : nullptr, g, sc, \
                      xin, M, shift, stream, phases);                                                               \
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

// ---- streaming launcher ---------------------------------------------------------------------------------------------
// state buffer: [256 B header: overflow count] [sd] [si] [ringV] [ringP]
size_t rzcc_stream_state_bytes(int nlanes)
{
    const size_t nblk = (nlanes + 63) / 64;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return 256 + al(nblk * (MICLOC_MAX_IIR + 1) * 64 * sizeof(double)) + al(nblk * 12 * 64 * sizeof(int)) +
           al(nblk * RZ_RING * 64 * sizeof(double)) + al(nblk * RZ_RING * 64 * sizeof(int));
}

template <int N>
static void launch_rz_stream(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts, int w, int bipolar,
                             int8_t *spikes, const RzStream &ss, hipStream_t stream)
{
    const int nblk = (nlanes + 63) / 64;
    RzGeom g;
    g.P = 1;
    g.Lt = (T + RZ_MT - 1) / RZ_MT;
    g.Vt = 1;
    g.V2t = 4;
    hipLaunchKernelGGL((bandpass_rzcc_fast_kernel<N, false, true>), dim3(nblk), dim3(384), 0, stream, h, nullptr, spikes, nullptr, nullptr,
                       coef, nlanes, C, T, Ts, w, bipolar, nullptr, 0, 0, g, nblk, nullptr, nullptr, ss);
}

// One tile of every stream: h [nlanes][Ts] planar (local time), T frames (a multiple of RZ_MT unless `final_tile`), absolute
// start t_base; spikes = the full-length raster [B][Ttot][C] (zeroed by the caller before the first tile).
hipError_t launch_stream_encode(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts, int robust_width,
                                int bipolar, int8_t *spikes, int Ttot, long long t_base, int first_tile, int final_tile,
                                void *state, hipStream_t stream, int pos_lo, const int *clk)
{
    // (clocked: t_base / pos_lo / first_tile live in the device clock; the caller's schedule keeps the tile inside the window)
    if (!clk && (t_base % RZ_MT != 0 || t_base + T > (long long)pos_lo + Ttot)) return hipErrorInvalidValue;
    if (!final_tile && T % RZ_MT != 0) return hipErrorInvalidValue;
    const size_t nblk = (nlanes + 63) / 64;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    unsigned char *base = reinterpret_cast<unsigned char *>(state);
    RzStream ss{};
    ss.overflow = reinterpret_cast<int *>(base);
    size_t off = 256;
    ss.sd = reinterpret_cast<double *>(base + off);
    off += al(nblk * (MICLOC_MAX_IIR + 1) * 64 * sizeof(double));
    ss.si = reinterpret_cast<int *>(base + off);
    off += al(nblk * 12 * 64 * sizeof(int));
    ss.ringV = reinterpret_cast<double *>(base + off);
    off += al(nblk * RZ_RING * 64 * sizeof(double));
    ss.ringP = reinterpret_cast<int *>(base + off);
    ss.resume = first_tile ? 0 : 1;
    ss.final_ = final_tile ? 1 : 0;
    ss.t_base = (int)t_base;
    ss.Ttot = Ttot;
    ss.on = 1;
    ss.pos_lo = pos_lo;
    ss.clk = clk;
    if (first_tile && !clk) {
        hipError_t e = zero_fill(base, 256, stream);
        if (e != hipSuccess) return e;
    }
#define RZ_CASE(NN)                                                                                          \
    case NN:                                                                                                 \
        launch_rz_stream<NN>(coef, h, nlanes, C, T, Ts, robust_width, bipolar, spikes, ss, stream);          \
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

// ---- streaming localisation: which frames of the spike raster are final? ------------------------------------------------
// After a tile every stage of the encoder has drained, and a stream can still emit spikes only at positions
//   >= the first candidate of its open cluster (per polarity), or, without one,
//   >= (left + t_end - 1) >> 1, the earliest position of a candidate that has not completed yet (left: last strict change).
// F = the minimum over all streams: every frame below F has its final spikes, so the chunks below F / chunk_frames can be
// filtered and beamformed now (LIF and beamforming are causal).  One workgroup; writes, for the launches that follow,
//   range  = {lo, hi}: window-relative chunk range to compute (lo = chunks done so far, hi = chunks final now),
//   frames = frames covered once they are added (t_end when the recording ends: its last chunk may be ragged),
// and flags a window that no longer holds the rows the next chunk needs (status[1]).
// ctl: {chunks done (absolute), chunks in the open reduction block}; committed by rz_stream_commit_kernel afterwards.
__global__ __launch_bounds__(256) void rz_stream_horizon_kernel(const int *__restrict__ si, const int *__restrict__ ringP, int nlanes,
                                                                 int bipolar, int t_end, int final_, int chunk_frames, int base_chunk,
                                                                 int nwin, const int *__restrict__ ctl, int *__restrict__ range,
                                                                 int *__restrict__ frames, int *__restrict__ status, int clocked)
{
    __shared__ int red[256];
    if (clocked) {  // the device clock instead of times by value (ctl is the head of loc_state)
        t_end = ctl[STREAM_CLK_TEND];
        base_chunk = ctl[STREAM_CLK_BASE] / chunk_frames;
    }
    int f = 0x7fffffff;
    for (int g = threadIdx.x; g < nlanes; g += 256) {
        const int blk = g >> 6, l = g & 63;
        const int *sib = si + (size_t)blk * 12 * 64;
        const int left = sib[0 * 64 + l];
        const int bits = sib[2 * 64 + l];
        const bool moved = (bits & 3) != 0;
        const int nextpos = moved ? (left + t_end - 1) >> 1 : t_end;
        for (int pol = 0; pol < (bipolar ? 2 : 1); ++pol) {
            if (sib[(6 + 4 * pol) * 64 + l]) continue;  // lost to a ring overflow: reported by the encoder's counter
            const int s_open = sib[(4 + 4 * pol) * 64 + l];
            const int fp = s_open >= 0 ? ringP[((size_t)blk * RZ_RING + (s_open & (RZ_RING - 1))) * 64 + l] >> 1 : nextpos;
            f = fp < f ? fp : f;
        }
    }
    red[threadIdx.x] = f;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] = red[threadIdx.x + h] < red[threadIdx.x] ? red[threadIdx.x + h] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int F = red[0] < t_end ? red[0] : t_end;
        const int done = ctl[0];
        int ready = final_ ? (t_end + chunk_frames - 1) / chunk_frames : F / chunk_frames;
        ready = ready < done ? done : ready;
        int lo = done - base_chunk, hi = ready - base_chunk;
        // chunk `lo` reads the LIF history in front of it: one whole chunk of rows must still be in the window
        if (hi > lo && ((base_chunk > 0 && lo < 1) || hi > nwin)) {
            atomicAdd(&status[1], 1);
            hi = lo;
            ready = done;
        }
        range[0] = lo;
        range[1] = hi;
        range[2] = ready;  // absolute, for the commit
        range[3] = nwin * chunk_frames;  // frames per trial of the window (trial stride of the beamforming kernels)
        range[6] = t_end - base_chunk * chunk_frames;  // frames of the window that exist: the `T` of the beamforming launch
        frames[0] = ready * chunk_frames >= t_end ? t_end : ready * chunk_frames;  // (the last chunk of a recording may be ragged)
    }
}

__global__ void rz_stream_commit_kernel(int *__restrict__ ctl, const int *__restrict__ range, int block_chunks)
{
    const int n = range[1] - range[0];
    ctl[0] = range[2];
    ctl[1] = (ctl[1] + n) % block_chunks;
}

hipError_t launch_stream_horizon(const void *enc_state, int nlanes, int bipolar, int t_end, int final_, int chunk_frames, int base_chunk,
                                 int nwin, int *ctl, hipStream_t stream, int clocked)
{
    const size_t nblk = (nlanes + 63) / 64;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const unsigned char *base = reinterpret_cast<const unsigned char *>(enc_state);
    size_t off = 256 + al(nblk * (MICLOC_MAX_IIR + 1) * 64 * sizeof(double));
    const int *si = reinterpret_cast<const int *>(base + off);
    off += al(nblk * 12 * 64 * sizeof(int));
    off += al(nblk * RZ_RING * 64 * sizeof(double));
    const int *ringP = reinterpret_cast<const int *>(base + off);
    // ctl words: [0] done, [1] open, [4..6] range lo / hi / ready, [8] frames, [12..] status
    hipLaunchKernelGGL(rz_stream_horizon_kernel, dim3(1), dim3(256), 0, stream, si, ringP, nlanes, bipolar, t_end, final_, chunk_frames,
                       base_chunk, nwin, ctl, ctl + 4, ctl + 8, ctl + 12, clocked);
    return hipGetLastError();
}

// ---- the device clock of a stream (loc_state control words STREAM_CLK_*) ------------------------------------------------------
// begin: t frames have been pushed, the next tile brings n more; the raster window [base, base + cap) must cover [.., t + n): it
// slides forward by whole chunks when it does not -- the schedule depends on the tile sizes only, the host can mirror it.
__global__ void rz_stream_clock_begin_kernel(int *__restrict__ ctl, int n, int cap, int chunk_frames)
{
    const int t = ctl[STREAM_CLK_T], base = ctl[STREAM_CLK_BASE];
    int nb = base;
    if (t + n > base + cap) nb = (t + n - cap + chunk_frames - 1) / chunk_frames * chunk_frames;
    ctl[STREAM_CLK_SHIFT] = nb - base;
    ctl[STREAM_CLK_BASE] = nb;
    ctl[STREAM_CLK_TEND] = t + n;
}

__global__ void rz_stream_clock_tick_kernel(int *__restrict__ ctl) { ctl[STREAM_CLK_T] = ctl[STREAM_CLK_TEND]; }

// the slide itself, by the clock's shift: pass 0 copies the surviving rows of every trial to `tmp`, pass 1 back to the front of the
// window and zeroes the rows behind them (a launch with shift == 0 returns at once: both passes are part of every tile's graph)
__global__ __launch_bounds__(256) void rz_window_slide_kernel(int8_t *__restrict__ win, int8_t *__restrict__ tmp, size_t row_bytes, int C,
                                                               const int *__restrict__ ctl, int chunk_frames, int pass, int *__restrict__ status)
{
    const size_t shift_bytes = (size_t)ctl[STREAM_CLK_SHIFT] * C;
    if (shift_bytes == 0) return;
    const int b = blockIdx.y;
    int8_t *w = win + (size_t)b * row_bytes, *t = tmp + (size_t)b * row_bytes;
    const size_t keep = shift_bytes < row_bytes ? row_bytes - shift_bytes : 0;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < row_bytes; i += (size_t)gridDim.x * 256 * 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const size_t e = i + u;
            if (e >= row_bytes) break;
            if (pass == 0) {
                if (e < keep) t[e] = w[e + shift_bytes];
            } else {
                w[e] = e < keep ? t[e] : (int8_t)0;
            }
        }
    }
    // ctl[0] chunks are done: the window must keep chunk (done - 1) for the LIF history
    const int new_base_chunk = ctl[STREAM_CLK_BASE] / chunk_frames;
    if (pass == 0 && b == 0 && blockIdx.x == 0 && threadIdx.x == 0 && new_base_chunk > 0 && ctl[0] - 1 < new_base_chunk) atomicAdd(&status[1], 1);
}

hipError_t launch_stream_begin_tile(int *ctl, int8_t *win, int8_t *tmp, int B, size_t row_bytes, int C, int n, int cap, int chunk_frames,
                                    hipStream_t stream)
{
    hipLaunchKernelGGL(rz_stream_clock_begin_kernel, dim3(1), dim3(1), 0, stream, ctl, n, cap, chunk_frames);
    size_t gx = (row_bytes / 16 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(rz_window_slide_kernel, dim3((unsigned)gx, B), dim3(256), 0, stream, win, tmp, row_bytes, C, ctl, chunk_frames, pass,
                           ctl + 12);
    return hipGetLastError();
}

hipError_t launch_stream_tick(int *ctl, hipStream_t stream)
{
    hipLaunchKernelGGL(rz_stream_clock_tick_kernel, dim3(1), dim3(1), 0, stream, ctl);
    return hipGetLastError();
}

// np.roll's wrap-around inside the in-phase rows of a tile's STHT output: in-phase[t] = x[T - L/2 + t] for absolute t < L/2 (rows a
// live source cannot know: zeros without `wrap`).  h planar [B][2M][Ts]; the tile's frame i (absolute t = clock + i) sits at column
// col0 + i.  One launch per tile, a no-op once the clock has passed L/2.
__global__ __launch_bounds__(256) void stht_wrap_rows_kernel(double *__restrict__ h, const double *__restrict__ wrap, int B, int M, int Ts,
                                                              int col0, int n, int half, const int *__restrict__ ctl)
{
    const int t = ctl[STREAM_CLK_T];
    if (t >= half) return;
    const int k = half - t < n ? half - t : n;
    const int total = B * M * k;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e % k, bm = e / k;
        const int b = bm / M, m = bm - b * M;
        h[((size_t)b * 2 * M + m) * Ts + col0 + i] = wrap ? wrap[((size_t)b * half + t + i) * M + m] : 0.0;
    }
}

hipError_t launch_stht_wrap_rows(double *h, const double *wrap, int B, int M, int Ts, int col0, int n, int half, const int *ctl,
                                 hipStream_t stream)
{
    const int total = B * M * (half < n ? half : n);
    const int grid = total > 0 ? (total + 255) / 256 : 1;
    hipLaunchKernelGGL(stht_wrap_rows_kernel, dim3(grid > 1024 ? 1024 : grid), dim3(256), 0, stream, h, wrap, B, M, Ts, col0, n, half, ctl);
    return hipGetLastError();
}

hipError_t launch_stream_commit(int *ctl, int block_chunks, hipStream_t stream)
{
    hipLaunchKernelGGL(rz_stream_commit_kernel, dim3(1), dim3(1), 0, stream, ctl, ctl + 4, block_chunks);
    return hipGetLastError();
}

hipError_t launch_zero_fill(void *ptr, size_t bytes, hipStream_t stream) { return zero_fill(ptr, bytes, stream); }

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
