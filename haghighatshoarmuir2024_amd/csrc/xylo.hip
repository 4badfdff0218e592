// Xylo-A2 (SYNS61201) hidden-layer integer LIF on gfx950 -- BASELINE config 4.
//
// PARITY UNPINNED.  In the reference this stage is rockpool's XyloSim (C++/pybind `xylosim`, third party, not
// vendored, unpinned in setup.py:19, not installed in the build image) called from
// micloc/xylo_snn_localization.py:269-290 (from_config) and :358-377 (xylo_process).  The kernel follows the
// published Xylo update rule exactly as restated in oracle/micloc_oracle.c (oracle_xylo_lif): bit-shift decay with
// the "at least 1 LSB" rule, 8-bit weights, saturating 16-bit synaptic current and membrane, subtractive reset
// with a per-step spike cap, one shared recurrent weight.  HIP == oracle bit for bit; oracle != checked
// against XyloSim.
//
// Two kernels.  xylo_lif_kernel (general: spike counts in, recurrence): lane = hidden neuron, workgroup = one trial x
// up to 1024 neurons, sequential over time (integer state recurrences); the input channels are packed 4 x int8 per dword
// and contracted with v_dot4_i32_i8; input spike rows are staged 256 steps at a time in LDS and read as wave-uniform
// (broadcast) dwords.  xylo_lif_pk_kernel (the sweep: binary events, no recurrence): two neurons per lane in packed
// 16-bit arithmetic, input currents from the int8 matrix cores -- see its header.
#include <vector>

#include "micloc_internal.h"

// (the recurrent instantiations keep a barrier inside the step loop, which the requested unrolling skips)
#pragma clang diagnostic ignored "-Wpass-failed"

namespace micloc {

constexpr int XY_TT = 256;   // time steps staged per LDS tile
constexpr int XY_MAXQ = 16;  // up to 64 input channels

__device__ __forceinline__ int xy_sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

// v - dv with dv = v >> dash, "at least one LSB" while v != 0.  The arithmetic shift of a negative value is never 0
// (-1 at least), so the rule only ever fires for 0 < v < 2^dash, where it sets dv = 1:  dv = max(v >> dash, min(v, 1))
// (v > 0: max(., 1);  v = 0: 0;  v < 0: v >> dash >= v).  Three instructions instead of seven.
__device__ __forceinline__ int xy_decay(int v, int dash)
{
    const int dv = max(v >> dash, min(v, 1));
    return v - dv;
}

// exact floor(v / th) for 0 < th <= v <= 32767 without the integer-division sequence: fp32 quotient, corrected by +-1
__device__ __forceinline__ int xy_div(int v, int th)
{
    int n = (int)((float)v * __builtin_amdgcn_rcpf((float)th));
    n = n * th > v ? n - 1 : n;
    n = (n + 1) * th <= v ? n + 1 : n;
    return n;
}

// NQ = dwords of packed input per step (compile time: the contraction unrolls into NQ v_dot4 with no branches; the
// staged row is read with wide, wave-uniform LDS loads)
template <bool REC, int NQ>
__global__ __launch_bounds__(1024) void xylo_lif_kernel(const uint8_t *__restrict__ spikes_in, int ternary_C, int T, int Cin,
                                                         const int *__restrict__ Wpk /*[nq][N]*/, int nq, int N, int w_rec,
                                                         const uint8_t *__restrict__ dash_syn,
                                                         const uint8_t *__restrict__ dash_mem,
                                                         const short *__restrict__ thr, int max_spikes,
                                                         uint8_t *__restrict__ spikes_out, int *__restrict__ rate)
{
    __shared__ __attribute__((aligned(16))) int tile[XY_TT][NQ];
    __shared__ int wsum[2][16];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    const bool act = g < N;
    int w[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) w[k] = (act && k < nq) ? Wpk[(size_t)k * N + g] : 0;
    const int ds = act ? dash_syn[g] : 0, dm = act ? dash_mem[g] : 0;
    const int th = act ? thr[g] : 32767;
    int isyn = 0, vmem = 0, total = 0, prev_total = 0;
    // ternary_C > 0: the input is the encoder's int8 raster [B][T][ternary_C] in {-1, 0, +1}; channel c carries its +1
    // events, channel ternary_C + c its -1 events (Demo.spike_encoding's split, xylo_snn_localization.py:350-354)
    const int Crow = ternary_C > 0 ? ternary_C : Cin;
    const uint8_t *sb = spikes_in + (size_t)b * T * Crow;
    const int8_t *sbt = reinterpret_cast<const int8_t *>(sb);
    uint8_t *ob = spikes_out ? spikes_out + (size_t)b * T * N : nullptr;
    const int nwaves = blockDim.x >> 6;

    for (int t0 = 0; t0 < T; t0 += XY_TT) {
        const int steps = (T - t0) < XY_TT ? (T - t0) : XY_TT;
        __syncthreads();
        // stage `steps` rows of Cin bytes, zero padded to NQ dwords
        uint8_t *tb = reinterpret_cast<uint8_t *>(&tile[0][0]);
        for (int e = threadIdx.x; e < steps * NQ * 4; e += blockDim.x) {
            const int r = e / (NQ * 4), c = e % (NQ * 4);
            uint8_t v = 0;
            if (c < Cin) {
                if (ternary_C > 0)
                    v = c < ternary_C ? (uint8_t)(sbt[(size_t)(t0 + r) * Crow + c] > 0) : (uint8_t)(sbt[(size_t)(t0 + r) * Crow + c - ternary_C] < 0);
                else
                    v = sb[(size_t)(t0 + r) * Cin + c];
            }
            tb[e] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < steps; ++j) {
            int in = 0;
#pragma unroll
            for (int k = 0; k < NQ; ++k) in = __builtin_amdgcn_sdot4(w[k], tile[j][k], in, false);
            int i2 = xy_decay(isyn, ds);
            int v2 = xy_decay(vmem, dm);
            i2 = xy_sat16(i2 + in + (REC ? w_rec * prev_total : 0));
            v2 = xy_sat16(v2 + i2);
            // subtractive reset until below threshold or until the per-step cap: almost always zero or one spike, so the
            // first one is branch-free and only a wave in which some neuron still sits above threshold takes the division
            int n = v2 >= th ? 1 : 0;
            v2 -= n ? th : 0;
            if (__builtin_expect(__any(v2 >= th), 0)) {  // (cap >= 1 by the API contract)
                if (v2 >= th) {
                    int more = xy_div(v2, th);
                    more = more < max_spikes - 1 ? more : max_spikes - 1;
                    n += more;
                    v2 -= more * th;
                }
            }
            isyn = i2;
            vmem = v2;
            total += n;
            if (ob && act) ob[(size_t)(t0 + j) * N + g] = (uint8_t)n;
            if (REC) {
                // block-wide sum of this step's spikes feeds every neuron at the next step
                int s = n;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                const int p = j & 1;
                if ((threadIdx.x & 63) == 0) wsum[p][threadIdx.x >> 6] = s;
                __syncthreads();
                int tot = 0;
                for (int k2 = 0; k2 < nwaves; ++k2) tot += wsum[p][k2];
                prev_total = tot;
            }
        }
    }
    if (rate && act) rate[(size_t)b * N + g] = total;
}

// ---------------------------------------------------------------------------------------------------------------
// Packed form for the sweep (binary input events, no recurrence): two neurons per lane in 16-bit halves.
//
// The state is 16 bit by definition, so every step of the rule has a packed instruction that is exact on it:
//   decay              v - max(v >> dash, min(v, 1))       v_pk_ashrrev_i16 / v_pk_min_i16 / v_pk_max_i16 / v_pk_sub_i16
//   sat16(a + b)       both operands 16 bit                v_pk_add_i16 clamp
//   first spike        d = sat(v - th); m = d >> 15 (all ones below threshold); v = m ? v : d; count += m
//                                                          v_pk_sub_i16 clamp / v_pk_ashrrev_i16 / v_bfi_b32 / v_pk_add_i16
// (a shift count above 15 acts like 15 on a 16-bit value, so dash is clamped; the spike count of a 128-step tile is
// 128 + the sum of the m's and is widened to 32 bit per tile).  9 instructions per neuron-step instead of ~41.
//
// The input current in[t][n] = sum_c W_in[c][n] s[t][c] is a plain matrix product of the event raster with the weight
// matrix, identical for every trial: it runs on the matrix cores (v_mfma_i32_16x16x32_i8, one instruction per 16 steps x
// 16 neurons; |in| <= 64 x 127 fits 16 bit because the events are 0/1).  A wave owns 128 neurons = 8 column tiles, whose
// weight fragments it keeps in registers; lane (q, lc) integrates neurons 32 q + lc and 32 q + 16 + lc of the wave.  The
// accumulators of column tiles 2k and 2k + 1 hold exactly those two neurons in lane (., lc), so one v_perm_b32 per step
// packs them; the wave parks the packed currents of the next 16 steps in its own LDS slice ([4 steps][lane], 16 B per
// lane: conflict-free both ways) while it integrates the current 16, and no barrier is needed inside a 128-step tile.
// The rare second spike within one step (v >= 2 th) leaves the packed path for the exact 32-bit sequence.
// ---------------------------------------------------------------------------------------------------------------
typedef short short2_t __attribute__((ext_vector_type(2)));
typedef int int4_t __attribute__((ext_vector_type(4)));

constexpr int XP_WAVES = 4;                  // 4 x 128 = 512 neurons per workgroup
constexpr int XP_TT = 128;                   // time steps staged per LDS tile: 4 KB + 4 KB per wave = 16 KB for the sweep's 360
                                             // neurons: all 1100 trials are resident at once (4-5 workgroups per CU)
constexpr int XP_NEUR = XP_WAVES * 128;

__device__ __forceinline__ short2_t xp_s2(int w) { return __builtin_bit_cast(short2_t, w); }
__device__ __forceinline__ int xp_i(short2_t v) { return __builtin_bit_cast(int, v); }

__device__ __forceinline__ short2_t xp_decay(short2_t v, short2_t dash)
{
    const short2_t one = {1, 1};
    return v - __builtin_elementwise_max(v >> dash, __builtin_elementwise_min(v, one));
}

// QUEUE (the sweep: counts only): the launch is a fixed set of PERSISTENT workgroups -- four per CU, twelve waves on four SIMDs --
// that take (trial, time chunk) tickets from a device counter, trial fastest, so that every trial advances chunk by chunk and the
// integer state of a trial (two packed words and two counts per lane) is handed from one workgroup to the next through memory.
// A trial is a serial chain of T steps and all chains are equally long: launched as one workgroup per trial, 1100 x 3 waves land
// on 1024 SIMDs as three waves here and four there, and the launch lasts as long as the SIMDs that got four (80 % balance,
// tools/wave_placement.hip).  With tickets every SIMD holds the same three waves from start to end and the chip as a whole works
// off ceil(chunks x trials / 1024) rounds.  A ticket's predecessor (same trial, previous chunk) was handed out `trials` tickets
// earlier to a workgroup that is RUNNING (tickets are only taken by resident workgroups), so the wait for it always ends; it is
// bounded all the same (a worker that gives up sets the error word and the launch drains).
constexpr int XQ_CTL = 64;         // control ints in front of the per-trial chunk counters: [0] ticket, [1] error
constexpr int XQ_SPIN_MAX = 1 << 24;  // x (s_sleep 8 + one coherent load, ~2 us): half a minute -- a predecessor on a crowded chip takes milliseconds

template <int KQ, bool WANT_OUT, bool QUEUE = false>
__global__ __launch_bounds__(XP_WAVES * 64) void xylo_lif_pk_kernel(const int8_t *__restrict__ raster, int tc, int T, int Cin,
                                                                     const long *__restrict__ Wb /*[Npad][4 KQ]*/, int N,
                                                                     const uint8_t *__restrict__ dash_syn,
                                                                     const uint8_t *__restrict__ dash_mem,
                                                                     const short *__restrict__ thr, int max_spikes,
                                                                     uint8_t *__restrict__ spikes_out, int *__restrict__ rate,
                                                                     int *__restrict__ qctl, int4_t *__restrict__ qstate, int Lc, int ntrials)
{
    __shared__ __attribute__((aligned(16))) unsigned char tile[XP_TT][32 * KQ];
    extern __shared__ __attribute__((aligned(16))) int cur_dyn[];  // [waves][4][64][4]: only the waves that hold neurons
    int(*cur)[4][64][4] = reinterpret_cast<int(*)[4][64][4]>(cur_dyn);
    int8_t *raw = reinterpret_cast<int8_t *>(cur_dyn);  // staging area for the raster bytes of a tile (between barriers)
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int wv = tid >> 6, l = tid & 63, lc = l & 15, q = l >> 4;
    int b = QUEUE ? 0 : blockIdx.y;
    const int nbase = (QUEUE ? 0 : blockIdx.x * XP_NEUR) + wv * 128;  // (QUEUE: one neuron block, N <= 512)

    // weight fragments of the wave's 8 column tiles: lane (q, lc) holds channels 8 q .. 8 q + 7 (+ 32 kq) of neuron 16 j + lc
    long Bw[8][KQ];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) Bw[j][kq] = Wb[(size_t)(nbase + 16 * j + lc) * (4 * KQ) + 4 * kq + q];

    const int n0 = nbase + 32 * q + lc, n1 = n0 + 16;
    const bool act0 = n0 < N, act1 = n1 < N;
    auto clamp15 = [](int d) { return d > 15 ? 15 : d; };
    const short2_t ds = {(short)(act0 ? clamp15(dash_syn[n0]) : 0), (short)(act1 ? clamp15(dash_syn[n1]) : 0)};
    const short2_t dm = {(short)(act0 ? clamp15(dash_mem[n0]) : 0), (short)(act1 ? clamp15(dash_mem[n1]) : 0)};
    const short2_t th = {(short)(act0 ? thr[n0] : 32767), (short)(act1 ? thr[n1] : 32767)};
    const int th_hi = (int)th.y << 16;  // v.hi >= th.hi  <=>  (int)word >= th.hi << 16
    short2_t isyn = {0, 0}, vmem = {0, 0};
    int total0 = 0, total1 = 0;

    const int8_t *sb = raster + (size_t)b * T * tc;
    uint8_t *ob = WANT_OUT ? spikes_out + (size_t)b * T * N : nullptr;
    const int nraster_trials = QUEUE ? ntrials : (int)gridDim.y;

    // currents of 16 steps (tile16 `i` of the staged rows) -> the wave's LDS slice.  ONE slice per wave (4 KB): the 16 steps of
    // a tile are read into registers before the next tile's currents overwrite them (LDS operations of a wave execute in
    // order), so a workgroup takes 16 instead of 28 KB and a CU that holds four or five of them still has room for a
    // workgroup of the stages in front (the chunked encoder: 80 KB) -- with 28 KB the three stages of the sweep took turns.
    auto produce = [&](int i) {
        int4_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = int4_t{0, 0, 0, 0};
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const long a = *reinterpret_cast<const long *>(&tile[16 * i + lc][32 * kq + 8 * q]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, Bw[j][kq], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int4_t pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pk[r] = (int)__builtin_amdgcn_perm((unsigned)acc[2 * k + 1][r], (unsigned)acc[2 * k][r], 0x05040100u);
            *reinterpret_cast<int4_t *>(&cur[wv][q][16 * k + lc][0]) = pk;  // steps 4 q .. 4 q + 3 of pair lane 16 k + lc
        }
    };

    short2_t cnt = {0, 0};
    // The second-spike test (v >= th again after the reset: two compares, a scalar or and a branch per step) is not run per step
    // in whole tiles: the steps only keep the largest membrane word they left behind, one test per tile decides, and a tile in
    // which some step did need the exact sequence is walked again from its saved state with the per-step test (`step` below).
    short2_t vmx = {-32768, -32768};
    auto step_fast = [&](int in_word, int t) {
        short2_t i2 = xp_decay(isyn, ds);
        short2_t v2 = xp_decay(vmem, dm);
        i2 = __builtin_elementwise_add_sat(i2, xp_s2(in_word));
        v2 = __builtin_elementwise_add_sat(v2, i2);
        const short2_t d = __builtin_elementwise_sub_sat(v2, th);
        const short2_t fifteen = {15, 15};
        const short2_t m = d >> fifteen;  // all ones: below threshold
        const int vw = (xp_i(m) & xp_i(v2)) | (~xp_i(m) & xp_i(d));
        cnt += m;
        vmx = __builtin_elementwise_max(vmx, xp_s2(vw));
        if constexpr (WANT_OUT) {
            const size_t row = (size_t)t * N;
            if (act0) ob[row + n0] = (uint8_t)(1 + m.x);
            if (act1) ob[row + n1] = (uint8_t)(1 + m.y);
        }
        isyn = i2;
        vmem = xp_s2(vw);
    };
    auto step = [&](int in_word, int t) {
        short2_t i2 = xp_decay(isyn, ds);
        short2_t v2 = xp_decay(vmem, dm);
        i2 = __builtin_elementwise_add_sat(i2, xp_s2(in_word));
        v2 = __builtin_elementwise_add_sat(v2, i2);
        const short2_t d = __builtin_elementwise_sub_sat(v2, th);
        const short2_t fifteen = {15, 15};
        const short2_t m = d >> fifteen;  // all ones: below threshold
        int vw = (xp_i(m) & xp_i(v2)) | (~xp_i(m) & xp_i(d));
        cnt += m;
        int extra0 = 0, extra1 = 0;
        // a second spike in the same step (v >= 2 th): exact 32-bit sequence, per half
        const bool again = ((short)vw >= th.x) | (vw >= th_hi);
        if (__builtin_expect(__any(again), 0)) {
            int lo = (short)vw, hi = vw >> 16;
            if (lo >= th.x) {
                int more = xy_div(lo, th.x);
                more = more < max_spikes - 1 ? more : max_spikes - 1;
                extra0 = more;
                lo -= more * th.x;
            }
            if (hi >= th.y) {
                int more = xy_div(hi, th.y);
                more = more < max_spikes - 1 ? more : max_spikes - 1;
                extra1 = more;
                hi -= more * th.y;
            }
            vw = (lo & 0xffff) | (hi << 16);
            total0 += extra0;
            total1 += extra1;
        }
        if constexpr (WANT_OUT) {
            const size_t row = (size_t)t * N;
            if (act0) ob[row + n0] = (uint8_t)(1 + m.x + extra0);
            if (act1) ob[row + n1] = (uint8_t)(1 + m.y + extra1);
        }
        isyn = i2;
        vmem = xp_s2(vw);
    };

    auto run_range = [&](int t_lo, int t_hi) {
    for (int t0 = t_lo; t0 < t_hi; t0 += XP_TT) {
        const int steps = (t_hi - t0) < XP_TT ? (t_hi - t0) : XP_TT;
        __syncthreads();  // every wave is done with the previous tile and with its slice of `cur`
        // the 128 x tc raster bytes of this tile are contiguous: one 16-byte load per thread into LDS (the slices of `cur`
        // are idle between these barriers), then the +1 / -1 split from there (channel c < tc: +1 events, tc + c: -1
        // events, zero padded to 32 KQ).  A byte-wise gather from global memory costs one exposed latency per element.
        {
            const int8_t *src = sb + (size_t)t0 * tc;
            const uintptr_t a0 = reinterpret_cast<uintptr_t>(src) & ~(uintptr_t)15;  // >= raster: the launcher checks its alignment
            const int off = (int)(reinterpret_cast<uintptr_t>(src) - a0);
            const int nvec = (off + steps * tc + 15) >> 4;
            const uint4 *vsrc = reinterpret_cast<const uint4 *>(a0);
            const int8_t *rend = raster + (size_t)nraster_trials * T * tc;
            for (int e = tid; e < nvec; e += nthreads) {
                uint4 v;
                if (reinterpret_cast<const int8_t *>(vsrc + e + 1) <= rend) {
                    v = vsrc[e];
                } else {  // the last vector of the whole raster: only the bytes that exist
                    unsigned w4[4] = {0u, 0u, 0u, 0u};
                    const int8_t *p = reinterpret_cast<const int8_t *>(vsrc + e);
                    for (int i = 0; i < 16 && p + i < rend; ++i) w4[i >> 2] |= (unsigned)(unsigned char)p[i] << (8 * (i & 3));
                    v = uint4{w4[0], w4[1], w4[2], w4[3]};
                }
                reinterpret_cast<uint4 *>(raw)[e] = v;
            }
            __syncthreads();
            unsigned *tile32 = reinterpret_cast<unsigned *>(&tile[0][0]);
            for (int e = tid; e < XP_TT * 8 * KQ; e += nthreads) {
                const int r = e / (8 * KQ), k = e % (8 * KQ);
                unsigned w = 0;
                if (r < steps) {
                    const int8_t *row = raw + off + r * tc;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = 4 * k + i;
                        if (c < 2 * tc && c < Cin) {
                            const int sv = c < tc ? row[c] : -row[c - tc];
                            w |= (unsigned)(sv > 0) << (8 * i);
                        }
                    }
                }
                tile32[e] = w;
            }
        }
        __syncthreads();
        const int ntile = (steps + 15) >> 4;
        cnt = short2_t{0, 0};
        produce(0);
        for (int i = 0; i < ntile; ++i) {
            const int jn = steps - 16 * i < 16 ? steps - 16 * i : 16;
            const int4_t *cw = reinterpret_cast<const int4_t *>(&cur[wv][0][l][0]);
            int4_t in4[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) in4[t4] = cw[64 * t4];
            if (i + 1 < ntile) produce(i + 1);  // (overwrites the slice: after the reads above, in program order)
            if (jn == 16) {
                const short2_t isyn_s = isyn, vmem_s = vmem, cnt_s = cnt;
                vmx = short2_t{-32768, -32768};
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) step_fast(in4[t4][r], t0 + 16 * i + 4 * t4 + r);
                }
                const short2_t fifteen = {15, 15};
                const short2_t below = __builtin_elementwise_sub_sat(vmx, th) >> fifteen;  // all ones: never reached th again
                if (__builtin_expect(__any(xp_i(below) != -1), 0)) {
                    isyn = isyn_s;
                    vmem = vmem_s;
                    cnt = cnt_s;
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) step(in4[t4][r], t0 + 16 * i + 4 * t4 + r);
                    }
                }
            } else {  // the last steps of the signal
                for (int j = 0; j < jn; ++j) {
                    int w_in = 0;
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                        for (int r = 0; r < 4; ++r) w_in = (j == 4 * t4 + r) ? in4[t4][r] : w_in;
                    step(w_in, t0 + 16 * i + j);
                }
            }
        }
        total0 += steps + cnt.x;
        total1 += steps + cnt.y;
    }
    };
    if constexpr (!QUEUE) {
        run_range(0, T);
        if (rate) {
            if (act0) rate[(size_t)b * N + n0] = total0;
            if (act1) rate[(size_t)b * N + n1] = total1;
        }
    } else {
        __shared__ int s_ticket;
        const int nchunks = (T + Lc - 1) / Lc;
        const int nitems = nchunks * ntrials;
        // (one barrier in front of the write keeps the previous ticket's readers ahead of it, one behind publishes the new one;
        //  the value is made a scalar: every branch on it is a scalar branch, no lane can leave a wave early)
        auto next_ticket = [&]() {
            __syncthreads();
            if (tid == 0) s_ticket = atomicAdd(&qctl[0], 1);
            __syncthreads();
            return __builtin_amdgcn_readfirstlane(s_ticket);
        };
        for (int ticket = next_ticket(); ticket < nitems; ticket = next_ticket()) {  // (the counter only grows: every worker leaves)
            const int chunk = ticket / ntrials;
            b = ticket - chunk * ntrials;
            sb = raster + (size_t)b * T * tc;
            int4_t *st = qstate + ((size_t)b * XP_WAVES + wv) * 64 + l;
            if (chunk > 0) {
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(&qctl[XQ_CTL + b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < chunk) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > XQ_SPIN_MAX) {  // never seen; keeps a broken launch from hanging the device
                            atomicExch(&qctl[1], 1);
                            break;
                        }
                    }
                }
                __syncthreads();
                __threadfence();  // acquire for every lane: the state below was written by another workgroup, maybe on another XCD
                const int4_t v = *st;
                isyn = xp_s2(v[0]);
                vmem = xp_s2(v[1]);
                total0 = v[2];
                total1 = v[3];
            } else {
                isyn = short2_t{0, 0};
                vmem = short2_t{0, 0};
                total0 = total1 = 0;
            }
            const int t_lo = chunk * Lc;
            const int t_hi = t_lo + Lc < T ? t_lo + Lc : T;
            run_range(t_lo, t_hi);
            if (chunk + 1 < nchunks) {
                *st = int4_t{xp_i(isyn), xp_i(vmem), total0, total1};
                __threadfence();  // release: the state is visible device-wide before the chunk counter says so
                __syncthreads();
                if (tid == 0) __hip_atomic_store(&qctl[XQ_CTL + b], chunk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // a worker that gave up waiting (never seen) went on from an unwritten state: whatever is finished after that is marked,
                // not returned as plausible counts (the host side raises at its next synchronisation point, XyloNetwork.check)
                const bool broken = __hip_atomic_load(&qctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (act0) rate[(size_t)b * N + n0] = broken ? -1 : total0;
                if (act1) rate[(size_t)b * N + n1] = broken ? -1 : total1;
            }
        }
    }
}

__global__ void xylo_queue_reset_kernel(int *qctl, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) qctl[i] = 0;
}

static int xp_npad(int N) { return ((N + XP_NEUR - 1) / XP_NEUR) * XP_NEUR; }
static int xp_kq(int Cin) { return Cin <= 32 ? 1 : 2; }

static size_t xylo_ws_base_bytes(int Cin, int N)
{
    const int nq = (Cin + 3) / 4;
    size_t bytes = ((size_t)nq * N * sizeof(int) + 255) & ~(size_t)255;
    bytes += 3 * (((size_t)N * 2 + 255) & ~(size_t)255);
    return bytes;
}

size_t xylo_ws_bytes(int Cin, int N)
{
    size_t bytes = xylo_ws_base_bytes(Cin, N);
    if (Cin <= 64) bytes += ((size_t)xp_npad(N) * 32 * xp_kq(Cin) + 255) & ~(size_t)255;  // neuron-major byte matrix (packed form)
    return bytes;
}

// Uploads the packed weights and per-neuron constants into `ws` (synchronises the stream once).
hipError_t xylo_upload(int Cin, const int8_t *W_in_host, int N, const uint8_t *dash_syn_host, const uint8_t *dash_mem_host,
                       const int16_t *thr_host, void *ws, hipStream_t stream)
{
    const int nq = (Cin + 3) / 4;
    if (nq > XY_MAXQ) return hipErrorInvalidValue;
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    int *dW = reinterpret_cast<int *>(base);
    size_t off = ((size_t)nq * N * sizeof(int) + 255) & ~(size_t)255;
    const size_t seg = ((size_t)N * 2 + 255) & ~(size_t)255;
    uint8_t *dds = base + off;
    uint8_t *ddm = base + off + seg;
    short *dth = reinterpret_cast<short *>(base + off + 2 * seg);
    // pack 4 input channels per dword: byte i of word k = W_in[4k + i][g]
    std::vector<int> pk((size_t)nq * N, 0);
    for (int k = 0; k < nq; ++k)
        for (int g = 0; g < N; ++g) {
            unsigned v = 0;
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * k + i;
                const unsigned byte = c < Cin ? (unsigned char)W_in_host[(size_t)c * N + g] : 0u;
                v |= byte << (8 * i);
            }
            pk[(size_t)k * N + g] = (int)v;
        }
    hipError_t e = hipMemcpyAsync(dW, pk.data(), pk.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemcpyAsync(dds, dash_syn_host, (size_t)N, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(ddm, dash_mem_host, (size_t)N, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(dth, thr_host, (size_t)N * 2, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    std::vector<int8_t> wb;
    if (Cin <= 64) {
        // neuron-major bytes for the matrix-core form: Wb[n][c], rows of 32 KQ bytes, neurons padded to whole workgroups
        const int row = 32 * xp_kq(Cin);
        wb.assign((size_t)xp_npad(N) * row, 0);
        for (int c = 0; c < Cin; ++c)
            for (int g = 0; g < N; ++g) wb[(size_t)g * row + c] = W_in_host[(size_t)c * N + g];
        if ((e = hipMemcpyAsync(base + xylo_ws_base_bytes(Cin, N), wb.data(), wb.size(), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    }
    return hipStreamSynchronize(stream);  // `pk` / `wb` are host temporaries
}

// Runs the network whose constants xylo_upload placed in `ws`: no host access, no synchronisation (graph-capturable).
hipError_t launch_xylo_resident(const void *spikes_in, int ternary_C, int B, int T, int Cin, int N, int w_rec, int max_spikes,
                                uint8_t *spikes_out, int32_t *rate, void *ws, hipStream_t stream)
{
    const int nq = (Cin + 3) / 4;
    if (nq > XY_MAXQ) return hipErrorInvalidValue;
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    int *dW = reinterpret_cast<int *>(base);
    size_t off = ((size_t)nq * N * sizeof(int) + 255) & ~(size_t)255;
    const size_t seg = ((size_t)N * 2 + 255) & ~(size_t)255;
    uint8_t *dds = base + off;
    uint8_t *ddm = base + off + seg;
    short *dth = reinterpret_cast<short *>(base + off + 2 * seg);
    const uint8_t *sp = reinterpret_cast<const uint8_t *>(spikes_in);
    if (w_rec != 0 && N > 1024) return hipErrorInvalidValue;
    if (w_rec == 0 && ternary_C > 0 && Cin <= 64 && max_spikes >= 1 && (reinterpret_cast<uintptr_t>(spikes_in) & 15) == 0) {
        // binary events, independent neurons: two neurons per lane, input currents on the matrix cores
        const long *dWb = reinterpret_cast<const long *>(base + xylo_ws_base_bytes(Cin, N));
        // as many waves as hold neurons (N = 360: 3, not 4), the same in every workgroup of the launch
        const int nblk = xp_npad(N) / XP_NEUR;
        const int waves = nblk > 1 ? XP_WAVES : (N + 127) / 128;
        const dim3 pgrid(nblk, B), pblock(waves * 64);
        size_t plds = (size_t)waves * 4 * 64 * 4 * sizeof(int);
        const size_t rawb = (size_t)XP_TT * ternary_C + 48;
        plds = plds > rawb ? plds : rawb;
        const int8_t *rs = reinterpret_cast<const int8_t *>(spikes_in);
#define XP_LAUNCH(KQ, WO)                                                                                                      \
    hipLaunchKernelGGL((xylo_lif_pk_kernel<KQ, WO>), pgrid, pblock, plds, stream, rs, ternary_C, T, Cin, dWb, N, dds, ddm, dth,       \
                       max_spikes, spikes_out, rate, nullptr, nullptr, 0, B)
        if (xp_kq(Cin) == 1) {
            if (spikes_out)
                XP_LAUNCH(1, true);
            else
                XP_LAUNCH(1, false);
        } else {
            if (spikes_out)
                XP_LAUNCH(2, true);
            else
                XP_LAUNCH(2, false);
        }
#undef XP_LAUNCH
        return hipGetLastError();
    }
    // without recurrence the neurons are independent: as few, as full workgroups as possible (N = 360 -> one of 384 threads
    // instead of 256 + 104: every wave of the launch issues the same instructions whether its lanes are used or not)
    const int nblocks = w_rec != 0 ? 1 : (N + 511) / 512;
    const int per = (N + nblocks - 1) / nblocks;
    const dim3 block(((per + 63) / 64) * 64), grid(nblocks, B);
#define XY_LAUNCH(NQ)                                                                                                          \
    do {                                                                                                                       \
        if (w_rec != 0)                                                                                                        \
            hipLaunchKernelGGL((xylo_lif_kernel<true, NQ>), grid, block, 0, stream, sp, ternary_C, T, Cin, dW, nq, N, w_rec, dds, \
                               ddm, dth, max_spikes, spikes_out, rate);                                                        \
        else                                                                                                                   \
            hipLaunchKernelGGL((xylo_lif_kernel<false, NQ>), grid, block, 0, stream, sp, ternary_C, T, Cin, dW, nq, N, 0, dds,    \
                               ddm, dth, max_spikes, spikes_out, rate);                                                        \
    } while (0)
    if (nq <= 2)
        XY_LAUNCH(2);
    else if (nq <= 4)
        XY_LAUNCH(4);
    else if (nq <= 7)
        XY_LAUNCH(7);
    else if (nq <= 8)
        XY_LAUNCH(8);
    else
        XY_LAUNCH(16);
#undef XY_LAUNCH
    return hipGetLastError();
}

// ---- the sweep's form: counts only, ternary raster, no recurrence -> persistent workgroups on a ticket queue -----------------
static bool xylo_sweep_queued(const void *raster, int Cin, int N)
{
    return Cin <= 64 && N <= XP_NEUR && (reinterpret_cast<uintptr_t>(raster) & 15) == 0;
}

size_t xylo_sweep_scratch_bytes(int B)
{
    // [XQ_CTL control ints][B chunk counters] | [B][XP_WAVES][64] int4 hand-over state
    return (((size_t)(XQ_CTL + B) * sizeof(int) + 255) & ~(size_t)255) + (size_t)B * XP_WAVES * 64 * sizeof(int4_t);
}

constexpr int XQ_CHUNK = 2048;  // steps per ticket (16 staged tiles): ~0.3 ms of work against a few microseconds of hand-over

hipError_t launch_xylo_sweep(const int8_t *raster, int ternary_C, int B, int T, int Cin, int N, int max_spikes, int32_t *rate, void *ws,
                             void *scratch, int workers_per_cu, hipStream_t stream)
{
    if (!xylo_sweep_queued(raster, Cin, N) || max_spikes < 1)
        return launch_xylo_resident(raster, ternary_C, B, T, Cin, N, 0, max_spikes, nullptr, rate, ws, stream);
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    const int nq = (Cin + 3) / 4;
    size_t off = ((size_t)nq * N * sizeof(int) + 255) & ~(size_t)255;
    const size_t seg = ((size_t)N * 2 + 255) & ~(size_t)255;
    uint8_t *dds = base + off;
    uint8_t *ddm = base + off + seg;
    short *dth = reinterpret_cast<short *>(base + off + 2 * seg);
    const long *dWb = reinterpret_cast<const long *>(base + xylo_ws_base_bytes(Cin, N));
    int *qctl = reinterpret_cast<int *>(scratch);
    int4_t *qstate = reinterpret_cast<int4_t *>(reinterpret_cast<unsigned char *>(scratch) + (((size_t)(XQ_CTL + B) * sizeof(int) + 255) & ~(size_t)255));
    static int num_cu = 0;  // (one device model per process: the library's code objects are gfx950 only)
    if (num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int waves = (N + 127) / 128;
    const int nchunks = (T + XQ_CHUNK - 1) / XQ_CHUNK;
    const long long nitems = (long long)nchunks * B;
    // four workgroups per CU: with three waves each every SIMD holds three; never more workers than trials (a worker beyond that
    // could only wait for a predecessor)
    int workers = (workers_per_cu > 0 ? workers_per_cu : 4) * num_cu;
    workers = workers > B ? B : workers;
    workers = (long long)workers > nitems ? (int)nitems : workers;
    size_t plds = (size_t)waves * 4 * 64 * 4 * sizeof(int);
    const size_t rawb = (size_t)XP_TT * ternary_C + 48;
    plds = plds > rawb ? plds : rawb;
    hipLaunchKernelGGL(xylo_queue_reset_kernel, dim3((XQ_CTL + B + 255) / 256), dim3(256), 0, stream, qctl, XQ_CTL + B);
    if (xp_kq(Cin) == 1)
        hipLaunchKernelGGL((xylo_lif_pk_kernel<1, false, true>), dim3(workers), dim3(waves * 64), plds, stream, raster, ternary_C, T, Cin, dWb, N,
                           dds, ddm, dth, max_spikes, nullptr, rate, qctl, qstate, XQ_CHUNK, B);
    else
        hipLaunchKernelGGL((xylo_lif_pk_kernel<2, false, true>), dim3(workers), dim3(waves * 64), plds, stream, raster, ternary_C, T, Cin, dWb, N,
                           dds, ddm, dth, max_spikes, nullptr, rate, qctl, qstate, XQ_CHUNK, B);
    return hipGetLastError();
}

hipError_t launch_xylo(const uint8_t *spikes_in, int B, int T, int Cin, const int8_t *W_in_host, int N, int w_rec,
                       const uint8_t *dash_syn_host, const uint8_t *dash_mem_host, const int16_t *thr_host,
                       int max_spikes, uint8_t *spikes_out, int32_t *rate, void *ws, hipStream_t stream)
{
    hipError_t e0 = xylo_upload(Cin, W_in_host, N, dash_syn_host, dash_mem_host, thr_host, ws, stream);
    if (e0 != hipSuccess) return e0;
    return launch_xylo_resident(spikes_in, 0, B, T, Cin, N, w_rec, max_spikes, spikes_out, rate, ws, stream);
}

}  // namespace micloc
