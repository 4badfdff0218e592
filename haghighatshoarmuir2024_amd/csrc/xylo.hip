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
// Mapping: lane = hidden neuron, workgroup = one trial x up to 1024 neurons, sequential over time (integer
// state recurrences).  The 28 input channels are packed 4 x int8 per dword and contracted with v_dot4_i32_i8;
// input spike rows are staged 256 steps at a time in LDS and read as wave-uniform (broadcast) dwords.
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

size_t xylo_ws_bytes(int Cin, int N)
{
    const int nq = (Cin + 3) / 4;
    size_t bytes = ((size_t)nq * N * sizeof(int) + 255) & ~(size_t)255;
    bytes += 3 * (((size_t)N * 2 + 255) & ~(size_t)255);
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
    return hipStreamSynchronize(stream);  // `pk` is a host temporary
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

hipError_t launch_xylo(const uint8_t *spikes_in, int B, int T, int Cin, const int8_t *W_in_host, int N, int w_rec,
                       const uint8_t *dash_syn_host, const uint8_t *dash_mem_host, const int16_t *thr_host,
                       int max_spikes, uint8_t *spikes_out, int32_t *rate, void *ws, hipStream_t stream)
{
    hipError_t e0 = xylo_upload(Cin, W_in_host, N, dash_syn_host, dash_mem_host, thr_host, ws, stream);
    if (e0 != hipSuccess) return e0;
    return launch_xylo_resident(spikes_in, 0, B, T, Cin, N, w_rec, max_spikes, spikes_out, rate, ws, stream);
}

}  // namespace micloc
