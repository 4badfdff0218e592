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

namespace micloc {

constexpr int XY_TT = 256;   // time steps staged per LDS tile
constexpr int XY_MAXQ = 16;  // up to 64 input channels

__device__ __forceinline__ int xy_sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

__device__ __forceinline__ int xy_decay(int v, int dash)
{
    int dv = v >> dash;
    dv = (dv == 0 && v != 0) ? (v > 0 ? 1 : -1) : dv;
    return v - dv;
}

template <bool REC>
__global__ __launch_bounds__(1024) void xylo_lif_kernel(const uint8_t *__restrict__ spikes_in, int T, int Cin, int nq,
                                                         const int *__restrict__ Wpk /*[nq][N]*/, int N, int w_rec,
                                                         const uint8_t *__restrict__ dash_syn,
                                                         const uint8_t *__restrict__ dash_mem,
                                                         const short *__restrict__ thr, int max_spikes,
                                                         uint8_t *__restrict__ spikes_out, int *__restrict__ rate)
{
    __shared__ int tile[XY_TT][XY_MAXQ];
    __shared__ int wsum[2][16];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    const bool act = g < N;
    int w[XY_MAXQ];
#pragma unroll
    for (int k = 0; k < XY_MAXQ; ++k) w[k] = (act && k < nq) ? Wpk[(size_t)k * N + g] : 0;
    const int ds = act ? dash_syn[g] : 0, dm = act ? dash_mem[g] : 0;
    const int th = act ? thr[g] : 32767;
    int isyn = 0, vmem = 0, total = 0, prev_total = 0;
    const uint8_t *sb = spikes_in + (size_t)b * T * Cin;
    uint8_t *ob = spikes_out ? spikes_out + (size_t)b * T * N : nullptr;
    const int nwaves = blockDim.x >> 6;

    for (int t0 = 0; t0 < T; t0 += XY_TT) {
        const int steps = (T - t0) < XY_TT ? (T - t0) : XY_TT;
        __syncthreads();
        // stage `steps` rows of Cin bytes, zero padded to nq dwords
        uint8_t *tb = reinterpret_cast<uint8_t *>(&tile[0][0]);
        for (int e = threadIdx.x; e < steps * XY_MAXQ * 4; e += blockDim.x) {
            const int r = e / (XY_MAXQ * 4), c = e % (XY_MAXQ * 4);
            tb[e] = c < Cin ? sb[(size_t)(t0 + r) * Cin + c] : 0;
        }
        __syncthreads();
        for (int j = 0; j < steps; ++j) {
            int in = 0;
#pragma unroll
            for (int k = 0; k < XY_MAXQ; ++k)
                if (k < nq) in = __builtin_amdgcn_sdot4(w[k], tile[j][k], in, false);
            int i2 = xy_decay(isyn, ds);
            int v2 = xy_decay(vmem, dm);
            i2 = xy_sat16(i2 + in + (REC ? w_rec * prev_total : 0));
            v2 = xy_sat16(v2 + i2);
            int n = 0;
            if (v2 >= th) {
                n = v2 / th;  // th > 0: subtractive reset until below threshold ...
                n = n < max_spikes ? n : max_spikes;  // ... or until the per-step cap
                v2 -= n * th;
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

hipError_t launch_xylo(const uint8_t *spikes_in, int B, int T, int Cin, const int8_t *W_in_host, int N, int w_rec,
                       const uint8_t *dash_syn_host, const uint8_t *dash_mem_host, const int16_t *thr_host,
                       int max_spikes, uint8_t *spikes_out, int32_t *rate, void *ws, hipStream_t stream)
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
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;  // `pk` is a host temporary
    if (w_rec != 0) {
        if (N > 1024) return hipErrorInvalidValue;
        dim3 block(((N + 63) / 64) * 64), grid(1, B);
        hipLaunchKernelGGL(xylo_lif_kernel<true>, grid, block, 0, stream, spikes_in, T, Cin, nq, dW, N, w_rec, dds, ddm,
                           dth, max_spikes, spikes_out, rate);
    } else {
        dim3 block(256), grid((N + 255) / 256, B);
        hipLaunchKernelGGL(xylo_lif_kernel<false>, grid, block, 0, stream, spikes_in, T, Cin, nq, dW, N, 0, dds, ddm, dth,
                           max_spikes, spikes_out, rate);
    }
    return hipGetLastError();
}

}  // namespace micloc
