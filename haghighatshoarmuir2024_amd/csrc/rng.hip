// Counter-based random numbers for the throughput-mode Monte-Carlo sweep: Philox-4x32-10 (Salmon et al., "Parallel
// random numbers: as easy as 1, 2, 3", SC'11) + Box-Muller in fp64.
//
// Reference: the sweeps draw `doa = np.random.rand(1)[0] * 2 * np.pi` and add
// `sqrt(mean(sig**2)) / sqrt(snr) * np.random.randn(T, M)` (paper_plots/target_snn_localization.py:452,
// micloc/snn_beamformer.py:270-275) from NumPy's global MT19937 stream.  That stream is sequential and host-only; parity
// runs replay it on the host (sweep.py "parity" mode).  Throughput runs use these kernels instead, so that no B x T x M
// tensor is generated on the host or crosses PCIe; they are validated statistically and bit-for-bit against
// oracle/micloc_oracle.c's restatement of the same generator (integers and uniforms exact, normals to ~1e-15: libm
// vs. device log / sin / cos).
//
// Stream layout: key = seed (64 bit); counter = (pair index, epoch, trial, substream) -- four independent words, so two
// draws share a Philox block only if they agree in all of them.  The uniforms carry the reserved trial word 0xFFFFFFFF
// (PHILOX_UNIFORM_DOMAIN): for one seed they are disjoint from the normals of every trial, substream and epoch.  (Round 2
// added the epoch to the substream word, which made (substream s, epoch e) and (s + e, 0) the same stream and let the DoA
// draws of substream 0 coincide with trial 0's noise.)  The normals of trial b are numbered by element pair: pair i covers
// elements 2i, 2i+1 of the flat [T][M] frame block (z0 = r cos, z1 = r sin); a call covers fewer than 2^32 pairs.
#include "micloc_internal.h"

namespace micloc {

constexpr uint32_t PHILOX_UNIFORM_DOMAIN = 0xFFFFFFFFu;  // trial word of the uniform generator (no trial carries it)

struct Philox4 {
    uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// 53-bit uniforms: [0, 1) and (0, 1]
__device__ __forceinline__ double u53_co(uint32_t lo, uint32_t hi) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * 0x1.0p-53; }
__device__ __forceinline__ double u53_oc(uint32_t lo, uint32_t hi)
{
    return (double)(((((uint64_t)hi << 32) | lo) >> 11) + 1) * 0x1.0p-53;
}

// out[i] = lo + (hi - lo) * u,  u in [0, 1): two per Philox call (words 0-1 and 2-3)
__global__ __launch_bounds__(256) void uniform_kernel(double *__restrict__ out, size_t n, uint32_t k0, uint32_t k1, uint32_t sub,
                                                       const uint32_t *__restrict__ epoch, double lo, double span)
{
    const size_t pair = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (2 * pair >= n) return;
    const uint32_t ep = epoch ? *epoch : 0u;
    const Philox4 r = philox4x32_10((uint32_t)pair, ep, PHILOX_UNIFORM_DOMAIN, sub, k0, k1);
    out[2 * pair] = lo + span * u53_co(r.v[0], r.v[1]);
    if (2 * pair + 1 < n) out[2 * pair + 1] = lo + span * u53_co(r.v[2], r.v[3]);
}

hipError_t launch_uniform(double *out, size_t n, uint64_t seed, uint32_t substream, const uint32_t *epoch, double lo, double hi,
                          hipStream_t stream)
{
    const size_t pairs = (n + 1) / 2;
    hipLaunchKernelGGL(uniform_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, out, n, (uint32_t)seed,
                       (uint32_t)(seed >> 32), substream, epoch, lo, hi - lo);
    return hipGetLastError();
}

// *counter += inc: the device-side step number a captured HIP graph advances between replays (`epoch` of the generators)
__global__ void counter_add_kernel(uint32_t *counter, uint32_t inc) { *counter += inc; }

hipError_t launch_counter_add(uint32_t *counter, uint32_t inc, hipStream_t stream)
{
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, stream, counter, inc);
    return hipGetLastError();
}

// ---- additive white Gaussian noise at a per-trial SNR ----------------------------------------------------------
constexpr int AWGN_BLOCK = 8192;  // elements per workgroup (256 threads x 16 pairs)

// partial[b][blk] = sum of squares of one block, fixed order: per-thread strided sum, then a binary tree in LDS
__global__ __launch_bounds__(256) void sumsq_kernel(const double *__restrict__ x, size_t n, int nblk, double *__restrict__ partial)
{
    __shared__ double red[256];
    const int b = blockIdx.y;
    const size_t lo = (size_t)blockIdx.x * AWGN_BLOCK;
    const double *xb = x + (size_t)b * n;
    double acc = 0.0;
    for (int i = threadIdx.x; i < AWGN_BLOCK; i += 256) {
        const size_t e = lo + i;
        if (e < n) {
            const double v = xb[e];
            acc = __builtin_fma(v, v, acc);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = red[0];
}

// sigma[b] = sqrt(mean(x[b]^2)) / sqrt(10^(snr_db[b] / 10))   (snn_beamformer.py:270-273)
__global__ __launch_bounds__(256) void sigma_kernel(const double *__restrict__ partial, int nblk, size_t n, const double *__restrict__ snr_db,
                                                     double *__restrict__ sigma)
{
    __shared__ double red[256];
    const int b = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[(size_t)b * nblk + i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) sigma[b] = sqrt(red[0] / (double)n) / sqrt(pow(10.0, snr_db[b] / 10.0));
}

__global__ __launch_bounds__(256) void awgn_kernel(double *__restrict__ x, size_t n, const double *__restrict__ sigma, uint32_t k0,
                                                    uint32_t k1, uint32_t sub, const uint32_t *__restrict__ epoch, uint32_t trial0)
{
    const int b = blockIdx.y;
    const uint32_t ep = epoch ? *epoch : 0u;
    double *xb = x + (size_t)b * n;
    const double sg = sigma[b];
    const size_t pair0 = (size_t)blockIdx.x * (AWGN_BLOCK / 2);
#pragma unroll 4
    for (int i = threadIdx.x; i < AWGN_BLOCK / 2; i += 256) {
        const size_t pair = pair0 + i;
        const size_t e = 2 * pair;
        if (e >= n) break;
        const Philox4 r = philox4x32_10((uint32_t)pair, ep, trial0 + (uint32_t)b, sub, k0, k1);
        const double u1 = u53_oc(r.v[0], r.v[1]);
        const double u2 = u53_co(r.v[2], r.v[3]);
        const double rad = sqrt(-2.0 * log(u1));
        const double ang = 6.283185307179586476925286766559 * u2;
        xb[e] = xb[e] + sg * (rad * cos(ang));
        if (e + 1 < n) xb[e + 1] = xb[e + 1] + sg * (rad * sin(ang));
    }
}

size_t awgn_ws_bytes(int B, size_t n)
{
    const size_t nblk = (n + AWGN_BLOCK - 1) / AWGN_BLOCK;
    return (((size_t)B * nblk + (size_t)B) * sizeof(double) + 255) & ~(size_t)255;
}

hipError_t launch_awgn(double *x, int B, size_t n, const double *snr_db, const double *sigma_in, uint64_t seed, uint32_t substream,
                       const uint32_t *epoch, uint32_t trial0, void *ws, hipStream_t stream)
{
    const int nblk = (int)((n + AWGN_BLOCK - 1) / AWGN_BLOCK);
    const double *sigma = sigma_in;
    if (!sigma) {
        double *partial = reinterpret_cast<double *>(ws);
        double *sg = partial + (size_t)B * nblk;
        hipLaunchKernelGGL(sumsq_kernel, dim3(nblk, B), dim3(256), 0, stream, x, n, nblk, partial);
        hipLaunchKernelGGL(sigma_kernel, dim3(B), dim3(256), 0, stream, partial, nblk, n, snr_db, sg);
        sigma = sg;
    }
    hipLaunchKernelGGL(awgn_kernel, dim3(nblk, B), dim3(256), 0, stream, x, n, sigma, (uint32_t)seed, (uint32_t)(seed >> 32),
                       substream, epoch, trial0);
    return hipGetLastError();
}

}  // namespace micloc
