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
#include "synth_dev.h"

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
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);  // cos / sin of 2 pi u2 without the product's rounding and without a pi reduction
        xb[e] = xb[e] + sg * (rad * cs);
        if (e + 1 < n) xb[e + 1] = xb[e + 1] + sg * (rad * sn);
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

// ---- synthesis + noise in one go (the input side of a throughput-mode Monte-Carlo step) ------------------------------------
// x[b] = s[b] + sigma_b N(0, 1),  s = the synthesised array signal (synth.hip),  sigma_b = sqrt(mean(s[b]^2)) / sqrt(snr_b).
// Done with the kernels above this costs three passes over the B x T x M tensor besides the one that has to happen (store s;
// read it for the sum of squares; read it again to add the noise).  Here the signal is never stored without its noise:
//   pass 1  synth_sumsq_kernel   recomputes s and reduces s^2 per block of AWGN_BLOCK elements -- the order of sumsq_kernel, so
//                                sigma is bit-identical to the unfused path's;
//   pass 2  synth_awgn_kernel    recomputes s, draws the normals of awgn_kernel (same counters) and stores s + sigma z once.
// The synthesis is a handful of instructions per sample next to Box-Muller's log / sincos, so computing it twice is cheaper than
// one round trip through HBM.  Same bits as synth_targets_kernel followed by awgn_kernel.
// (t, m) of flat element e advance with e by fixed steps: no division in the loops.
struct FlatTM {
    int t, m;
    __device__ __forceinline__ FlatTM(size_t e, int M) : t((int)(e / (size_t)M)), m((int)(e - (size_t)t * M)) {}
    __device__ __forceinline__ void advance(int dt, int dm, int M)
    {
        t += dt;
        m += dm;
        if (m >= M) {
            m -= M;
            ++t;
        }
    }
};

// The template rows a block of AWGN_BLOCK flat elements can touch, staged in LDS: the block covers the time steps
// [t_lo, t_hi] and a constant-DoA trial shifts them by at most max |delay| -- for the paper's 4.5 cm array 13 samples.  With the
// rows in LDS a sample costs about 25 instructions and no global load (np.interp's bracket search runs on LDS; the original
// kernel spends most of its time on a 64-bit division and five dependent global loads per sample).  Samples whose bracket
// falls outside the staged window (a wider array than the window allows, a non-uniform grid) take the global-memory path:
// same arithmetic, same bits either way.
constexpr int SF_WIN = 1536;  // staged template rows (3 x 12 KB)

struct SynthWindow {
    const double *xp, *fp, *sl;  // LDS
    int w0, wn;                  // first staged row, number of staged rows (0: nothing staged)
};

__device__ __forceinline__ SynthWindow stage_template(const SynthArgs &a, const double *dl, bool cached, double shift, size_t lo, size_t n,
                                                      double *xs, double *fs_, double *ss, int *ired)
{
    SynthWindow w{xs, fs_, ss, 0, 0};
    if (!cached) return w;
    // largest |argument shift| in samples over the trial's K x M delays
    double dm = 0.0;
    for (int e = threadIdx.x; e < a.K * a.M; e += 256) {
        const double d = a.mode == 0 ? (a.shift ? dl[e] - shift : dl[e]) : dl[e];
        dm = fabs(d) > dm ? fabs(d) : dm;
    }
    int js = (int)(dm * a.inv_step) + 3;
    js = dm * a.inv_step < 1.0e6 ? js : SF_WIN;  // (absurd delays: nothing staged)
    ired[threadIdx.x] = js;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) ired[threadIdx.x] = ired[threadIdx.x + h] > ired[threadIdx.x] ? ired[threadIdx.x + h] : ired[threadIdx.x];
        __syncthreads();
    }
    const int jd = ired[0];
    const size_t last = (lo + AWGN_BLOCK < n ? lo + AWGN_BLOCK : n) - 1;
    const int t_lo = (int)(lo / (size_t)a.M), t_hi = (int)(last / (size_t)a.M);
    int w0 = t_lo - jd, w1 = t_hi + jd;  // rows [w0, w1]
    w0 = w0 < 0 ? 0 : w0;
    w1 = w1 > a.T - 1 ? a.T - 1 : w1;
    const int wn = w1 - w0 + 1;
    if (wn > SF_WIN) return w;
    for (int e = threadIdx.x; e < wn; e += 256) {
        xs[e] = a.time[w0 + e];
        fs_[e] = a.sig[w0 + e];
        ss[e] = w0 + e < a.T - 1 ? a.slopes[w0 + e] : 0.0;
    }
    __syncthreads();
    w.w0 = w0;
    w.wn = wn;
    return w;
}

// synth_sample for a constant-DoA trial with the template window in LDS.  np.interp's bracket search (two data-dependent
// loops, each trip a dependent read) becomes straight-line code: on the nominally uniform grid the truncating guess j is the
// bracket or one row low, so the rows j .. j + 2 are read together, the bracket q in {j, j + 1} is selected and VERIFIED
// (xp[q] <= x < xp[q + 1]); anything else -- the window's edge, the ends of the template, a grid that is not uniform -- takes
// the original routine.  Same bracket, same arithmetic, same bits.
// SIMPLE: one target, no per-sample gain (the throughput-mode sweep): no loop over targets
template <bool SIMPLE>
__device__ __forceinline__ double synth_sample_win(const SynthArgs &a, const SynthWindow &w, const double *__restrict__ dl, int b, int t, int m,
                                                   double x0, double shift)
{
    const double tt = w.xp[t - w.w0];  // (callers take this routine only with a staged window: an LDS-or-global select here becomes a FLAT load per sample)
    double acc = 0.0;
    const int K = SIMPLE ? 1 : a.K;
    for (int k = 0; k < K; ++k) {
        double d = dl[k * a.M + m];
        double x;
        if (a.mode == 0) {
            if (a.shift) d = d - shift;
            x = tt - d;
            x = x < x0 ? x0 : x;
        } else {
            x = tt + d;
        }
        int j = (int)((x - x0) * a.inv_step);
        j = j < 0 ? 0 : (j > a.T - 1 ? a.T - 1 : j);
        const int jl = j - w.w0;
        // The guess truncates, so when it is wrong it is one row LOW (on the sweep's grid 3 % of the samples: a microphone whose
        // delay is zero sits exactly on the grid points) -- and a wave takes the slow routine if ANY lane needs it: rows j .. j + 2
        // are read together, the bracket q = j or j + 1 selected and verified, its value and slope read from there.
        const bool inside = jl >= 0 && jl + 2 <= w.wn - 1 && !(x < x0);  // rows j .. j + 2 staged (hence q + 1 <= T - 1)
        const int jc = inside ? jl : 0;
        const double xa = w.xp[jc], xb = w.xp[jc + 1], xc = w.xp[jc + 2];
        const bool up = xb <= x;
        const int q = jc + (up ? 1 : 0);
        const double xq = up ? xb : xa, xq1 = up ? xc : xb;
        const double fq = w.fp[q], sq = w.sl[q];
        double r = (xq == x) ? fq : sq * (x - xq) + fq;
        if (__builtin_expect(!(inside && xq <= x && x < xq1), 0)) r = interp_one(a.time, a.sig, a.slopes, a.T, x, x0, a.inv_step);
        if (SIMPLE) return r;
        if (a.gain) r = a.gain[((size_t)b * a.K + k) * a.T + t] * r;
        acc = (a.K == 1) ? r : acc + r;
    }
    return acc;
}

// The straight-line part of synth_sample_win<true> alone: value, argument and whether the bracket verified.  Callers keep
// several samples of a lane in flight (their LDS round trips overlap) and run the slow routine after the batch.
__device__ __forceinline__ double synth_fast(const SynthArgs &a, const SynthWindow &w, const double *__restrict__ dl, int t, int m, double x0,
                                             double shift, double &x_out, bool &ok)
{
    const double tt = w.xp[t - w.w0];
    double d = dl[m];
    double x;
    if (a.mode == 0) {
        if (a.shift) d = d - shift;
        x = tt - d;
        x = x < x0 ? x0 : x;
    } else {
        x = tt + d;
    }
    int j = (int)((x - x0) * a.inv_step);
    j = j < 0 ? 0 : (j > a.T - 1 ? a.T - 1 : j);
    const int jl = j - w.w0;
    const bool inside = jl >= 0 && jl + 2 <= w.wn - 1 && !(x < x0);
    const int jc = inside ? jl : 0;
    const double xa = w.xp[jc], xb = w.xp[jc + 1], xc = w.xp[jc + 2];
    const bool up = xb <= x;
    const int q = jc + (up ? 1 : 0);
    const double xq = up ? xb : xa, xq1 = up ? xc : xb;
    const double fq = w.fp[q], sq = w.sl[q];
    x_out = x;
    ok = inside && xq <= x && x < xq1;
    return (xq == x) ? fq : sq * (x - xq) + fq;
}

// CONST_DOA: the K x M delays of a trial do not depend on time (and fit the LDS table): template window + straight-line
// interpolation.  Otherwise (moving DoAs) the general per-sample routine with its in-loop cos.
template <int MODE>  // 0: moving DoAs; 1: constant DoAs; 2: constant DoAs, one target, no gain
__global__ __launch_bounds__(256) void synth_sumsq_kernel(SynthArgs a, int nblk, double *__restrict__ partial)
{
    constexpr bool CONST_DOA = MODE >= 1;
    constexpr bool SIMPLE = MODE == 2;
    __shared__ double dl[512];
    __shared__ double red[256];
    __shared__ double xs[CONST_DOA ? SF_WIN : 1], fs_[CONST_DOA ? SF_WIN : 1], ss[CONST_DOA ? SF_WIN : 1];
    const int b = blockIdx.y;
    constexpr bool cached = CONST_DOA;
    if (cached) {
        // (the launcher has put the trial's K x M delays into a table: no cos -- and none of its registers -- in this kernel)
        for (int e = threadIdx.x; e < a.K * a.M; e += 256) dl[e] = a.delays[(size_t)b * a.K * a.M + e];
        __syncthreads();
    }
    const size_t n = (size_t)a.T * a.M;
    const size_t lo = (size_t)blockIdx.x * AWGN_BLOCK;
    const double x0 = a.time[0];
    const double shift = a.shift ? a.shift[b] : 0.0;
    SynthWindow w{xs, fs_, ss, 0, 0};
    if constexpr (CONST_DOA) w = stage_template(a, dl, true, shift, lo, n, xs, fs_, ss, reinterpret_cast<int *>(red));
    __syncthreads();
    FlatTM tm(lo + threadIdx.x, a.M);
    const int dt = 256 / a.M, dm = 256 % a.M;
    double acc = 0.0;
    if (SIMPLE && w.wn > 0) {  // (workgroup-uniform)
        // four samples of a lane in flight; the per-thread sum keeps its order
        static_assert(AWGN_BLOCK % 1024 == 0, "batches of four strided samples");
        for (int i = threadIdx.x; i < AWGN_BLOCK; i += 1024) {
            double xv[4], rv[4];
            bool okv[4], live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                live[u] = lo + i + 256 * u < n;
                okv[u] = true;
                xv[u] = 0.0;
                rv[u] = live[u] ? synth_fast(a, w, dl, tm.t, tm.m, x0, shift, xv[u], okv[u]) : 0.0;
                tm.advance(dt, dm, a.M);
            }
            if (__any(!(okv[0] && okv[1] && okv[2] && okv[3]))) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (!okv[u]) rv[u] = interp_one(a.time, a.sig, a.slopes, a.T, xv[u], x0, a.inv_step);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (live[u]) acc = __builtin_fma(rv[u], rv[u], acc);
        }
    } else {
        for (int i = threadIdx.x; i < AWGN_BLOCK; i += 256) {
            if (lo + i < n) {
                double v;
                if (CONST_DOA && w.wn > 0)  // (workgroup-uniform)
                    v = synth_sample_win<SIMPLE>(a, w, dl, b, tm.t, tm.m, x0, shift);
                else
                    v = synth_sample(a, CONST_DOA ? dl : nullptr, b, tm.t, tm.m, x0, shift);
                acc = __builtin_fma(v, v, acc);
            }
            tm.advance(dt, dm, a.M);
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

template <int MODE>
__global__ __launch_bounds__(256) void synth_awgn_kernel(SynthArgs a, const double *__restrict__ sigma, uint32_t k0, uint32_t k1, uint32_t sub,
                                                          const uint32_t *__restrict__ epoch, uint32_t trial0)
{
    constexpr bool CONST_DOA = MODE >= 1;
    constexpr bool SIMPLE = MODE == 2;
    __shared__ double dl[512];
    __shared__ double xs[CONST_DOA ? SF_WIN : 1], fs_[CONST_DOA ? SF_WIN : 1], ss[CONST_DOA ? SF_WIN : 1];
    __shared__ int ired[256];
    const int b = blockIdx.y;
    constexpr bool cached = CONST_DOA;
    if (cached) {
        for (int e = threadIdx.x; e < a.K * a.M; e += 256) dl[e] = a.delays[(size_t)b * a.K * a.M + e];
        __syncthreads();
    }
    const uint32_t ep = epoch ? *epoch : 0u;
    const size_t n = (size_t)a.T * a.M;
    double *xb = a.x + (size_t)b * n;
    const double sg = sigma[b];
    const double x0 = a.time[0];
    const double shift = a.shift ? a.shift[b] : 0.0;
    const size_t pair0 = (size_t)blockIdx.x * (AWGN_BLOCK / 2);
    SynthWindow w{xs, fs_, ss, 0, 0};
    if constexpr (CONST_DOA) w = stage_template(a, dl, true, shift, 2 * pair0, n, xs, fs_, ss, ired);
    FlatTM tm(2 * (pair0 + threadIdx.x), a.M);
    const int dt = 512 / a.M, dm = 512 % a.M;
    for (int i = threadIdx.x; i < AWGN_BLOCK / 2; i += 256) {
        const size_t pair = pair0 + i;
        const size_t e = 2 * pair;
        if (e >= n) break;
        FlatTM t1 = tm;
        t1.advance(0, 1, a.M);
        double s0, s1 = 0.0;
        if (CONST_DOA && w.wn > 0) {  // (workgroup-uniform)
            s0 = synth_sample_win<SIMPLE>(a, w, dl, b, tm.t, tm.m, x0, shift);
            if (e + 1 < n) s1 = synth_sample_win<SIMPLE>(a, w, dl, b, t1.t, t1.m, x0, shift);
        } else {
            s0 = synth_sample(a, CONST_DOA ? dl : nullptr, b, tm.t, tm.m, x0, shift);
            if (e + 1 < n) s1 = synth_sample(a, CONST_DOA ? dl : nullptr, b, t1.t, t1.m, x0, shift);
        }
        const Philox4 r = philox4x32_10((uint32_t)pair, ep, trial0 + (uint32_t)b, sub, k0, k1);
        const double u1 = u53_oc(r.v[0], r.v[1]);
        const double u2 = u53_co(r.v[2], r.v[3]);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        xb[e] = s0 + sg * (rad * cs);
        if (e + 1 < n) xb[e + 1] = s1 + sg * (rad * sn);
        tm.advance(dt, dm, a.M);
    }
}

// table[b][k][m] = the constant-DoA delays of every trial, from the DoAs and the geometry (the cos of mic_delay, once per entry)
__global__ __launch_bounds__(256) void delay_table_kernel(SynthArgs a, double *__restrict__ table)
{
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < a.K * a.M; e += 256) table[(size_t)b * a.K * a.M + e] = mic_delay(a, b, e / a.M, 0, e % a.M);
}

size_t synth_awgn_ws_bytes(int B, size_t n, int K, int M)
{
    return awgn_ws_bytes(B, n) + (((size_t)B * K * M * sizeof(double) + 255) & ~(size_t)255);
}

hipError_t launch_synth_awgn(const SynthArgs &a_in, const double *snr_db, uint64_t seed, uint32_t substream, const uint32_t *epoch,
                             uint32_t trial0, void *ws, hipStream_t stream)
{
    SynthArgs a = a_in;
    const size_t n = (size_t)a.T * a.M;
    const int nblk = (int)((n + AWGN_BLOCK - 1) / AWGN_BLOCK);
    double *partial = reinterpret_cast<double *>(ws);
    double *sg = partial + (size_t)a.B * nblk;
    const bool const_doa = !a.moving && a.K * a.M <= 512;
    if (const_doa && !a.delays) {
        double *table = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(ws) + awgn_ws_bytes(a.B, n));
        hipLaunchKernelGGL(delay_table_kernel, dim3(a.B), dim3(256), 0, stream, a, table);
        a.delays = table;
    }
    const int mode = !const_doa ? 0 : ((a.K == 1 && !a.gain) ? 2 : 1);
#define SA_LAUNCH(MD)                                                                                                                      \
    do {                                                                                                                                   \
        hipLaunchKernelGGL(synth_sumsq_kernel<MD>, dim3(nblk, a.B), dim3(256), 0, stream, a, nblk, partial);                               \
        hipLaunchKernelGGL(sigma_kernel, dim3(a.B), dim3(256), 0, stream, partial, nblk, n, snr_db, sg);                                   \
        hipLaunchKernelGGL(synth_awgn_kernel<MD>, dim3(nblk, a.B), dim3(256), 0, stream, a, sg, (uint32_t)seed, (uint32_t)(seed >> 32),    \
                           substream, epoch, trial0);                                                                                      \
    } while (0)
    if (mode == 2)
        SA_LAUNCH(2);
    else if (mode == 1)
        SA_LAUNCH(1);
    else
        SA_LAUNCH(0);
#undef SA_LAUNCH
    return hipGetLastError();
}

}  // namespace micloc
