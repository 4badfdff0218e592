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
        // (three-input xor: one v_bitop3_b32, truth table 0x96, instead of two v_xor_b32)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
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

// log(u) for a normal u > 0 (Box-Muller calls it on (0, 1]), < 1 ulp: u = 2^e m with m in [sqrt(1/2), sqrt(2)), f = m - 1,
// s = f / (2 + f), log(1 + f) = 2 atanh(s) = f - f^2/2 + s (f^2/2 + R(s^2)) with the classic degree-7 minimax R (the fdlibm
// coefficients), e ln 2 added in two parts.  About half the instructions of the device library's log, whose double-double
// reduction this range does not need; the noise kernels are bound by their instruction count.
__device__ __forceinline__ double log_pos(double u)
{
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    constexpr double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                     L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                     L7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(u);  // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(u);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0, dk = (double)e;
    // s = f / (2 + f): reciprocal + one Newton step, quotient + one correction
    const double den = 2.0 + f;
    double rc = __builtin_amdgcn_rcp(den);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    double sq = f * rc;
    sq = __builtin_fma(__builtin_fma(-den, sq, f), rc, sq);
    const double z = sq * sq, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, L6, L4), L2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, L7, L5), L3), L1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * LN2_HI - ((hfsq - __builtin_fma(sq, hfsq + R, dk * LN2_LO)) - f);
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

// Order of the sum of squares of one block of AWGN_BLOCK flat elements (rows t of M microphones, element e = t M + m): the
// block's rows are cut into chunks of 64, a TASK is (chunk, microphone), task j = chunk M + m goes to wave j mod 4 of the
// workgroup, and lane l of that wave takes row 64 chunk + l.  A thread adds the squares of its elements in task order; then a
// binary tree over the 256 threads in LDS.  A wave that holds 64 consecutive time steps of ONE microphone reads 64 consecutive
// template rows in the fused synthesis below (no LDS bank conflicts, one delay and one row offset for the whole wave); the plain
// kernel uses the same order so that both give the same sigma bit for bit.
struct TimeMap {
    int M, nel, ntask, off0, r0;
    int j, c, m;  // current task of this wave, its chunk and microphone (wave-uniform: scalar registers)
    int laneM;    // lane * M
    int ebase;    // 64 c M + m - off0
    __device__ __forceinline__ TimeMap(size_t lo, size_t hi, int M_) : M(M_)
    {
        r0 = (int)(lo / (size_t)M);
        const int r1 = (int)((hi - 1) / (size_t)M);
        off0 = (int)(lo - (size_t)r0 * M);  // the block starts at microphone off0 of row r0
        nel = (int)(hi - lo);
        ntask = ((r1 - r0 + 64) / 64) * M;
        j = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        c = j / M;
        m = j - c * M;
        laneM = (int)(threadIdx.x & 63) * M;
        ebase = 64 * c * M + m - off0;
    }
    __device__ __forceinline__ bool more() const { return j < ntask; }
    __device__ __forceinline__ void next()
    {
        j += 4;
        m += 4;
        ebase += 4;
        while (m >= M) {
            m -= M;
            ++c;
            ebase += 63 * M;
        }
    }
    // row of this lane in the current task, relative to the block's first row
    __device__ __forceinline__ int row() const { return 64 * c + (int)(threadIdx.x & 63); }
    // element of this lane relative to the block's first; live (inside the block) iff 0 <= element < nel
    __device__ __forceinline__ int element() const { return ebase + laneM; }
    __device__ __forceinline__ bool live(int er) const { return (unsigned)er < (unsigned)nel; }
};

__device__ __forceinline__ double block_tree_sum(double acc, double *red)
{
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    return red[0];
}

// partial[b][blk] = sum of squares of one block
__global__ __launch_bounds__(256) void sumsq_kernel(const double *__restrict__ x, size_t n, int M, int nblk, double *__restrict__ partial)
{
    __shared__ double red[256];
    const int b = blockIdx.y;
    const size_t lo = (size_t)blockIdx.x * AWGN_BLOCK;
    const size_t hi = lo + AWGN_BLOCK < n ? lo + AWGN_BLOCK : n;
    const double *xb = x + (size_t)b * n;
    double acc = 0.0;
    const double *xblk = xb + lo;
    for (TimeMap tk(lo, hi, M); tk.more(); tk.next()) {
        const int er = tk.element();
        if (tk.live(er)) {
            const double v = xblk[er];
            acc = __builtin_fma(v, v, acc);
        }
    }
    const double tot = block_tree_sum(acc, red);
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = tot;
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
        const double rad = sqrt(-2.0 * log_pos(u1));
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

hipError_t launch_awgn(double *x, int B, size_t n, int M, const double *snr_db, const double *sigma_in, uint64_t seed, uint32_t substream,
                       const uint32_t *epoch, uint32_t trial0, void *ws, hipStream_t stream)
{
    const int nblk = (int)((n + AWGN_BLOCK - 1) / AWGN_BLOCK);
    const double *sigma = sigma_in;
    if (!sigma) {
        double *partial = reinterpret_cast<double *>(ws);
        double *sg = partial + (size_t)B * nblk;
        hipLaunchKernelGGL(sumsq_kernel, dim3(nblk, B), dim3(256), 0, stream, x, n, M, nblk, partial);
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
constexpr int SF_WIN = 1280;  // staged template rows (3 x 10 KB)

struct SynthWindow {
    const double *xp, *fp, *sl;  // LDS
    int w0, wn;                  // first staged row, number of staged rows (0: nothing staged)
};

__device__ __forceinline__ SynthWindow stage_template(const SynthArgs &a, const double *dl, bool cached, double shift, size_t lo, size_t n,
                                                      double *xs, double *fs_, double *ss, int *ired, int block = AWGN_BLOCK, int cap = SF_WIN)
{
    SynthWindow w{xs, fs_, ss, 0, 0};
    if (!cached) return w;
    // largest |argument shift| in samples over the trial's K x M delays (a short table: every thread scans it, no reduction)
    double dm = 0.0;
    const bool scan_all = a.K * a.M <= 64;
    for (int e = scan_all ? 0 : threadIdx.x; e < a.K * a.M; e += scan_all ? 1 : 256) {
        const double d = a.mode == 0 ? (a.shift ? dl[e] - shift : dl[e]) : dl[e];
        dm = fabs(d) > dm ? fabs(d) : dm;
    }
    int js = (int)(dm * a.inv_step) + 3;
    js = dm * a.inv_step < 1.0e6 ? js : 2 * SF_WIN;  // (absurd delays: nothing staged)
    if (!scan_all) {
        ired[threadIdx.x] = js;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) ired[threadIdx.x] = ired[threadIdx.x + h] > ired[threadIdx.x] ? ired[threadIdx.x + h] : ired[threadIdx.x];
            __syncthreads();
        }
        js = ired[0];
    }
    const int jd = js;
    const size_t last = (lo + block < n ? lo + block : n) - 1;
    const int t_lo = (int)(lo / (size_t)a.M), t_hi = (int)(last / (size_t)a.M);
    int w0 = t_lo - jd, w1 = t_hi + jd;  // rows [w0, w1]
    w0 = w0 < 0 ? 0 : w0;
    w1 = w1 > a.T - 1 ? a.T - 1 : w1;
    const int wn = w1 - w0 + 1;
    if (wn > cap) return w;
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

// One target at constant delays, the template window in LDS, tasks of TimeMap: the 64 lanes of a wave hold 64 consecutive time
// steps of ONE microphone, and sample t interpolates in the template row t + k with one integer k per microphone (its delay in
// whole grid steps) -- on the sweep's uniform grid always, elsewhere for as long as it lasts.  np.interp's bracket search
// becomes a check: read the rows t + k, t + k + 1 (consecutive LDS words across the wave), interpolate, VERIFY
// xp[q] <= x < xp[q + 1].  A lane that fails (the clamped head of a trial, the template's ends, a grid that is not uniform, a
// k that was guessed one row low) searches the window like interp_bracket does -- or calls it, outside the window -- and
// leaves the corrected k in the table.  Same bracket, same arithmetic, same bits as interp_one.
// kt[m]: row offsets, set up by time_tasks_init.  sink(live, element - lo, value) once per lane and task, in task order.
template <bool MODE0>
__device__ __forceinline__ double synth_arg(double tt, double d, double x0)
{
    if (MODE0) {
        const double x = tt - d;
        return x < x0 ? x0 : x;
    }
    return tt + d;
}

__device__ __forceinline__ void time_tasks_init(const SynthArgs &a, const SynthWindow &w, const double *dl, double shift, double x0, size_t lo, int *kt)
{
    const int r0 = (int)(lo / (size_t)a.M);
    const double tt = w.xp[r0 - w.w0];
    for (int m = threadIdx.x; m < a.M; m += 256) {
        const double d = a.mode == 0 && a.shift ? dl[m] - shift : dl[m];
        const double x = a.mode == 0 ? synth_arg<true>(tt, d, x0) : synth_arg<false>(tt, d, x0);
        int j = (int)((x - x0) * a.inv_step);  // the uniform-grid guess (right, or one row low when x sits on a grid point)
        j = j < 0 ? 0 : (j > a.T - 1 ? a.T - 1 : j);
        kt[m] = j - r0;
    }
    __syncthreads();
}

template <bool MODE0, class Sink>
__device__ __forceinline__ void time_tasks(const SynthArgs &a, const SynthWindow &w, const double *dl, double shift, double x0, size_t lo, size_t hi,
                                           int *kt, Sink &&sink)
{
    constexpr int U = 4;  // tasks in flight per wave: their LDS round trips overlap
    TimeMap tk(lo, hi, a.M);
    const int il0 = tk.r0 - w.w0;  // window row of the block's first row
    const bool shifted = MODE0 && a.shift;
    while (tk.more()) {
        int mv[U], erv[U], ilv[U], qv[U];
        bool livev[U], badv[U], insidev[U];
        double dv[U], xv[U], rv[U];
        int kv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool on = tk.more();  // (wave-uniform)
            mv[u] = on ? tk.m : 0;
            erv[u] = tk.element();
            livev[u] = on && tk.live(erv[u]);
            int il = il0 + tk.row();
            ilv[u] = il > w.wn - 1 ? w.wn - 1 : il;  // (lanes past the block's last row)
            tk.next();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dv[u] = dl[mv[u]];
            kv[u] = kt[mv[u]];
            xv[u] = w.xp[ilv[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u] = synth_arg<MODE0>(xv[u], shifted ? dv[u] - shift : dv[u], x0);
            qv[u] = ilv[u] + kv[u];
            insidev[u] = (unsigned)qv[u] < (unsigned)(w.wn - 1);  // rows q, q + 1 staged
        }
        double xav[U], xbv[U], fqv[U], sqv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int qc = insidev[u] ? qv[u] : 0;
            xav[u] = w.xp[qc];
            xbv[u] = w.xp[qc + 1];
            fqv[u] = w.fp[qc];
            sqv[u] = w.sl[qc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(sqv[u]));  // (all rows read before the first use, none behind the xa == x test)
        bool anybad = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double r = sqv[u] * (xv[u] - xav[u]) + fqv[u];
            rv[u] = (xav[u] == xv[u]) ? fqv[u] : r;
            badv[u] = livev[u] && !(insidev[u] && xav[u] <= xv[u] && xv[u] < xbv[u]);
            anybad = anybad || badv[u];
        }
        if (__any(anybad)) {
            for (int u = 0; u < U; ++u)
                if (badv[u]) {
                    const double x = xv[u];
                    if (x < x0) {
                        rv[u] = a.sig[0];
                    } else if (w.xp[0] <= x && x < w.xp[w.wn - 1]) {
                        int qs = qv[u] < 0 ? 0 : (qv[u] > w.wn - 2 ? w.wn - 2 : qv[u]);
                        while (qs > 0 && w.xp[qs] > x) --qs;
                        while (qs < w.wn - 2 && w.xp[qs + 1] <= x) ++qs;
                        const double xs = w.xp[qs];
                        rv[u] = (xs == x) ? w.fp[qs] : w.sl[qs] * (x - xs) + w.fp[qs];
                        kt[mv[u]] = qs - ilv[u];
                    } else {
                        const int j = interp_bracket(a.time, a.T, x, x0, a.inv_step);
                        rv[u] = interp_at(a.time, a.sig, a.slopes, a.T, x, j);
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) sink(livev[u], erv[u], rv[u]);
    }
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
    __shared__ int kt[SIMPLE ? 512 : 1];
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
    const size_t hi = lo + AWGN_BLOCK < n ? lo + AWGN_BLOCK : n;
    double acc = 0.0;
    // the order of sumsq_kernel (TimeMap)
    if (SIMPLE && w.wn > 0) {  // (workgroup-uniform)
        time_tasks_init(a, w, dl, shift, x0, lo, kt);
        auto add = [&](bool live, int, double v) {
            if (live) acc = __builtin_fma(v, v, acc);
        };
        if (a.mode == 0)
            time_tasks<true>(a, w, dl, shift, x0, lo, hi, kt, add);
        else
            time_tasks<false>(a, w, dl, shift, x0, lo, hi, kt, add);
    } else {
        for (TimeMap tk(lo, hi, a.M); tk.more(); tk.next()) {
            if (tk.live(tk.element())) {
                const int t = tk.r0 + tk.row();
                double v;
                if (CONST_DOA && w.wn > 0)  // (workgroup-uniform)
                    v = synth_sample_win<SIMPLE>(a, w, dl, b, t, tk.m, x0, shift);
                else
                    v = synth_sample(a, CONST_DOA ? dl : nullptr, b, t, tk.m, x0, shift);
                acc = __builtin_fma(v, v, acc);
            }
        }
    }
    const double tot = block_tree_sum(acc, red);
    if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = tot;
}

// Pass 2.  MODE 2 (one target, constant DoA, no gains -- the Monte-Carlo sweep) works on blocks of SA2_BLOCK flat elements in
// two phases: the waves first take 64 time steps of one microphone at a time (TimeMap / time_tasks, as in pass 1) and leave the
// block's clean samples in LDS, then turn to the flat pair order of awgn_kernel, draw the normals and store s + sigma z.  The
// other modes interpolate per sample in the pair loop.
constexpr int SA2_BLOCK = 2048;  // elements per workgroup in MODE 2 (16 KB of samples in LDS)
constexpr int SA2_WIN = 512;     // staged template rows in MODE 2 (3 x 4 KB)

template <int MODE>
__global__ __launch_bounds__(256) void synth_awgn_kernel(SynthArgs a, const double *__restrict__ sigma, uint32_t k0, uint32_t k1, uint32_t sub,
                                                          const uint32_t *__restrict__ epoch, uint32_t trial0)
{
    constexpr bool CONST_DOA = MODE >= 1;
    constexpr bool SIMPLE = MODE == 2;
    constexpr int BLOCK = SIMPLE ? SA2_BLOCK : AWGN_BLOCK;
    constexpr int WIN = SIMPLE ? SA2_WIN : SF_WIN;
    __shared__ double dl[512];
    __shared__ double xs[CONST_DOA ? WIN : 1], fs_[CONST_DOA ? WIN : 1], ss[CONST_DOA ? WIN : 1];
    __shared__ double2 tile[SIMPLE ? SA2_BLOCK / 2 : 1];
    __shared__ int kt[SIMPLE ? 512 : 1];
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
    const size_t pair0 = (size_t)blockIdx.x * (BLOCK / 2);
    const size_t lo = 2 * pair0;
    SynthWindow w{xs, fs_, ss, 0, 0};
    if constexpr (CONST_DOA) w = stage_template(a, dl, true, shift, lo, n, xs, fs_, ss, ired, BLOCK, WIN);
    bool staged = false;
    if constexpr (SIMPLE) {
        if (w.wn > 0) {  // (workgroup-uniform)
            const size_t hi = lo + BLOCK < n ? lo + BLOCK : n;
            double *flat = reinterpret_cast<double *>(tile);
            time_tasks_init(a, w, dl, shift, x0, lo, kt);
            auto put = [&](bool live, int er, double v) {
                if (live) flat[er] = v;
            };
            if (a.mode == 0)
                time_tasks<true>(a, w, dl, shift, x0, lo, hi, kt, put);
            else
                time_tasks<false>(a, w, dl, shift, x0, lo, hi, kt, put);
            __syncthreads();
            staged = true;
        }
    }
    FlatTM tm(2 * (pair0 + threadIdx.x), a.M);
    const int dt = 512 / a.M, dm = 512 % a.M;
    for (int i = threadIdx.x; i < BLOCK / 2; i += 256) {
        const size_t pair = pair0 + i;
        const size_t e = 2 * pair;
        if (e >= n) break;
        double s0, s1 = 0.0;
        if (SIMPLE && staged) {
            const double2 sv = tile[i];
            s0 = sv.x;
            s1 = sv.y;
        } else {
            FlatTM t1 = tm;
            t1.advance(0, 1, a.M);
            if (CONST_DOA && w.wn > 0) {  // (workgroup-uniform)
                s0 = synth_sample_win<SIMPLE>(a, w, dl, b, tm.t, tm.m, x0, shift);
                if (e + 1 < n) s1 = synth_sample_win<SIMPLE>(a, w, dl, b, t1.t, t1.m, x0, shift);
            } else {
                s0 = synth_sample(a, CONST_DOA ? dl : nullptr, b, tm.t, tm.m, x0, shift);
                if (e + 1 < n) s1 = synth_sample(a, CONST_DOA ? dl : nullptr, b, t1.t, t1.m, x0, shift);
            }
            tm.advance(dt, dm, a.M);
        }
        const Philox4 r = philox4x32_10((uint32_t)pair, ep, trial0 + (uint32_t)b, sub, k0, k1);
        const double u1 = u53_oc(r.v[0], r.v[1]);
        const double u2 = u53_co(r.v[2], r.v[3]);
        const double rad = sqrt(-2.0 * log_pos(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        xb[e] = s0 + sg * (rad * cs);
        if (e + 1 < n) xb[e + 1] = s1 + sg * (rad * sn);
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
        const int nblk2 = (MD) == 2 ? (int)((n + SA2_BLOCK - 1) / SA2_BLOCK) : nblk;                                                        \
        hipLaunchKernelGGL(synth_awgn_kernel<MD>, dim3(nblk2, a.B), dim3(256), 0, stream, a, sg, (uint32_t)seed, (uint32_t)(seed >> 32),   \
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
