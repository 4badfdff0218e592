// STHT kernel for gfx950: in-phase = circular roll by L/2, quadrature = causal FIR with the Hilbert
// kernel (reference: micloc/snn_beamformer.py:325-327, micloc/beamformer.py:281-283).
//
// Mapping (wave64): one block = one trial x one 512-sample time tile x up to 8 microphones; each wave
// owns one microphone, each lane R = 8 consecutive output samples (8 independent fp64 FMA chains).  The
// (tile + halo) x mics frame block is read once from HBM with coalesced loads ([t][mic] rows are
// contiguous) and transposed into per-microphone LDS rows.  The per-lane R-sample sliding window
// lives in registers; every tap costs one 16-byte LDS read (kstep 2) and R v_fma_f64, the tap value
// comes from a scalar load.  Rows are skewed by 16 B every R doubles so the lane stride is
// conflict-free for ds_read_b128.
// R is a compile-time constant (STHT_R).  R = 16 (half the LDS reads per FMA: with R = 8 the 64-lane b128 read of a tap
// holds the CU's LDS pipe for 8 cycles against 32 cycles of FMAs per wave, 14 waves on 4 SIMDs) was built and measured in
// round 3: 1024-sample tiles need 13.7 KB of LDS per microphone row, so a block holds 4 (2) microphones and a CU 8 (10)
// waves instead of 14 -- 0.49 (0.53) ms against 0.44-0.46 ms on the sweep shape.  The kernel is bound by how well staging,
// FIR and stores of co-resident blocks overlap, not by the LDS pipe.
//
// Arithmetic contract (== oracle/micloc_oracle.c oracle_stht): acc = +0; for taps k ascending:
// acc = fma(ker[k], x[t-k], acc); exact-zero taps contribute nothing (skipped when every second tap
// is zero, which is the case for every even-length Hilbert kernel).

#include "micloc_internal.h"

namespace micloc {

constexpr int R = STHT_R;
static_assert(R == 8 || R == 16, "window / skew period");
__device__ __forceinline__ int skew(int q) { return q + 2 * (q / R); }

template <int S, bool WRITE_RE>
__global__ __launch_bounds__(64 * STHT_MAX_MB) void stht_kernel(const double *__restrict__ x,
                                                                 double *__restrict__ h,
                                                                 const double *__restrict__ taps, int ngroups,
                                                                 int klo, int halo, int shift, int T, int M,
                                                                 int Ts, int MB, int rowstride)
{
    extern __shared__ __attribute__((aligned(16))) double Xs[];
    constexpr int U = R / S;
    const int tid = threadIdx.x;
    // XCD-aware placement (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, each with
    // its own 4 MB L2, and consecutive time tiles of a (trial, mic group) row share `halo` input samples -- half of
    // what a tile reads.  Linear id L goes to XCD L % 8; give that XCD a whole row at a time, so the halo is an L2 hit
    // instead of a second trip to HBM (measured: 2 x the unique input bytes fetched without this).
    int tile_x = blockIdx.x, row_yz = blockIdx.y + gridDim.y * blockIdx.z;
    {
        const int ntile = gridDim.x, nrow = gridDim.y * gridDim.z;
        const int L = tile_x + ntile * row_yz;
        const int full = (nrow >> 3) << 3;  // rows in complete groups of 8
        if (L < full * ntile) {
            const int j = L >> 3;
            const int rq = j / ntile;
            tile_x = j - rq * ntile;
            row_yz = 8 * rq + (L & 7);
        }
    }
    const int t0 = tile_x * STHT_TILE;
    const int m0 = (row_yz % gridDim.y) * MB;
    const int b = row_yz / gridDim.y;
    const int N = R + halo + STHT_TILE;
    const int tq0 = t0 - halo - R;  // global time of logical LDS index 0
    const double *xb = x + (size_t)b * T * M;

    // ---- stage (tile + halo) x MB mics, transposed to per-mic rows, zero outside [0, T) ----------
    // Loads are issued in one batch of NB per thread (clamped addresses, so unconditional) before any LDS write: the
    // block pays the HBM latency once per tile instead of once per row.
    {
        const int mm = tid % MB;
        const int m = m0 + mm;
        const int mc = m < M ? m : M - 1;
        double *row = Xs + (size_t)mm * rowstride;
        const int q0 = tid / MB;
        constexpr int NB = R == 8 ? 16 : 24;  // loads in flight per thread: the whole tile + halo of the default shape in one batch
        if (m < M && tq0 >= 0 && tq0 + N <= T && N <= 64 * NB) {
            // interior tile (workgroup-uniform apart from m < M): no clamping, constant strides on both sides.
            // skew(q0 + 64 i) = skew(q0) + (64 + 128 / R) i because 64 is a multiple of R.
            const double *src = xb + (size_t)(tq0 + q0) * M + m;
            double *dst = row + skew(q0);
            double v[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int q = q0 + 64 * i;
                v[i] = src[(size_t)(q < N ? 64 * i : 0) * M];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (q0 + 64 * i < N) dst[(64 + 128 / R) * i] = v[i];
        } else {
            for (int qb = q0; qb < N; qb += 64 * NB) {
                double v[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    int t = tq0 + qb + 64 * i;
                    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
                    v[i] = xb[(size_t)t * M + mc];
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int q = qb + 64 * i;
                    const int t = tq0 + q;
                    if (q < N) row[skew(q)] = (t >= 0 && t < T && m < M) ? v[i] : 0.0;
                }
            }
        }
    }
    __syncthreads();

    const int wv = tid >> 6;
    const int lane = tid & 63;
    const int m = m0 + wv;
    const int tb = t0 + lane * R;
    if (m >= M || tb >= T) return;
    const double *row = Xs + (size_t)wv * rowstride;

    // ---- quadrature: FIR over the sliding register window -------------------------------------------
    double acc[R];
    double w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    int j = lane * R + (halo + R - klo);  // logical index of x[tb - klo]; j % R == 0
    int pj = skew(j);
#pragma unroll
    for (int r = 0; r < R; r += 2) {
        const double2 v2 = *reinterpret_cast<const double2 *>(row + pj + r);
        w[r] = v2.x;
        w[r + 1] = v2.y;
    }
    // taps of group g+1 are fetched (scalar loads) while group g is being accumulated; the table is padded
    // with one extra all-zero group so the look-ahead never reads out of bounds
    double tpn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) tpn[u] = taps[u];
    for (int g = 0; g < ngroups; ++g) {
        pj -= R + 2;  // skew(j - R)
        double tp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) tp[u] = tpn[u];
#pragma unroll
        for (int u = 0; u < U; ++u) tpn[u] = taps[(g + 1) * U + u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double nx[S];
            if (S == 2) {
                // pj and the offset are even and rows are 16-byte aligned: one ds_read_b128
                const double2 v2 = *reinterpret_cast<const double2 *>(row + pj + (R - (u + 1) * S));
                nx[0] = v2.x;
                nx[S - 1] = v2.y;
            } else {
#pragma unroll
                for (int e = 0; e < S; ++e) nx[e] = row[pj + (R - (u + 1) * S + e)];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_fma(tp[u], w[(r - u * S) & (R - 1)], acc[r]);
#pragma unroll
            for (int e = 0; e < S; ++e) w[(R - (u + 1) * S + e) & (R - 1)] = nx[e];
        }
    }

    const int C = 2 * M;
    double *him = h + ((size_t)b * C + M + m) * Ts + tb;
    // (rows are padded to a multiple of 8 samples, a lane covers 16: its second half may lie beyond the row)
#pragma unroll
    for (int r = 0; r < R; r += 2) {
        double2 v2 = make_double2(acc[r], acc[r + 1]);
        if (tb + r < Ts) *reinterpret_cast<double2 *>(him + r) = v2;
    }

    // ---- in-phase: np.roll(x, shift, axis=0) --------------------------------------------------------
    // (skipped in the fused pipelines: the band-pass kernel's loader reads the rolled input frames itself)
    if (!WRITE_RE) return;
    const int sh = shift % T;
    double re[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = tb + r;
        double v = 0.0;
        if (t < T) {
            int src = t - sh;
            if (src < 0) src += T;
            const int q = src - tq0;
            if (q >= 0 && src <= t)
                v = row[skew(q)];
            else
                v = xb[(size_t)src * M + m];
        }
        re[r] = v;
    }
    double *hre = h + ((size_t)b * C + m) * Ts + tb;
#pragma unroll
    for (int r = 0; r < R; r += 2) {
        double2 v2 = make_double2(re[r], re[r + 1]);
        if (tb + r < Ts) *reinterpret_cast<double2 *>(hre + r) = v2;
    }
}

static int stht_rowstride(const SthtTaps &tp)
{
    const int N = STHT_R + tp.halo + STHT_TILE;
    int np = N + 2 * ((N + STHT_R - 1) / STHT_R) + 2;
    np = (np + 1) & ~1;  // keep rows 16-byte aligned
    return np + 2;       // +16 B so consecutive mic rows start on different bank groups
}

// ---- the quadrature FIR on the matrix cores (fused pipelines: quadrature rows only) ---------------------------------------
// On gfx950 the fp64 matrix pipe and the fp64 vector pipe have the same peak, so a FIR gains no flops by moving to
// v_mfma_f64_16x16x4_f64 -- but the vector form above is pinned to ~57 % of it: every tap costs a 16-byte LDS read per lane,
// which keeps the CU's LDS pipe exactly as busy as the FMAs keep the SIMDs.  As a Toeplitz product the same chain reads
// two 8-byte fragments per 1024 multiply-adds.
//   Every second tap of an even-length Hilbert kernel is zero (kstep == 2): y[t] = sum_j g[j] x[t - klo - 2j].  Outputs of one
//   parity, t = 2i + p, see inputs of ONE parity: u_p[i] = sum_j g[j] X_pi[i + o - j] with X_pi[v] = x[2v + pi],
//   pi = (p - klo) mod 2, o = floor((p - klo) / 2) -- two dense FIRs of J taps on de-interleaved data.
//   D[stream][i] += A[stream][k] B[k][i]:  A = X_pi[v_top - 4s - k][stream] (16 streams = 16 rows, data from LDS),
//   B = G[il + 4s + k] with G[idx] = g[idx - 15] (zero outside the kernel: the Toeplitz corners), k-steps s = 0 .. NK - 1 from
//   the newest input backwards.  For output i the k-th product of step s is tap j = il - 15 + 4s + k: the chain runs over the
//   taps in ascending order, products with a zero coefficient in front of and behind it leave the accumulator untouched (it
//   is never -0), and the matrix core accumulates its four products in k order as fused multiply-adds (the LIF product of
//   csrc/beamform.hip rests on the same two facts) -- bit for bit the contract of the kernel above.
// A workgroup = 16 consecutive (trial, microphone) streams x TI = 128 NTW outputs per parity, 8 waves, NTW tiles per wave and
// parity (2 NTW independent accumulator chains per wave, two waves per SIMD: the matrix pipe never waits).  Both parities of
// (stream, i) sit in the same lane: y[2i], y[2i + 1] leave as one 16-byte store, 256 contiguous bytes per stream row.
constexpr int SM_WAVES = 8;
constexpr int SM_THREADS = 64 * SM_WAVES;
constexpr int SM_BAND = 8;  // neighbouring stream groups an XCD walks together when the recording is long (see the kernel)

// NKT: the number of k-steps when it is known at compile time (the 480-tap kernel of the paper: 64), else 0.  With a constant
// trip count the multiply loop unrolls completely and every s_waitcnt counts exactly the reads it needs; inside a loop the
// compiler waits for the operands of the NEXT group as well (it cannot tell the back edge's reads from the new ones).
template <int NTW, int NKT>
__global__ __launch_bounds__(SM_THREADS, 2) void stht_mfma_kernel(const double *__restrict__ x, double *__restrict__ h,
                                                                  const double *__restrict__ taps, int J, int klo, int NK_rt, int T, int M,
                                                                  int Ts, int nstreams)
{
    const int NK = NKT ? NKT : NK_rt;
    typedef double double4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) double Xs[];
    constexpr int TI = SM_WAVES * NTW * 16;  // outputs per parity and workgroup
    const int R = TI + 4 * NK - 16;          // staged rows per input parity
    // ONE staged array: the two output parities are independent problems on different input samples, so the workgroup loads
    // both into registers at once, then stages / multiplies one after the other through the same LDS rows -- half the LDS,
    // two workgroups per CU (one multiplies while the other waits for its loads).
    double *XS = Xs, *G = Xs + (size_t)R * 16;
    const int tid = threadIdx.x;
    // XCD-aware order (xcd_walk, micloc_internal.h).  Two kinds of workgroups read the same input lines: consecutive time tiles of
    // a stream group (the halo, half of what a tile reads) and NEIGHBOURING stream groups (16 streams end in the middle of a trial
    // whenever 16 % M != 0).  An XCD walks a contiguous range of groups, tile by tile: both kinds of sharing meet in one L2.
    // Dealing the groups round-robin (neighbours on different XCDs; until round 4) fetched 451 MB per launch on the sweep shape
    // for 296 MB of input, this walk 308 MB (FETCH_SIZE, tools/dev/stht_fetch_ab.sh).
    // Inside its range an XCD holds 64 workgroups at a time.  Group by group that is 64 / ntile neighbouring groups: fine for short
    // recordings (T = 4799: ten tiles, six groups), but a 94-tile recording has ONE group resident and its neighbours' shared lines are
    // long gone when they come (T = 48 000: 5.1 GB fetched for 3.0 GB of input).  Long recordings are therefore walked in bands of
    // SM_BAND groups, tile by tile across the band -- ~8 consecutive tiles of 8 neighbouring groups resident: 3.4 GB.  (Measured band
    // widths 1 / 2 / 4 / 8 / 16: 5.1 / 4.0 / 3.6 / 3.4 / 3.9 GB at T = 48 000, 315 / 319 / 330 / 336 / 377 MB at T = 4799.)
    const int ntile = gridDim.x, ngrp = gridDim.y;
    const int vid = xcd_walk(blockIdx.x + ntile * blockIdx.y, ntile * ngrp);
    const int bw = ntile <= 16 ? 1 : SM_BAND;
    const int band = vid / (bw * ntile), rem = vid - band * (bw * ntile);
    const int inband = ngrp - band * bw < bw ? ngrp - band * bw : bw;  // (the last band may be narrower)
    const int tile = rem / inband, grp = band * bw + (rem - tile * inband);
    const int I0 = tile * TI;
    // input parity and offset of the two output parities
    const int e0 = -klo, e1 = 1 - klo;
    const int pi0 = e0 & 1, pi1 = e1 & 1;
    const int o0 = (e0 - pi0) / 2, o1 = (e1 - pi1) / 2;

    for (int idx = tid; idx < 4 * NK + 16; idx += SM_THREADS) {
        const int j = idx - 15;
        G[idx] = (j >= 0 && j < J) ? taps[j] : 0.0;
    }
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, lc = l & 15, q = l >> 4;
    const int irel = wv * NTW * 16;                  // first output of this wave, relative to I0
    const bool active = 2 * (I0 + irel) < Ts;        // (wave-uniform)
    double4_t acc0[NTW], acc1[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
        acc0[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
        acc1[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
    }
    // the Toeplitz product of one parity: row of k-step s, lane group q, tile tt = irel + 16 tt + 4 NK - 1 - 4 s - q
    // The operands of a group of k-steps are requested while the previous group is being multiplied (two register sets): issued
    // right before their own MFMAs, the reads left the matrix pipe idle for an LDS round trip per group (12 % of the kernel by
    // ablation).
    auto multiply = [&](double4_t *acc) {
        constexpr int GK = 2;  // k-steps per group: 2 NTW MFMAs (>= 256 cycles of matrix pipe) cover the next group's LDS round trip
        const double *ap0 = XS + (size_t)(irel + 4 * NK - 1 - q) * 16 + lc;
        const double *bp0 = G + lc + q;
        auto fetch = [&](int g, double (&av)[GK][NTW], double (&bv)[GK]) {
            const double *pa = ap0 - (size_t)(64 * GK) * (g + 1), *pb = bp0 + 4 * GK * g;
#pragma unroll
            for (int u = 0; u < GK; ++u) {
                bv[u] = pb[4 * u];
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) av[u][tt] = pa[(GK - u) * 64 + tt * 256];
            }
        };
        auto mult = [&](const double (&av)[GK][NTW], const double (&bv)[GK]) {
#pragma unroll
            for (int u = 0; u < GK; ++u)
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][tt], bv[u], acc[tt], 0, 0, 0);
        };
        const int NG = NK / GK;
        if constexpr (NKT > 0) {
            static_assert(NKT % (2 * GK) == 0, "whole pairs of groups");
            double a0[GK][NTW], b0[GK], a1[GK][NTW], b1[GK];
            fetch(0, a0, b0);
#pragma unroll
            for (int g = 0; g < NKT / GK; g += 2) {
                fetch(g + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the reads back in front of their MFMAs)
                mult(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < NKT / GK) fetch(g + 2, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mult(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (NG > 0) {
            double a0[GK][NTW], b0[GK], a1[GK][NTW], b1[GK];
            fetch(0, a0, b0);
            for (int g = 0;; g += 2) {
                if (g + 1 < NG) fetch(g + 1, a1, b1);
                mult(a0, b0);
                if (g + 1 >= NG) break;
                if (g + 2 < NG) fetch(g + 2, a0, b0);
                mult(a1, b1);
                if (g + 2 >= NG) break;
            }
        }
        const double *ap = ap0 - (size_t)(64 * GK) * NG, *bp = bp0 + 4 * GK * NG;
        for (int s = GK * NG; s < NK; ++s) {
            const double bn = *bp;
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[tt * 256], bn, acc[tt], 0, 0, 0);
            ap -= 64;
            bp += 4;
        }
    };

    const int sl = tid & 15;
    const int sg = grp * 16 + sl;
    const bool valid = sg < nstreams;
    const int bs = valid ? sg / M : 0, ms = valid ? sg - bs * M : 0;
    const double *xs = x + (size_t)bs * T * M + ms;
    constexpr int RP = SM_THREADS / 16;  // rows per pass
    constexpr int NB = 16;               // loads in flight per thread and parity: the whole tile of the default shape in one batch
    const int tau00 = 2 * (I0 + o0 - (4 * NK - 16)) + pi0, tau01 = 2 * (I0 + o1 - (4 * NK - 16)) + pi1;  // input time of staged row 0
    if (R <= RP * NB) {
        // both parities in flight at once; the second one waits in registers while the first is multiplied
        double v0[NB], v1[NB];
        const int r0 = tid >> 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int ta = tau00 + 2 * (r0 + RP * i), tb = tau01 + 2 * (r0 + RP * i);
            ta = ta < 0 ? 0 : (ta >= T ? T - 1 : ta);
            tb = tb < 0 ? 0 : (tb >= T ? T - 1 : tb);
            v0[i] = xs[(size_t)ta * M];
            v1[i] = xs[(size_t)tb * M];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int rho = r0 + RP * i, ta = tau00 + 2 * rho;
            if (rho < R) XS[rho * 16 + sl] = (valid && ta >= 0 && ta < T) ? v0[i] : 0.0;
        }
        __syncthreads();
        if (active) multiply(acc0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int rho = r0 + RP * i, tb = tau01 + 2 * rho;
            if (rho < R) XS[rho * 16 + sl] = (valid && tb >= 0 && tb < T) ? v1[i] : 0.0;
        }
        __syncthreads();
        if (active) multiply(acc1);
    } else {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int tau0 = par ? tau01 : tau00;
            if (par) __syncthreads();
            for (int r0 = tid >> 4; r0 < R; r0 += RP * NB) {
                double v[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    int ta = tau0 + 2 * (r0 + RP * i);
                    ta = ta < 0 ? 0 : (ta >= T ? T - 1 : ta);
                    v[i] = xs[(size_t)ta * M];
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int rho = r0 + RP * i, ta = tau0 + 2 * rho;
                    if (rho < R) XS[rho * 16 + sl] = (valid && ta >= 0 && ta < T) ? v[i] : 0.0;
                }
            }
            __syncthreads();
            if (active) multiply(par ? acc1 : acc0);
        }
    }
    if (!active) return;

    const int C = 2 * M;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int sigma = grp * 16 + q + 4 * r;
        if (sigma >= nstreams) continue;
        const int b = sigma / M, m = sigma - b * M;
        double *row = h + ((size_t)b * C + M + m) * Ts;
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
            const int t = 2 * (I0 + irel + 16 * tt + lc);
            if (t < Ts) *reinterpret_cast<double2 *>(row + t) = make_double2(acc0[tt][r], acc1[tt][r]);
        }
    }
}

// ---- the walking form of the matrix-core kernel ---------------------------------------------------------------------------
// Both output parities see the SAME staged row index: u_p[i] = sum_j g[j] x[2 (i - j) + p - klo], so with row r of the staged
// array holding the frame pair x[2 r - klo], x[2 r - klo + 1] of a stream, the two parities of a stream are two COLUMNS of one
// array and one Toeplitz product serves both.  A workgroup is then 8 streams x 2 parities (the 16 matrix rows) and one staged
// array holds everything a tile needs -- so a workgroup can WALK consecutive tiles of its streams: the newest 4 NK - 16 rows of a
// tile are the halo of the next one (copied inside LDS), only TI new rows per tile come from memory, and they are requested before
// the multiply of the tile in front of them.  Column c of the array is matrix row c: parity (c >> 2) & 1 of stream
// (c & 3) + 4 (c >> 3), so that lane (q, lc) ends with both parities of streams q and q + 4 and stores 16 bytes per stream as before.
// The multiply is the one of the kernel above (same k order, same operands): bit for bit the same results.
// NCP: 16-byte pairs of the halo a thread copies per tile (4: halos up to 256 rows, 8: up to 512)
template <int NTW, int NKT, int NCP = 4>
__global__ __launch_bounds__(SM_THREADS, 4) void stht_walk_kernel(const double *__restrict__ x, double *__restrict__ h,
                                                                  const double *__restrict__ taps, int J, int klo, int NK_rt, int T, int M,
                                                                  int Ts, int nstreams, int ntile, int tpw)
{
    const int NK = NKT ? NKT : NK_rt;
    typedef double double4_t __attribute__((ext_vector_type(4)));
    typedef double double2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) double Xs[];
    constexpr int TI = SM_WAVES * NTW * 16;  // outputs per parity and tile
    const int HR = 4 * NK - 16;              // halo rows
    const int R = TI + HR;
    double *XS = Xs, *G = Xs + (size_t)R * 16;
    const int tid = threadIdx.x;
    // XCD-aware order as above; the unit is (group of 8 streams, segment of tpw tiles)
    const int nseg = gridDim.x, ngrp = gridDim.y;
    const int vid = xcd_walk(blockIdx.x + nseg * blockIdx.y, nseg * ngrp);
    const int bw = nseg <= 16 ? 1 : SM_BAND;
    const int band = vid / (bw * nseg), rem = vid - band * (bw * nseg);
    const int inband = ngrp - band * bw < bw ? ngrp - band * bw : bw;
    const int seg = rem / inband, grp = band * bw + (rem - seg * inband);
    const int t_lo = seg * tpw, t_hi = (t_lo + tpw) < ntile ? (t_lo + tpw) : ntile;

    for (int idx = tid; idx < 4 * NK + 16; idx += SM_THREADS) {
        const int j = idx - 15;
        G[idx] = (j >= 0 && j < J) ? taps[j] : 0.0;
    }
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, lc = l & 15, q = l >> 4;
    const int irel = wv * NTW * 16;
    auto multiply = [&](double4_t *acc, int lq, int llc) {  // (lane coordinates passed in: see the opaque move in the tile loop)
        constexpr int GK = 2;
        // (the index of the LOWEST operand row passes through an opaque move: the compiler otherwise re-bases every read on the newest
        //  row and the immediate offsets become negative, i.e. one vector add per read)
        int aoff = (irel + 4 * NK - 1 - lq) * 16 + llc - 64 * ((NKT ? NKT : 1) - 1);
        if (NKT) asm volatile("" : "+v"(aoff));
        const double *alo = XS + aoff;
        const double *ap0 = alo + 64 * ((NKT ? NKT : 1) - 1);
        const double *bp0 = G + llc + lq;
        // (compile-time trip count: the operand rows are addressed upwards from the LOWEST one -- LDS instructions take unsigned
        //  immediate offsets only, and a pointer that walks down from the newest row cost one vector add per read: 1.3 vector
        //  instructions per matrix instruction in the 124-k-step form, every one of them an interruption of the matrix pipe)
        auto fetch = [&](int g, double (&av)[GK][NTW], double (&bv)[GK]) {
            const double *pa = ap0 - (size_t)(64 * GK) * (g + 1), *pb = bp0 + 4 * GK * g;
#pragma unroll
            for (int u = 0; u < GK; ++u) {
                bv[u] = pb[4 * u];
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt)
                    av[u][tt] = NKT ? alo[64 * (NKT - 1 - (GK * g + u)) + tt * 256] : pa[(GK - u) * 64 + tt * 256];
            }
        };
        auto mult = [&](const double (&av)[GK][NTW], const double (&bv)[GK]) {
#pragma unroll
            for (int u = 0; u < GK; ++u)
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][tt], bv[u], acc[tt], 0, 0, 0);
        };
        const int NG = NK / GK;
        if constexpr (NKT > 0) {
            static_assert(NKT % (2 * GK) == 0, "whole pairs of groups");
            double a0[GK][NTW], b0[GK], a1[GK][NTW], b1[GK];
            fetch(0, a0, b0);
#pragma unroll
            for (int g = 0; g < NKT / GK; g += 2) {
                fetch(g + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mult(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < NKT / GK) fetch(g + 2, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mult(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (NG > 0) {
            double a0[GK][NTW], b0[GK], a1[GK][NTW], b1[GK];
            fetch(0, a0, b0);
            for (int g = 0;; g += 2) {
                if (g + 1 < NG) fetch(g + 1, a1, b1);
                mult(a0, b0);
                if (g + 1 >= NG) break;
                if (g + 2 < NG) fetch(g + 2, a0, b0);
                mult(a1, b1);
                if (g + 2 >= NG) break;
            }
        }
        const double *ap = ap0 - (size_t)(64 * GK) * NG, *bp = bp0 + 4 * GK * NG;
        for (int s = GK * NG; s < NK; ++s) {
            const double bn = *bp;
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[tt * 256], bn, acc[tt], 0, 0, 0);
            ap -= 64;
            bp += 4;
        }
    };

    // this thread's column of the staged array
    const int c = tid & 15, par = (c >> 2) & 1, sc = (c & 3) + 4 * (c >> 3);
    const int sg = grp * 8 + sc;
    const bool valid = sg < nstreams;
    const int bs = valid ? sg / M : 0, ms = valid ? sg - bs * M : 0;
    const double *xs = x + (size_t)bs * T * M + ms;
    constexpr int RP = SM_THREADS / 16;  // rows per pass
    const int r0 = tid >> 4;
    auto time_of = [&](int I0, int rho) { return 2 * (I0 - HR + rho) + par - klo; };  // input time of staged row rho of the tile at I0
    auto load = [&](int I0, int rho) {
        int ta = time_of(I0, rho);
        ta = ta < 0 ? 0 : (ta >= T ? T - 1 : ta);
        return xs[(size_t)ta * M];
    };
    auto put = [&](int I0, int rho, double v) {
        const int ta = time_of(I0, rho);
        XS[rho * 16 + c] = (valid && ta >= 0 && ta < T) ? v : 0.0;
    };
    // the first tile of the walk: the whole window
    {
        const int I0 = t_lo * TI;
        constexpr int NB = 16;
        for (int rb = r0; rb < R; rb += RP * NB) {
            double v[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) v[i] = load(I0, rb + RP * i < R ? rb + RP * i : R - 1);
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (rb + RP * i < R) put(I0, rb + RP * i, v[i]);
        }
    }
    __syncthreads();
    constexpr int NEWB = TI / RP;  // new rows per thread and tile (8)
    const int C2 = 2 * M;
    for (int t = t_lo; t < t_hi; ++t) {
        const int I0 = t * TI;
        const bool more = t + 1 < t_hi;
        // (the row and column of this thread pass through an opaque move once per tile: everything derived from them -- eight
        // addresses, eight edge predicates -- is then recomputed per tile instead of being kept in registers across the multiply,
        // where the loop-invariant copies cost 92 bytes of scratch)
        int r0 = tid >> 4, c = tid & 15, lq = q, llc = lc;
        asm volatile("" : "+v"(r0), "+v"(c), "+v"(lq), "+v"(llc));
        const int par = (c >> 2) & 1;
        auto time_of = [&](int I0_, int rho) { return 2 * (I0_ - HR + rho) + par - klo; };
        auto load = [&](int I0_, int rho) {
            int ta = time_of(I0_, rho);
            ta = ta < 0 ? 0 : (ta >= T ? T - 1 : ta);
            return xs[(size_t)ta * M];
        };
        auto put = [&](int I0_, int rho, double v) {
            const int ta = time_of(I0_, rho);
            XS[rho * 16 + c] = (valid && ta >= 0 && ta < T) ? v : 0.0;
        };
        double vnew[NEWB];
        // the next tile's new rows: in flight during the multiply.  Interior tiles (all of a thread's eight frames inside the recording)
        // are one pointer and constant strides; the others clamp frame by frame
        const int tn0 = time_of(I0 + TI, HR + r0);
        const bool inside = tn0 >= 0 && tn0 + 2 * RP * (NEWB - 1) < T;
        if (more) {
            if (inside) {
                const double *pl = xs + (size_t)tn0 * M;
                const size_t st = (size_t)2 * RP * M;
#pragma unroll
                for (int i = 0; i < NEWB; ++i) vnew[i] = pl[st * i];
            } else {
#pragma unroll
                for (int i = 0; i < NEWB; ++i) vnew[i] = load(I0 + TI, HR + r0 + RP * i);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool active = 2 * (I0 + irel) < Ts;  // (wave-uniform)
        if (active) {
            double4_t acc[NTW];
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) acc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
            multiply(acc, lq, llc);
            asm volatile("" : "+v"(lq), "+v"(llc));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int sigma = grp * 8 + lq + 4 * k;
                if (sigma >= nstreams) continue;
                const int b = sigma / M, m = sigma - b * M;
                double *row = h + ((size_t)b * C2 + M + m) * Ts;
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) {
                    const int to = 2 * (I0 + irel + 16 * tt + llc);
                    if (to < Ts) *reinterpret_cast<double2 *>(row + to) = make_double2(acc[tt][2 * k], acc[tt][2 * k + 1]);
                }
            }
        }
        if (!more) break;
        __builtin_amdgcn_sched_barrier(0);
        // slide: the newest HR rows become the halo (source rows [TI, R), destination rows [0, HR)).  The two ranges OVERLAP whenever
        // HR > TI (the 480-tap instantiation: HR = 480, TI = 128); the copy is correct because every thread reads ALL of its NCP pairs
        // into registers before the barrier below and writes them behind it -- do not turn it into an LDS-to-LDS copy or drop the
        // barrier.  (The launcher guarantees NCP * SM_THREADS >= 8 * HR: every pair of the halo has a thread.)  The new rows overwrite
        // [HR, R) once every wave has finished reading.
        const int ncopy = HR * 8;  // 16-byte pairs: at most NCP per thread (the launcher's choice)
        double2_t hc[NCP];
#pragma unroll
        for (int i = 0; i < NCP; ++i) {
            const int e = tid + SM_THREADS * i;
            hc[i] = e < ncopy ? *reinterpret_cast<const double2_t *>(XS + (size_t)TI * 16 + 2 * e) : double2_t{0.0, 0.0};
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCP; ++i) {
            const int e = tid + SM_THREADS * i;
            if (e < ncopy) *reinterpret_cast<double2_t *>(XS + 2 * e) = hc[i];
        }
        if (inside) {
            double *pd = XS + (size_t)(HR + r0) * 16 + c;
#pragma unroll
            for (int i = 0; i < NEWB; ++i) pd[(size_t)RP * 16 * i] = valid ? vnew[i] : 0.0;
        } else {
#pragma unroll
            for (int i = 0; i < NEWB; ++i) put(I0 + TI, HR + r0 + RP * i, vnew[i]);
        }
        __syncthreads();
    }
}

// in-phase rows beside the matrix-core form: h[b][m][t] = x[b][(t - shift) mod T][m] (np.roll), zero in the row padding.
// 64 frames x M microphones per workgroup through an LDS tile: row-major reads, planar writes, both coalesced.
__global__ __launch_bounds__(256) void stht_inphase_kernel(const double *__restrict__ x, double *__restrict__ h, int shift, int T, int M, int Ts)
{
    extern __shared__ __attribute__((aligned(16))) double Xs[];
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    const int sh = shift % T;
    const double *xb = x + (size_t)b * T * M;
    for (int e = threadIdx.x; e < 64 * M; e += 256) {
        const int tl = e / M, m = e - tl * M;
        const int t = t0 + tl;
        double v = 0.0;
        if (t < T) {
            int src = t - sh;
            if (src < 0) src += T;
            v = xb[(size_t)src * M + m];
        }
        Xs[tl * M + m] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * M; e += 256) {
        const int m = e >> 6, tl = e & 63;
        const int t = t0 + tl;
        if (t < Ts) h[((size_t)b * 2 * M + m) * Ts + t] = Xs[tl * M + m];
    }
}

// LDS of the matrix-core form with NTW tiles per wave and parity; NK k-steps for J compact taps
static size_t stht_mfma_lds(int J, int NTW)
{
    const int NK = (J + 15 + 3) / 4;
    const size_t R = (size_t)SM_WAVES * NTW * 16 + 4 * NK - 16;
    return (R * 16 + 4 * (size_t)NK + 16) * sizeof(double);
}

size_t stht_lds_bytes(const SthtTaps &tp, int M)
{
    const int MB = M < STHT_MAX_MB ? M : STHT_MAX_MB;
    return (size_t)MB * stht_rowstride(tp) * sizeof(double);
}

hipError_t launch_stht(const SthtTaps &tp, const double *x, double *h, int B, int T, int M, int Ts,
                       hipStream_t stream, bool write_re)
{
    // every second tap zero: the matrix-core form, if its tile fits LDS (never in the stht_valu variant build), in-phase rows by a copy kernel
    if (tp.kstep == 2 && tp.ngroups > 0 && !VARIANT_STHT_VECTOR_FORM && ((size_t)B * M + 15) / 16 <= 65535 && B <= 65535) {
        const int J = tp.ngroups * (STHT_R / tp.kstep);  // compact taps incl. the zero padding of the last group
        const int NK = (J + 15 + 3) / 4;
        // two workgroups per CU if the tile allows it
        const int ntw = (VARIANT_STHT_WIDE_TWO_TILES && stht_mfma_lds(J, 2) > 80 * 1024 && stht_mfma_lds(J, 2) <= 160 * 1024) ? 2 :
                        stht_mfma_lds(J, 2) <= 80 * 1024 ? 2 : (stht_mfma_lds(J, 1) <= 80 * 1024 ? 1 : (stht_mfma_lds(J, 2) <= 160 * 1024 ? 2 : (stht_mfma_lds(J, 1) <= 160 * 1024 ? 1 : 0)));
        if (ntw) {
            const int TI = SM_WAVES * ntw * 16;
            const int nstreams = B * M;
            dim3 grid((Ts / 2 + TI - 1) / TI, (nstreams + 15) / 16);
            const size_t lds = stht_mfma_lds(J, ntw);
            if (!VARIANT_STHT_ONE_TILE && 4 * NK - 16 <= 512 && (nstreams + 7) / 8 <= 65535) {
                // the walking form (halos up to 256 rows: four 16-byte copies per thread).  Tiles per workgroup: as many as leave
                // one workgroup per slot of the chip (512), at most 16 -- a single trial still spreads over its ten tiles, the
                // sweep's 963 groups walk their ten tiles each.  Measured on the sweep shape (step, ms): 3 / 4 / 5 / 10 tiles per
                // workgroup 1.545 / 1.483 / 1.475 / 1.466 against 1.50-1.51 with one tile per workgroup; T = 48 000: 16 / 32 / 64
                // tiles 3.18 / 3.25 / 4.15 ms against 3.30.
                const int ntile = (Ts / 2 + TI - 1) / TI, ngrp8 = (nstreams + 7) / 8;
                int tpw = (int)(((long long)ntile * ngrp8) / 512);
                tpw = tpw < 1 ? 1 : (tpw > 16 ? 16 : tpw);
                tpw = tpw > ntile ? ntile : tpw;
                const bool wide = 4 * NK - 16 > 256;  // (J = 480, the 96 kHz kernel: 480 halo rows, one tile per wave)
                // the slide copies the halo through registers, NCP 16-byte pairs per thread: every pair needs a thread
                if ((4 * NK - 16) * 8 > (wide ? 8 : 4) * SM_THREADS) return hipErrorInvalidValue;
                auto kw = ntw == 2 ? (NK == 64 ? &stht_walk_kernel<2, 64, 4> : (wide ? &stht_walk_kernel<2, 0, 8> : &stht_walk_kernel<2, 0, 4>))
                                   : (wide ? (NK == 124 ? &stht_walk_kernel<1, 124, 8> : &stht_walk_kernel<1, 0, 8>) : &stht_walk_kernel<1, 0, 4>);
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kw, dim3((ntile + tpw - 1) / tpw, ngrp8), dim3(SM_THREADS), lds, stream, x, h, tp.taps, J, tp.klo, NK, T, M,
                                   Ts, nstreams, ntile, tpw);
            } else {
            auto k = ntw == 2 ? (NK == 64 ? &stht_mfma_kernel<2, 64> : &stht_mfma_kernel<2, 0>) : &stht_mfma_kernel<1, 0>;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k, grid, dim3(SM_THREADS), lds, stream, x, h, tp.taps, J, tp.klo, NK, T, M, Ts, nstreams);
            }
            if (write_re)
                hipLaunchKernelGGL(stht_inphase_kernel, dim3((Ts + 63) / 64, B), dim3(256), (size_t)64 * M * sizeof(double), stream, x, h,
                                   tp.shift, T, M, Ts);
            return hipGetLastError();
        }
    }
    const int MB = M < STHT_MAX_MB ? M : STHT_MAX_MB;
    const int rowstride = stht_rowstride(tp);
    const size_t lds = (size_t)MB * rowstride * sizeof(double);
    dim3 grid((T + STHT_TILE - 1) / STHT_TILE, (M + MB - 1) / MB, B);
    dim3 block(64 * MB);
#define STHT_LAUNCH(SS, WR)                                                                                        \
    do {                                                                                                           \
        auto k = &stht_kernel<SS, WR>;                                                                             \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),                                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL(k, grid, block, lds, stream, x, h, tp.taps, tp.ngroups, tp.klo, tp.halo, tp.shift, T,  \
                           M, Ts, MB, rowstride);                                                                  \
    } while (0)
    if (tp.kstep == 2) {
        if (write_re)
            STHT_LAUNCH(2, true);
        else
            STHT_LAUNCH(2, false);
    } else {
        if (write_re)
            STHT_LAUNCH(1, true);
        else
            STHT_LAUNCH(1, false);
    }
#undef STHT_LAUNCH
    return hipGetLastError();
}

}  // namespace micloc
