// Alpha-kernel ("LIF") FIR + beamforming + power for gfx950, on the fp64 matrix cores.
// Reference: micloc/snn_beamformer.py:364-368 (vmem = lfilter(nir,[1],spikes); vmem @ bf_mat),
// micloc/beamformer.py:290 (sig @ bf_mat.conj()), paper_plots/target_snn_localization.py:462-464
// (power = mean |y|^2, argmax).
//
// Everything here is v_mfma_f64_16x16x4_f64 (D[16x16] += A[16x4] B[4x16]; lane l holds A[l&15][l>>4],
// B[l>>4][l&15] and D[(l>>4)+4r][l&15], r = 0..3):
//   1. LIF as a Toeplitz product, transposed:  V^T[c][t] = sum_tau S^T[c][tau] * N^T[tau][t],
//      N^T[tau][t] = nir[t - tau].  A = int8 spikes from the LDS tile (converted to fp64), B = a
//      zero-padded lookup of nir.  The result lands as lane l -> V[t = l&15][c = (l>>4)+4r], which is
//      *exactly* the A-operand fragment of k-step r of the next product, so the membrane signal never
//      leaves registers.
//   2. Beamforming  Y[t][g] = sum_c V[t][c] W[c][g]  with W fragments read from LDS (or L2 when the
//      matrix is too large), 4 independent time tiles per wave in flight.
//   3. power: each lane squares its 4 results and keeps a running sum per DoA column; lanes sharing a
//      column are combined with two xor-shuffles, waves through LDS, time chunks by a second tiny
//      kernel in a fixed order (deterministic, no atomics).  The T x G product is only written to
//      HBM when the caller asks for it (API parity with apply_to_signal).
//
// Three kernels share these steps and differ in what a wave keeps stationary:
//   beamform_ws_kernel    bf_mat fragments in registers, membrane fragments parked in LDS   (<= 16 channels, <= 512
//                         DoAs: the sweep; the fastest form, see its header for why; with y stored, the rows leave
//                         through an LDS block as whole-workgroup contiguous stores)
//   beamform_wsc_kernel   the same ownership for the complex Beamformer's planar source (<= 8 microphones, any DoA count)
//   beamform_gen_kernel   membrane fragments in registers, bf_mat streamed through LDS slabs  (everything else, <= 128 channels)
//
// Summation order (== oracle): LIF over past samples in chronological order (tau ascending),
// beamforming over channels ascending; both as fused multiply-add chains starting from +0.
#include "micloc_internal.h"

namespace micloc {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int BF_THREADS = BF_WAVES * 64;

typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

// Sum over the four 16-lane rows of a wave, result in every lane: gfx950's v_permlane16_swap / v_permlane32_swap
// exchange rows between two registers in the VALU (no LDS round trip as with ds_bpermute-based __shfl_xor).
__device__ __forceinline__ double row_sum4(double x)
{
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    uint2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    uint2_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);  // rows (0+1, 0+1, 2+3, 2+3)
    lo = __double2loint(s);
    hi = __double2hiint(s);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// The same sums for TWO values at once (the two 16-column tiles of a slab): rows 0 / 1 of the result hold the row sums of a / of b (rows 2 / 3
// repeat them).  The first exchange pairs a with b instead of a value with a copy of itself, so one add serves both: 2 + 1 + (2 + 2 + 1) = 8
// vector instructions for two sums instead of 2 x 10 -- and the additions are the ones of row_sum4 in the same pairing, (r0 + r1) + (r2 + r3):
// the same bits.
__device__ __forceinline__ double row_sum4_pair(double a, double b)
{
    uint2_t lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    uint2_t hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    // first operand: rows (a0, b0, a2, b2); second: rows (a1, b1, a3, b3)
    const double s = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);  // rows (a0+a1, b0+b1, a2+a3, b2+b3)
    const unsigned sl = __double2loint(s), sh = __double2hiint(s);
    lo = __builtin_amdgcn_permlane32_swap(sl, sl, false, false);
    hi = __builtin_amdgcn_permlane32_swap(sh, sh, false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// XCD-aware placement (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs (4 MB of L2
// each); consecutive chunks of a trial share the LIF halo rows, so an XCD is given a whole trial at a time and the halo
// becomes an L2 hit instead of a second HBM fetch.  Grid (nchunks, B), linear id L = chunk + nchunks * b -> XCD L % 8.
__device__ __forceinline__ void xcd_chunk_order(int &chunk, int &b)
{
    const int nchunks = gridDim.x, nb = gridDim.y;
    const int L = chunk + nchunks * b;
    const int full = (nb >> 3) << 3;  // trials in complete groups of 8
    if (L < full * nchunks) {
        const int j = L >> 3;
        const int bq = j / nchunks;
        chunk = j - bq * nchunks;
        b = 8 * bq + (L & 7);
    }
}

int beamform_nchunks(int T) { return (T + BF_CHUNK - 1) / BF_CHUNK; }

size_t beamform_partial_bytes(int B, int T, int Gp)
{
    // sized for 256-frame chunks, the finest chunking of any kernel here
    return ((size_t)B * ((T + 255) / 256) * Gp * sizeof(double) + 255) & ~(size_t)255;
}

// ---------------------------------------------------------------------------------------------------------------
// bf_mat-stationary form for up to 16 channels and power-only output (the sweep configuration, C = 14).
// Same arithmetic as beamform_kernel, different ownership: a wave keeps the bf_mat fragments of its own NGW DoA tiles
// in registers for the whole workgroup lifetime and walks over ALL 32 time tiles of the 512-frame chunk, whose membrane
// fragments the eight waves produced (4 tiles each) and parked in LDS in fragment order.  Per (time tile, DoA tile) the
// wave issues 4 MFMAs and 4 FMAs (running per-lane sum of squares); cross-lane reduction and the store happen once per
// workgroup instead of once per DoA tile, there is no cross-wave reduction at all (a DoA column belongs to one wave),
// and bf_mat never goes through LDS.  VALU/LDS instructions per 16 MFMAs: ~25 instead of ~50.
// ---------------------------------------------------------------------------------------------------------------

// KV > 0 (14 channels -- the 7-microphone array; also 6 and 10): the last k-step would multiply KV real channels and
// 4 - KV rows of zero padding.  A 16x16x4 fp64 MFMA holds the SIMD for 64 cycles and fp64 VALU work is not hidden behind
// it (tools/mfma_valu_overlap.hip), so the last KV channels are cheaper as 4 KV plain FMAs on the accumulators (17.6
// cycles each set of 4): 3 MFMAs + 8 FMAs = 227 instead of 256 cycles per (time tile, DoA tile) for C = 14.  The order of
// the sum is unchanged -- channels 0..11 inside the MFMAs, then 12, 13 -- so y and the power stay bit-identical.
// KM = k-steps on the matrix cores, KV = channels 4 KM .. 4 KM + KV - 1 on the vector ALU (fewer than 13 channels: only the
// k-steps that hold channels at all).
template <int NG, int TILES, int KM, int KV, bool LEAN = false>
__device__ __forceinline__ void ws_stage2_kv(const double *Vl, const double *__restrict__ Wp, int Gp, int wv, int l, int ntile,
                                             double *__restrict__ pout, int j0 = 0)
{
    if constexpr (NG > 0) {
        const int lc = l & 15;
        const int q = l >> 4;
        constexpr int KVD = KV > 0 ? KV : 1;
        double Wf[NG][KM], Wv[NG][KVD];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const double *wp = Wp + 16 * (wv + BF_WAVES * (j0 + j)) + lc;
#pragma unroll
            for (int k = 0; k < KM; ++k) Wf[j][k] = wp[(size_t)(4 * k + q) * Gp];
#pragma unroll
            for (int i = 0; i < KV; ++i) Wv[j][i] = wp[(size_t)(4 * KM + i) * Gp];
        }
        double sq[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) sq[j] = 0.0;
        auto ldv = [&](int tile, double (&V)[KM]) {
            const double *p = Vl + (size_t)(tile < TILES ? tile : TILES - 1) * 256 + l;
#pragma unroll
            for (int k = 0; k < KM; ++k) V[k] = p[64 * k];
        };
        // membrane values of channels 4 KM + i at this lane's accumulator rows t = q + 4 r: fragment KM keeps channel
        // 4 KM + i at lane index 16 i + t
        auto ldvv = [&](int tile, double (&Vv)[KVD][4]) {
            const double *p = Vl + (size_t)tile * 256 + 64 * KM + q;
#pragma unroll
            for (int i = 0; i < KV; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Vv[i][r] = p[16 * i + 4 * r];
        };
        auto tile_step = [&](int tile, const double (&V)[KM]) {
            double Vv[KVD][4];
            if (!LEAN) ldvv(tile, Vv);
            double4_t acc[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < KM; ++k)
#pragma unroll
                for (int j = 0; j < NG; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[k], Wf[j][k], acc[j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KV; ++i) {
                if (LEAN) {  // one channel's four rows at a time: 8 registers instead of 8 KV
                    const double *p = Vl + (size_t)tile * 256 + 64 * KM + q + 16 * i;
#pragma unroll
                    for (int r = 0; r < 4; ++r) Vv[0][r] = p[4 * r];
                }
#pragma unroll
                for (int j = 0; j < NG; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[j][r] = __builtin_fma(Vv[LEAN ? 0 : i][r], Wv[j][i], acc[j][r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NG; ++j) sq[j] = __builtin_fma(acc[j][r], acc[j][r], sq[j]);
        };
        if (LEAN) {
            // register-lean form (three workgroups per CU): fragments fetched per tile, the other waves of the SIMD cover the
            // LDS latency
            for (int t = 0; t < ntile; ++t) {
                double V[KM];
                ldv(t, V);
                tile_step(t, V);
            }
        } else {
            double VA[KM], VB[KM];
            if (ntile > 0) ldv(0, VA);
            int t = 0;
            for (; t + 1 < ntile; t += 2) {
                ldv(t + 1, VB);
                tile_step(t, VA);
                ldv(t + 2, VA);
                tile_step(t + 1, VB);
            }
            if (t < ntile) tile_step(t, VA);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const double s = row_sum4(sq[j]);
            if (l < 16) pout[16 * (wv + BF_WAVES * (j0 + j)) + l] = s;
        }
    }
}

template <int NG, int TILES, bool PING>
__device__ __forceinline__ void ws_stage2(const double *Vl, const double *__restrict__ Wp, int Gp, int wv, int l,
                                          int ntile, double *__restrict__ pout)
{
    if constexpr (NG > 0) {
        const int lc = l & 15;
        const int q = l >> 4;
        // bf_mat fragments of this wave's DoA tiles (gt = wv, wv + 8, ...): straight from L2, once
        double Wf[NG][4];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const double *wp = Wp + 16 * (wv + BF_WAVES * j) + lc;
#pragma unroll
            for (int k = 0; k < 4; ++k) Wf[j][k] = wp[(size_t)(4 * k + q) * Gp];
        }
        double sq[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) sq[j] = 0.0;

        auto ldv = [&](int tile, double (&V)[4]) {
            const double *p = Vl + (size_t)(tile < TILES ? tile : TILES - 1) * 256 + l;
#pragma unroll
            for (int k = 0; k < 4; ++k) V[k] = p[64 * k];
        };
        auto mm = [&](const double (&V)[4], double4_t (&acc)[NG]) {
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < NG; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[k], Wf[j][k], acc[j], 0, 0, 0);
        };
        auto sqr = [&](const double4_t (&acc)[NG]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < NG; ++j) sq[j] = __builtin_fma(acc[j][r], acc[j][r], sq[j]);
        };

        if (!PING) {
            // register-lean form (three workgroups per CU): one accumulator set; the squares wait for their own MFMAs
            // and the other waves of the SIMD fill the gap
            double VA[4], VB[4];
            double4_t acc[NG];
            if (ntile > 0) ldv(0, VA);
            int t = 0;
            for (; t + 1 < ntile; t += 2) {
                ldv(t + 1, VB);
                mm(VA, acc);
                sqr(acc);
                ldv(t + 2, VA);
                mm(VB, acc);
                sqr(acc);
            }
            if (t < ntile) {
                mm(VA, acc);
                sqr(acc);
            }
        } else if (ntile > 0) {
            double VA[4], VB[4];
            double4_t accA[NG], accB[NG];
            ldv(0, VA);
            ldv(1, VB);
            mm(VA, accA);
            int t = 0;
            // Each half of the loop body = 4 NG MFMAs of one time tile + the 4 NG squares of the previous one + the
            // fragment loads of the next.  Left alone the compiler folds both accumulator sets into one and runs
            // MFMAs -> s_nop -> squares back to back; the group barriers pin "all MFMAs, then all squares of the
            // previous tile" (VALU work is never hidden behind MFMAs on this SIMD, but batching it pays the
            // MFMA<->VALU switch once per tile), the full barriers keep the two halves apart.
            auto pin = [&]() {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // DS read (next tile's fragments)
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NG - 1, 0);  // the remaining MFMAs, back to back
                __builtin_amdgcn_sched_group_barrier(0x002, 4 * NG, 0);  // then the squares of the previous tile
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            for (; t + 2 < ntile; t += 2) {
                mm(VB, accB);
                ldv(t + 2, VA);
                sqr(accA);
                pin();
                mm(VA, accA);
                ldv(t + 3, VB);
                sqr(accB);
                pin();
            }
            if (t + 1 < ntile) {
                mm(VB, accB);
                sqr(accA);
                pin();
                sqr(accB);
            } else {
                sqr(accA);
            }
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const double s = row_sum4(sq[j]);
            if (l < 16) pout[16 * (wv + BF_WAVES * j) + l] = s;
        }
    }
}

// The same ownership with the T x G product stored (API parity with apply_to_signal): HBM-write bound, and what decides
// the rate is the shape of the stores (tools/store_bw.hip: accumulator tiles stored as they are -- 4 rows x 128 B per
// instruction -- reach 3.7 TB/s for 64-byte aligned rows and 2.5 TB/s for G = 449; a workgroup writing a contiguous
// block front to back reaches 5.6 TB/s whatever the row length).  The 8 waves walk the time tiles together, so after each
// tile the workgroup holds 16 complete rows of y = one contiguous block of 16 G doubles: it is assembled in LDS in
// memory order and copied out by all 512 threads.  A wave without a DoA tile in its last slot (gt >= GT) multiplies
// against a clamped fragment and stores nothing, so every wave runs the same code and reaches the same barriers.
template <int NG, int TILES>
__device__ __forceinline__ void ws_stage2_y(const double *Vl, const double *__restrict__ Wp, int GT, int G, int wv, int l,
                                            int tid, int ntile, int nrows, double *stg, double *__restrict__ yb,
                                            double *__restrict__ pout)
{
    const int Gp = 16 * GT;
    const int lc = l & 15;
    const int q = l >> 4;
    double Wf[NG][4];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int gt = wv + BF_WAVES * j;
        const double *wp = Wp + 16 * (gt < GT ? gt : GT - 1) + lc;
#pragma unroll
        for (int k = 0; k < 4; ++k) Wf[j][k] = wp[(size_t)(4 * k + q) * Gp];
    }
    double sq[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) sq[j] = 0.0;
    for (int t = 0; t < ntile; ++t) {
        double V[4];
        const double *p = Vl + (size_t)t * 256 + l;
#pragma unroll
        for (int k = 0; k < 4; ++k) V[k] = p[64 * k];
        double4_t acc[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[k], Wf[j][k], acc[j], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NG; ++j) sq[j] = __builtin_fma(acc[j][r], acc[j][r], sq[j]);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int col = 16 * (wv + BF_WAVES * j) + lc;
            if (col < G) {
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[(q + 4 * r) * G + col] = acc[j][r];
            }
        }
        __syncthreads();
        const int rows = nrows - 16 * t < 16 ? nrows - 16 * t : 16;
        const int n = rows * G;
        double *yo = yb + (size_t)16 * t * G;
#pragma unroll 4
        for (int e = tid; e < n; e += BF_THREADS) yo[e] = stg[e];
        __syncthreads();  // the block is on its way: the next tile may overwrite it
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const double s = row_sum4(sq[j]);
        const int gt = wv + BF_WAVES * j;
        if (pout && l < 16 && gt < GT) pout[16 * gt + l] = s;
    }
}

// Waves per SIMD the 6 / 10 / 14-channel instantiations are compiled for.  6 = three workgroups per CU, which needs the
// register-lean stage 2 (fragments fetched per tile, one tail channel's rows at a time: 70 VGPRs for three DoA tiles per
// wave; the double-buffered form takes 100 and runs two per CU: 1.13 against 1.07 ms per launch, step 1.63 against 1.59 ms).
// Four DoA tiles per wave (G > 384) do not fit 80 registers and run as two passes of two.
constexpr int WS_KV_WAVES = 6;
template <int NGW, int NT, bool WANT_Y, int KM, int KV>
__global__ __launch_bounds__(BF_THREADS, WANT_Y ? 2 : (KV ? WS_KV_WAVES : (NT == 2 ? 6 : 4))) void beamform_ws_kernel(
    const int8_t *__restrict__ spikes, const double *__restrict__ ntab_g, int NK, const double *__restrict__ Wp, int GT, int C,
    int T, double *__restrict__ partial, int G, double *__restrict__ y, const int *__restrict__ chunk_range)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Gp = 16 * GT;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, in an SGPR
    const int l = tid & 63;
    const int lc = l & 15;
    const int q = l >> 4;
    int chunk = blockIdx.x, b = blockIdx.y;
    xcd_chunk_order(chunk, b);
    if (chunk_range && (chunk < chunk_range[0] || chunk >= chunk_range[1])) return;  // (workgroup-uniform)
    if (chunk_range) T = chunk_range[6];  // streaming: the frames of the window that exist so far (decided on the device)
    const int nchunks = gridDim.x;
    constexpr int CH = BF_WAVES * NT * 16;  // frames per workgroup
    constexpr int TILES = CH / 16;
    const int cs = chunk * CH;

    // [ union{ spike tile as fp64 [R][16] , V fragments [32 tiles][4 k-steps][64 lanes] } ][ nir table ]
    // The spikes are converted to fp64 once, while they are staged (each row feeds ~3 time tiles of the Toeplitz
    // product): the LIF loop is then LDS reads + MFMAs only.  VALU instructions do not overlap MFMAs on a gfx950 SIMD
    // (tools/mfma_valu_overlap.hip), so every conversion saved is matrix-pipe time won.
    const int R = CH + 4 * NK - 16;
    double *S = reinterpret_cast<double *>(smem);
    double *Vl = S;
    double *ntab = S + (R * 16 > TILES * 256 ? R * 16 : TILES * 256);
    const int ntab_len = 4 * NK + 16;

    for (int e = tid; e < ntab_len; e += BF_THREADS) ntab[e] = ntab_g[e];
    {
        const int8_t *sb = spikes + (size_t)b * (chunk_range ? chunk_range[3] : T) * C;  // ([3]: frames per trial of a raster window)
        const int tau0 = cs + 16 - 4 * NK;
        const int c = tid & 15;
        constexpr int RP = BF_THREADS / 16;  // rows per pass
        const int rr = tid >> 4;
        if (c >= C) {
            for (int rho = rr; rho < R; rho += RP) S[rho * 16 + c] = 0.0;  // channel padding
        } else if (tau0 >= 0 && tau0 + R <= T) {
            // interior chunk (workgroup-uniform): no clamping, constant strides
            const int8_t *p = sb + (size_t)(tau0 + rr) * C + c;
            double *d = S + rr * 16 + c;
            int rho = rr;
            for (; rho + 5 * RP < R; rho += 6 * RP) {
                int8_t v[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) v[i] = p[(size_t)i * RP * C];
#pragma unroll
                for (int i = 0; i < 6; ++i) d[i * RP * 16] = (double)v[i];
                p += (size_t)6 * RP * C;
                d += 6 * RP * 16;
            }
            for (; rho < R; rho += RP) {
                *d = (double)*p;
                p += (size_t)RP * C;
                d += RP * 16;
            }
        } else {
            // first / last chunk of a trial: rows outside [0, T) are zero
            for (int r0 = rr; r0 < R; r0 += RP * 6) {
                int8_t v[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    int tau = tau0 + r0 + RP * i;
                    tau = tau < 0 ? 0 : (tau >= T ? T - 1 : tau);
                    v[i] = sb[(size_t)tau * C + c];
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int rho = r0 + RP * i;
                    const int tau = tau0 + rho;
                    if (rho < R) S[rho * 16 + c] = (tau >= 0 && tau < T) ? (double)v[i] : 0.0;
                }
            }
        }
    }
    __syncthreads();

    // ---- stage 1: membrane fragments of this wave's 4 time tiles ---------------------------------------------
    const int tb0 = cs + wv * NT * 16;
    const bool active = tb0 < T;  // wave-uniform
    double4_t vacc[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) vacc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
    if (active) {
        const double *sp = S + (size_t)(tb0 - cs + q) * 16 + lc;
        const double *np_ = ntab + (lc - q + 4 * NK - 16 + 15 - 12);  // 12 below the tap row of k-step 0
        // no register double-buffering here: it costs v_mov's (VALU time = MFMA time lost); the other waves of the
        // SIMD cover the LDS latency
        auto kstep = [&](const double *spk_, const double *nir_, int u) {
            const double bn = nir_[12 - 4 * u];  // non-negative immediate offsets only
            double a[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) a[tt] = spk_[(16 * tt + 4 * u) * 16];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                vacc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tt], bn, vacc[tt], 0, 0, 0);
        };
        int ks = 0;
        for (; ks + 4 <= NK; ks += 4) {  // four k-steps per pointer update (immediate LDS offsets)
#pragma unroll
            for (int u = 0; u < 4; ++u) kstep(sp, np_, u);
            sp += 4 * 4 * 16;
            np_ -= 4 * 4;
        }
        for (; ks < NK; ++ks) {
            kstep(sp, np_, 0);
            sp += 4 * 16;
            np_ -= 4;
        }
    }
    __syncthreads();  // every wave is done with the spike tile: the V fragments may overwrite it
    if (active) {
        if (tb0 + NT * 16 > T) {  // only the wave that straddles the end of the trial masks
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const bool tvalid = (tb0 + 16 * tt + lc) < T;
#pragma unroll
                for (int r = 0; r < 4; ++r) vacc[tt][r] = tvalid ? vacc[tt][r] : 0.0;
            }
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            double *vp = Vl + (size_t)(wv * NT + tt) * 256 + l;
#pragma unroll
            for (int r = 0; r < 4; ++r) vp[64 * r] = vacc[tt][r];
        }
    }
    __syncthreads();

    // ---- stage 2: this wave's DoA tiles against every time tile of the chunk --------------------------------
    int ntile = (T - cs + 15) >> 4;
    ntile = ntile > TILES ? TILES : ntile;
    double *pout = partial + ((size_t)b * nchunks + chunk) * Gp;
    if constexpr (WANT_Y) {
        double *stg = ntab + ((ntab_len + 1) & ~1);  // 16 rows x G, exactly as they lie in memory
        ws_stage2_y<NGW, TILES>(Vl, Wp, GT, G, wv, l, tid, ntile, T - cs, stg, y + ((size_t)b * T + cs) * G,
                                partial ? pout : nullptr);
        return;
    }
    // NGW = ceil(GT / 8): every wave owns NGW or NGW - 1 DoA tiles (wave-uniform choice of the instantiation).  (G = 360: 23 tiles,
    // seven waves own three and one owns two.  Rotating the lighter share over the waves with the workgroup index -- in case the same
    // SIMD of a compute unit were the light one in all of its workgroups -- was measured: 1007 against 999 us, G = 449: 1227 against
    // 1212; profiles/r5/experiments/ws_owner_rotation.txt.  Not that.)
    const int wo = wv;
    if constexpr (KM < 4 || NGW == 4) {  // (16 channels with four DoA tiles per wave: the two-pass form as well -- 8 B of scratch otherwise)
        constexpr bool LEAN = (KV > 0 || NGW == 4) && WS_KV_WAVES >= 6;
        if constexpr (LEAN && NGW == 4) {
            // four DoA tiles per wave do not fit 80 registers: two passes of two over the parked fragments (the LDS reads
            // double, the MFMAs do not) keep three workgroups per CU
            ws_stage2_kv<2, TILES, KM, KV, true>(Vl, Wp, Gp, wo, l, ntile, pout, 0);
            if (wo + BF_WAVES * 3 < GT)
                ws_stage2_kv<2, TILES, KM, KV, true>(Vl, Wp, Gp, wo, l, ntile, pout, 2);
            else
                ws_stage2_kv<1, TILES, KM, KV, true>(Vl, Wp, Gp, wo, l, ntile, pout, 2);
            return;
        }
        if (wo + BF_WAVES * (NGW - 1) < GT)
            ws_stage2_kv<NGW, TILES, KM, KV, LEAN>(Vl, Wp, Gp, wo, l, ntile, pout);
        else
            ws_stage2_kv<NGW - 1, TILES, KM, KV, LEAN>(Vl, Wp, Gp, wo, l, ntile, pout);
        return;
    }
    if (wo + BF_WAVES * (NGW - 1) < GT)
        ws_stage2<NGW, TILES, NT != 2>(Vl, Wp, Gp, wo, l, ntile, pout);
    else
        ws_stage2<NGW - 1, TILES, NT != 2>(Vl, Wp, Gp, wo, l, ntile, pout);
}

static size_t ws_lds_bytes(const NeuronTab &nt, int NT, int Gy = 0)
{
    const size_t tile = (size_t)(BF_WAVES * NT * 16 + 4 * nt.NK - 16) * 16, vfrag = (size_t)BF_WAVES * NT * 256;
    const size_t tab = (size_t)(4 * nt.NK + 16);
    return ((tile > vfrag ? tile : vfrag) + (Gy ? ((tab + 1) & ~(size_t)1) + (size_t)16 * Gy : tab)) * sizeof(double);
}

template <int NGW, int NT, bool WANT_Y, int KM = 4, int KV = 0>
static hipError_t launch_ws_n(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T,
                              double *partial, double *y, hipStream_t stream)
{
    if constexpr (!WANT_Y && KM == 4) {
        // only the k-steps that hold channels; 6 / 10 / 14 channels: the last two on the vector ALU instead of a half-empty
        // k-step (the ws_k4 variant build keeps four k-steps: ablation only, same results)
        if (!VARIANT_WS_FOUR_KSTEPS) {
#define WS_K(KM_, KV_) return launch_ws_n<NGW, NT, false, KM_, KV_>(W, nt, spikes, B, T, partial, y, stream)
            switch (W.C) {  // (the plan's channel count is 2 x microphones: always even)
                case 1: case 2: case 3: case 4: WS_K(1, 0);
                case 6: WS_K(1, 2);
                case 5: case 7: case 8: WS_K(2, 0);
                case 10: WS_K(2, 2);
                case 9: case 11: case 12: WS_K(3, 0);
                case 14: WS_K(3, 2);
                default: break;
            }
#undef WS_K
        }
    }
    dim3 grid((T + BF_WAVES * NT * 16 - 1) / (BF_WAVES * NT * 16), B), block(BF_THREADS);
    const size_t lds = ws_lds_bytes(nt, NT, WANT_Y ? W.G : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto k = &beamform_ws_kernel<NGW, NT, WANT_Y, KM, KV>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, block, lds, stream, spikes, nt.tab, nt.NK, W.Wp, W.GT, W.C, T, partial, W.G, y, W.chunk_range);
    return hipGetLastError();
}

// 16-frame tiles per wave.  2 = 256-frame workgroups, 38 KB of LDS and (with the single-accumulator stage 2) 66 VGPRs:
// three workgroups per CU, whose staging / LIF / beamforming phases overlap better than those of two 512-frame ones
// (1.15 ms against 1.21 ms on the sweep shape; the multi-stream sweep, where other kernels fill those gaps anyway,
// is unchanged).
constexpr int WS_NT = 2;

static bool ws_eligible(const BeamformW &W, const NeuronTab &nt, bool want_y)
{
    return W.CT == 1 && W.GT <= 4 * BF_WAVES && ws_lds_bytes(nt, WS_NT, want_y ? W.G : 0) <= 160 * 1024;
}

template <int NT, bool WANT_Y>
static hipError_t launch_ws_nt(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T,
                               double *partial, double *y, hipStream_t stream)
{
    switch ((W.GT + BF_WAVES - 1) / BF_WAVES) {
        case 1: return launch_ws_n<1, NT, WANT_Y>(W, nt, spikes, B, T, partial, y, stream);
        case 2: return launch_ws_n<2, NT, WANT_Y>(W, nt, spikes, B, T, partial, y, stream);
        case 3: return launch_ws_n<3, NT, WANT_Y>(W, nt, spikes, B, T, partial, y, stream);
        case 4: return launch_ws_n<4, NT, WANT_Y>(W, nt, spikes, B, T, partial, y, stream);
        default: return hipErrorInvalidValue;
    }
}

static hipError_t launch_ws(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T,
                            double *partial, double *y, hipStream_t stream, int *nchunks)
{
    *nchunks = (T + BF_WAVES * WS_NT * 16 - 1) / (BF_WAVES * WS_NT * 16);
    if (y) return launch_ws_nt<WS_NT, true>(W, nt, spikes, B, T, partial, y, stream);
    return launch_ws_nt<WS_NT, false>(W, nt, spikes, B, T, partial, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// The complex Beamformer's contraction  y = h @ conj(bf_mat)  (micloc/beamformer.py:290) in the bf_mat-stationary form.
// Real stacked form: [h_re | h_im] (T x 2M) @ [Wre; Wim | -Wim; Wre] (2M x 2 Ghp): columns [0, Ghp) are Re y, columns
// [Ghp, 2 Ghp) Im y -- 8 M G flop per frame, exactly the complex product's count.  No LIF: the planar band-passed rows
// [B][2M][Ts] are read as MFMA A-fragments straight from HBM (lane l of k-step k: channel 4k + (l>>4), frame l&15 -- four
// 128-byte segments per load) and parked in LDS in fragment order; from there on it is beamform_ws_kernel's stage 2: a
// wave keeps the fragments of up to three DoA tiles in registers and walks the 16 time tiles of the chunk, then takes
// its next three tiles (G = 360 complex: 46 tiles, two passes) -- 32 KB of LDS, <= 80 VGPRs, three workgroups per CU.
// Same order of the channel sum as beamform_kernel (k-steps ascending, the last KV channels as plain FMAs behind them).
//
// With y stored (apply_to_signal's T x G complex128 array, 16 G bytes per frame: HBM-write bound) a wave owns PAIRS of
// tiles (Re and Im of the same 16 DoAs), two pairs per pass = 256 complex columns per workgroup and pass; after every
// time tile the workgroup assembles 8 rows x 256 interleaved (re, im) values in LDS and every wave copies one row segment
// (4 KB contiguous) out -- whole-cache-line stores whatever G is (tools/store_bw.hip).
// ---------------------------------------------------------------------------------------------------------------
template <int TILES, int KM, int KV, bool FLAT>
__device__ __forceinline__ void wsc_stage2_y(const double *Vl, const double *__restrict__ Wp, int GT, int Gc, int wv, int l, int ntile,
                                             int nrows, double2 *stg, double2 *__restrict__ yb, double *__restrict__ pout)
{
    const int Gp = 16 * GT, GTc = GT >> 1;
    const int lc = l & 15, q = l >> 4;
    constexpr int KVD = KV > 0 ? KV : 1;
    for (int p0 = 0; 16 * p0 < GTc; ++p0) {  // pass: complex DoA tiles 16 p0 .. 16 p0 + 15 (the same trip count in every wave)
        const int g0 = 256 * p0;
        const int ncol = Gc - g0 < 256 ? Gc - g0 : 256;
        double Wf[2][2][KM], Wv[2][2][KVD];
        bool own[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ct = 16 * p0 + wv + BF_WAVES * j;
            own[j] = ct < GTc;  // (wave-uniform; a wave without a pair multiplies a clamped one and stores nothing)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const double *wp = Wp + 16 * ((own[j] ? ct : GTc - 1) + part * GTc) + lc;
#pragma unroll
                for (int k = 0; k < KM; ++k) Wf[j][part][k] = wp[(size_t)(4 * k + q) * Gp];
#pragma unroll
                for (int i = 0; i < KV; ++i) Wv[j][part][i] = wp[(size_t)(4 * KM + i) * Gp];
            }
        }
        double sq[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        for (int t = 0; t < ntile; ++t) {
            double V[KM];
            const double *p = Vl + (size_t)t * 256 + l;
#pragma unroll
            for (int k = 0; k < KM; ++k) V[k] = p[64 * k];
            double4_t acc[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int part = 0; part < 2; ++part) acc[j][part] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < KM; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int part = 0; part < 2; ++part)
                        acc[j][part] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[k], Wf[j][part][k], acc[j][part], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < KV; ++i) {
                double Vv[4];
                const double *pv = Vl + (size_t)t * 256 + 64 * KM + q + 16 * i;
#pragma unroll
                for (int r = 0; r < 4; ++r) Vv[r] = pv[4 * r];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int part = 0; part < 2; ++part)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[j][part][r] = __builtin_fma(Vv[r], Wv[j][part][i], acc[j][part][r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int part = 0; part < 2; ++part) sq[j][part] = __builtin_fma(acc[j][part][r], acc[j][part][r], sq[j][part]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // rows 8h .. 8h + 7 of the time tile (accumulator rows q + 4r, r = 2h, 2h + 1)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = 16 * (wv + BF_WAVES * j) + lc;
                    if (own[j] && col < ncol) {
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            double2 v;
                            v.x = acc[j][0][2 * h + rr];
                            v.y = acc[j][1][2 * h + rr];
                            stg[(q + 4 * rr) * ncol + col] = v;
                        }
                    }
                }
                __syncthreads();
                if constexpr (FLAT) {
                    // (G <= 256) one pass covers every column: the 8 rows are ONE contiguous block of y, copied front to back
                    const int r0 = 16 * t + 8 * h;
                    const int n = (nrows - r0 < 8 ? (nrows - r0 > 0 ? nrows - r0 : 0) : 8) * Gc;
                    double2 *yo = yb + (size_t)r0 * Gc;
                    for (int e = wv * 64 + l; e < n; e += BF_THREADS) yo[e] = stg[e];
                } else {
                    const int trow = 16 * t + 8 * h + wv;  // wave wv copies row wv of the block: one contiguous segment
                    if (trow < nrows) {
                        double2 *yo = yb + (size_t)trow * Gc + g0;
                        const double2 *so = stg + wv * ncol;
                        for (int c = l; c < ncol; c += 64) yo[c] = so[c];
                    }
                }
                __syncthreads();  // the block is on its way: the next half tile may overwrite it
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const double s = row_sum4(sq[j][part]);
                const int ct = 16 * p0 + wv + BF_WAVES * j;
                if (pout && l < 16 && own[j]) pout[16 * (ct + part * GTc) + l] = s;
            }
    }
}

template <int KM, int KV, bool WANT_Y, bool FLAT = false>
__global__ __launch_bounds__(BF_THREADS, WANT_Y ? 4 : 6) void beamform_wsc_kernel(const double *__restrict__ pre,
                                                                                   const double *__restrict__ Wp, int GT, int C, int T,
                                                                                   int Ts, double *__restrict__ partial, int Gc,
                                                                                   double *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 2;                   // 16-frame tiles per wave
    constexpr int CH = BF_WAVES * NT * 16;  // frames per workgroup (256)
    constexpr int TILES = CH / 16;
    constexpr int KF = KM + (KV > 0 ? 1 : 0);  // fragments per time tile (the last one holds the KV tail channels)
    const int Gp = 16 * GT;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int lc = l & 15;
    const int q = l >> 4;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int nchunks = gridDim.x;
    const int cs = chunk * CH;
    double *Vl = reinterpret_cast<double *>(smem);  // [TILES][4][64]: fragment order

    // ---- stage 1: this wave's two time tiles as A-fragments, HBM -> registers -> LDS ------------------------------
    {
        const double *pb = pre + (size_t)b * C * Ts;
        double v[NT][KF];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int t = cs + (wv * NT + tt) * 16 + lc;
#pragma unroll
            for (int k = 0; k < KF; ++k) {
                const int c = 4 * k + q;
                v[tt][k] = pb[(size_t)(c < C ? c : C - 1) * Ts + (t < T ? t : T - 1)];  // (clamped: unconditional loads)
            }
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int t = cs + (wv * NT + tt) * 16 + lc;
#pragma unroll
            for (int k = 0; k < KF; ++k) {
                const int c = 4 * k + q;
                Vl[(size_t)(wv * NT + tt) * 256 + 64 * k + l] = (t < T && c < C) ? v[tt][k] : 0.0;
            }
        }
    }
    __syncthreads();

    // ---- stage 2 ---------------------------------------------------------------------------------------------------
    int ntile = (T - cs + 15) >> 4;
    ntile = ntile > TILES ? TILES : ntile;
    double *pout = partial ? partial + ((size_t)b * nchunks + chunk) * Gp : nullptr;
    if constexpr (WANT_Y) {
        double2 *stg = reinterpret_cast<double2 *>(Vl + TILES * 256);  // 8 rows x <= 256 complex columns
        wsc_stage2_y<TILES, KM, KV, FLAT>(Vl, Wp, GT, Gc, wv, l, ntile, T - cs, stg, reinterpret_cast<double2 *>(y) + ((size_t)b * T + cs) * Gc,
                                    pout);
    } else {
        const int ntl = (GT - wv + BF_WAVES - 1) / BF_WAVES;  // DoA tiles of this wave: wv, wv + 8, ... (wave-uniform)
        for (int j0 = 0; j0 < ntl; j0 += 3) {
            const int n = ntl - j0;
            if (n >= 3)
                ws_stage2_kv<3, TILES, KM, KV, true>(Vl, Wp, Gp, wv, l, ntile, pout, j0);
            else if (n == 2)
                ws_stage2_kv<2, TILES, KM, KV, true>(Vl, Wp, Gp, wv, l, ntile, pout, j0);
            else
                ws_stage2_kv<1, TILES, KM, KV, true>(Vl, Wp, Gp, wv, l, ntile, pout, j0);
        }
    }
}

constexpr int WSC_CHUNK = BF_WAVES * 2 * 16;

static bool wsc_eligible(const BeamformW &W) { return W.CT == 1 && W.complex_pairs && (W.GT & 1) == 0; }

template <int KM, int KV>
static hipError_t launch_wsc_k(const BeamformW &W, const double *pre, int B, int T, int Ts, double *y, double *partial,
                               hipStream_t stream)
{
    const size_t lds = (size_t)16 * 256 * sizeof(double) + (y ? (size_t)8 * 256 * sizeof(double2) : 0);
    dim3 grid((T + WSC_CHUNK - 1) / WSC_CHUNK, B), block(BF_THREADS);
    if (y && W.G / 2 <= 256)
        hipLaunchKernelGGL((beamform_wsc_kernel<KM, KV, true, true>), grid, block, lds, stream, pre, W.Wp, W.GT, W.C, T, Ts, partial, W.G / 2, y);
    else if (y)
        hipLaunchKernelGGL((beamform_wsc_kernel<KM, KV, true>), grid, block, lds, stream, pre, W.Wp, W.GT, W.C, T, Ts, partial, W.G / 2, y);
    else
        hipLaunchKernelGGL((beamform_wsc_kernel<KM, KV, false>), grid, block, lds, stream, pre, W.Wp, W.GT, W.C, T, Ts, partial, W.G / 2, y);
    return hipGetLastError();
}

static hipError_t launch_wsc(const BeamformW &W, const double *pre, int B, int T, int Ts, double *y, double *partial, hipStream_t stream)
{
#define WSC_K(KM_, KV_) return launch_wsc_k<KM_, KV_>(W, pre, B, T, Ts, y, partial, stream)
    switch (W.C) {  // (2 x microphones: even)
        case 2: case 4: WSC_K(1, 0);
        case 6: WSC_K(1, 2);
        case 8: WSC_K(2, 0);
        case 10: WSC_K(2, 2);
        case 12: WSC_K(3, 0);
        case 14: WSC_K(3, 2);
        case 16: WSC_K(4, 0);
        default: return hipErrorInvalidValue;
    }
#undef WSC_K
}

// ---------------------------------------------------------------------------------------------------------------
// General form (everything the bf_mat-stationary kernels do not cover: more than 16 channels -- up to the 128 of the
// 64-microphone stress configuration --, more than 512 DoAs, y stored for those shapes, the planar source of the complex
// Beamformer with more than 8 microphones).  Membrane-stationary: a wave keeps the fragments of its NT time tiles in
// registers (NT x CT x 8 VGPRs; NT = 1 for 128 channels) and bf_mat is streamed through LDS in double-buffered slabs of
// 32 DoA columns shared by the 8 waves (every byte leaves L2 once per workgroup and 128-frame sub-chunk), the per-slab
// column sums are combined across waves right away.  Sized for TWO workgroups per CU (<= 128 VGPRs, <= 80 KB of LDS): one
// workgroup's staging / LIF / barrier bubbles are the other's MFMA time -- the one-workgroup form of rounds 1-3 (two time
// tiles per wave, 170 VGPRs + 80 B of scratch, padded 98 KB slabs) sat at 0.72 of the matrix peak with the pipe idle a
// quarter of the time.  What makes it fit:
//   * slab rows are 32 doubles (256 B) with the two 16-column halves swapped in odd rows: the B-fragment read of a k-step
//     (rows 4k + q, 16 columns) touches disjoint banks for q and q + 1 without the 50 % row padding;
//   * the int8 spike tile of the sub-chunk and the nir table alias slab buffer 1, which is first written after the LIF
//     phase (one extra barrier);
//   * a workgroup walks NSUB sub-chunks and keeps the column sums of its chunk in LDS: one row of partial sums per
//     NSUB x 128 frames instead of one per sub-chunk.
// ---------------------------------------------------------------------------------------------------------------
constexpr int GEN_COLS = 32;  // DoA columns per slab (two 16-column tiles)
constexpr int gen_nt(int CT) { return CT > 4 ? 1 : 2; }
constexpr int gen_nsub(int CT) { return CT > 4 ? 4 : 1; }
constexpr int gen_chunk(int CT) { return BF_WAVES * gen_nt(CT) * 16 * gen_nsub(CT); }

int beamform_nchunks_ct(int T, int CT)
{
    const int ch = CT > 4 ? gen_chunk(8) : gen_chunk(1);
    return (T + ch - 1) / ch;
}

struct GenLds {
    size_t u_doubles, total_bytes;
};
static GenLds gen_lds(int CT, int NK, bool src_spikes, int Gp)
{
    const int Kp = 16 * CT, NT = gen_nt(CT), NSUB = gen_nsub(CT);
    size_t u = (size_t)Kp * GEN_COLS;  // slab buffer 1
    if (src_spikes) {
        const size_t tile = (size_t)(4 * NK + 16) + ((size_t)(BF_WAVES * NT * 16 + 4 * NK - 16) * Kp + 7) / 8;
        u = tile > u ? tile : u;
    }
    u = (u + 1) & ~(size_t)1;
    size_t d = (size_t)Kp * GEN_COLS + u + (size_t)2 * BF_WAVES * GEN_COLS + (NSUB > 1 ? (size_t)Gp + GEN_COLS : 0);
    return {u, d * sizeof(double)};
}

template <int CT, bool SRC_SPIKES, bool WANT_Y>
__global__ __launch_bounds__(BF_THREADS, 4) void beamform_gen_kernel(const int8_t *__restrict__ spikes, const double *__restrict__ pre,
                                                                      const double *__restrict__ ntab_g, int NK,
                                                                      const double *__restrict__ Wp, int GT, int C, int G, int T, int Ts,
                                                                      double *__restrict__ y, int y_complex, double *__restrict__ partial,
                                                                      const int *__restrict__ chunk_range, int u_doubles, int vec_stage)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = gen_nt(CT), NSUB = gen_nsub(CT);
    constexpr int Kp = 16 * CT;
    constexpr int Cs = 16 * CT;
    constexpr int KS = 4 * CT;
    constexpr int SUB = BF_WAVES * NT * 16;  // frames per sub-chunk
    constexpr int CH = SUB * NSUB;           // frames per workgroup = per row of partial sums
    const int Gp = 16 * GT;
    const int NSL = (GT + 1) >> 1;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int lc = l & 15;
    const int q = l >> 4;
    int chunk = blockIdx.x, b = blockIdx.y;
    xcd_chunk_order(chunk, b);
    if (chunk_range && (chunk < chunk_range[0] || chunk >= chunk_range[1])) return;  // (workgroup-uniform)
    if (chunk_range) T = chunk_range[6];  // streaming: the frames of the window that exist so far (decided on the device)
    const int nchunks = gridDim.x;

    double *slab0 = reinterpret_cast<double *>(smem);  // [Kp][32], halves swapped in odd rows
    double *U = slab0 + Kp * GEN_COLS;                 // slab buffer 1 | nir table + int8 spike tile (LIF phase only)
    double *red = U + u_doubles;                       // [2][BF_WAVES][32]
    double *accl = red + 2 * BF_WAVES * GEN_COLS;      // [Gp + 32] column sums of the chunk (NSUB > 1)
    double *ntab = U;
    const int ntab_len = SRC_SPIKES ? 4 * NK + 16 : 0;
    int8_t *spk = reinterpret_cast<int8_t *>(U + ntab_len);
    const int R = SUB + 4 * NK - 16;

    // Slab staging by LDS-DMA (global_load_lds_dwordx4: no staging registers -- with 64 VGPRs of membrane fragments at 128
    // channels there are none to spare -- and no ds_write pass).  One wave-instruction lands 64 x 16 B = 4 slab rows
    // contiguously; the destination is lane-linear, so the swap of the 16-column halves in odd rows is applied to the SOURCE
    // address (lane p of an odd row fetches pair p ^ 8) and again by the B-fragment reads below.
    constexpr int NI = Kp / 4;  // wave-instructions per slab, dealt round-robin over the 8 waves
    auto stage_slab = [&](int sl, int buf) {
        double *sb = buf ? U : slab0;
        const int rq = l >> 4, p = l & 15;
#pragma unroll
        for (int j0 = 0; j0 < NI; j0 += BF_WAVES) {
            const int j = j0 + wv;
            if (j < NI) {  // (wave-uniform)
                const int row = 4 * j + rq;
                // (a last, half-empty slab of an odd tile count reads 16 columns past the row: the next row's head or the
                //  allocation's 16 doubles of slack -- columns nobody stores)
                const double *src = Wp + (size_t)row * Gp + GEN_COLS * sl + 2 * (p ^ ((rq & 1) << 3));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(sb + (size_t)j * 128), 16, 0, 0);
            }
        }
    };
    auto slabs_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    double *pout = partial ? partial + ((size_t)b * nchunks + chunk) * Gp : nullptr;
    const int Ghp = Gp >> 1;

    for (int s = 0; s < NSUB; ++s) {
        const int cs = chunk * CH + s * SUB;
        if (s > 0 && cs >= T) break;  // (workgroup-uniform) the rest of a trial's last chunk is empty
        stage_slab(0, 0);
        if (SRC_SPIKES) {
            for (int e = tid; e < ntab_len; e += BF_THREADS) ntab[e] = ntab_g[e];
            const int8_t *sb = spikes + (size_t)b * (chunk_range ? chunk_range[3] : T) * C;  // ([3]: frames per trial of a raster window)
            const int tau0 = cs + 16 - 4 * NK;
            if (vec_stage && tau0 >= 0 && tau0 + R <= T) {
                // interior sub-chunk, C == 16 CT, 16-byte aligned raster: the tile is one contiguous range of the trial
                const uint4 *src = reinterpret_cast<const uint4 *>(sb + (size_t)tau0 * C);
                uint4 *dst = reinterpret_cast<uint4 *>(spk);
                const int n16 = R * (Cs / 16);
                for (int e0 = tid; e0 < n16; e0 += BF_THREADS * 4) {
                    uint4 v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = e0 + i * BF_THREADS;
                        v[i] = src[e < n16 ? e : n16 - 1];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = e0 + i * BF_THREADS;
                        if (e < n16) dst[e] = v[i];
                    }
                }
            } else {
                for (int e0 = tid; e0 < R * Cs; e0 += BF_THREADS * 8) {
                    int8_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = e0 + i * BF_THREADS;
                        const int rho = e / Cs, c = e % Cs;
                        int tau = tau0 + rho;
                        tau = tau < 0 ? 0 : (tau >= T ? T - 1 : tau);
                        v[i] = sb[(size_t)tau * C + (c < C ? c : C - 1)];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int e = e0 + i * BF_THREADS;
                        const int rho = e / Cs, c = e % Cs;
                        const int tau = tau0 + rho;
                        if (e < R * Cs) spk[e] = (c < C && tau >= 0 && tau < T) ? v[i] : (int8_t)0;
                    }
                }
            }
        }
        slabs_landed();
        __syncthreads();

        // ---- membrane fragments of this wave's NT time tiles ----
        const int tb0 = cs + wv * NT * 16;
        double4_t Vf[NT][CT];
        if (SRC_SPIKES) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                double4_t acc[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};
                if (tb0 < T) {  // wave-uniform
                    // two k-steps per trip, the operands of k-step k + 1 / k + 2 requested in front of the matrix instruction of k-step
                    // k / k + 1 into the OTHER register set (no copies), both pointers stepping once per trip with non-negative immediate
                    // offsets: 1 conversion + ~1.5 address instructions per k-step (the one-step loop with its carried copies: 5 -- and
                    // every vector instruction between two matrix instructions costs the pipe its switch, tools/mfma_valu_overlap.hip).
                    // The same products in the same order.
                    const int8_t *sp = spk + (size_t)(tb0 - cs + q) * Cs + 16 * ct + lc;  // row of k-step 0 (walks up, 8 rows per trip)
                    const double *nq = ntab + (lc - q + 4 * NK - 16 + 15) - 8;            // tap of k-step 2 (walks down, 8 taps per trip)
                    int a0[NT], a1[NT];
                    double b0 = nq[8], b1;
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) a0[tt] = sp[(size_t)(16 * tt) * Cs];
                    int ks = 0;
                    for (; ks + 2 <= NK; ks += 2) {
                        b1 = nq[4];
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) a1[tt] = sp[(size_t)(16 * tt + 4) * Cs];
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
                            acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a0[tt], b0, acc[tt], 0, 0, 0);
                        if (ks + 2 < NK) {
                            b0 = nq[0];
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt) a0[tt] = sp[(size_t)(16 * tt + 8) * Cs];
                        }
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
                            acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a1[tt], b1, acc[tt], 0, 0, 0);
                        sp += (size_t)8 * Cs;
                        nq -= 8;
                    }
                    if (ks < NK) {  // (an odd number of k-steps: the last one's operands are in set 0)
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
                            acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a0[tt], b0, acc[tt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const bool tvalid = (tb0 + 16 * tt + lc) < T;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[tt][r] = tvalid ? acc[tt][r] : 0.0;
                    Vf[tt][ct] = acc[tt];
                }
            }
            __syncthreads();  // every wave is done with the spike tile and the nir table: slab buffer 1 may overwrite them
        } else {
            // planar source: channel c = 16 ct + 4 r + q of frame tb + lc -- one running pointer that steps 4 rows per load (32 loads
            // with their own 64-bit addresses each were the 148-244 B of scratch this branch had at 128 channels)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int tb = tb0 + 16 * tt;
                const bool tvalid = (tb + lc) < T;
                const double *pp = pre + ((size_t)b * C + q) * Ts + (tvalid ? tb + lc : 0);
                const size_t step = (size_t)4 * Ts;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    double4_t acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * ct + 4 * r + q;
                        double v = 0.0;
                        if (tvalid && c < C) v = *pp;
                        acc[r] = v;
                        pp += step;
                    }
                    Vf[tt][ct] = acc;
                }
            }
        }

        // ---- the DoA slabs ----
        // B fragment of k-step k, tile gl: row 4k + q, column 16 gl + lc, halves swapped in odd rows (q & 1)
        const int boff0 = q * GEN_COLS + (lc ^ ((q & 1) << 4));
        for (int sl = 0; sl < NSL; ++sl) {
            const int buf = sl & 1;
            const double *sbuf = buf ? U : slab0;
            if (sl + 1 < NSL) stage_slab(sl + 1, buf ^ 1);  // in flight behind this slab's MFMAs (the other buffer was released by the last barrier)
            double sq2[2];
#pragma unroll
            for (int gl = 0; gl < 2; ++gl) {
                // one 16-column tile at a time: one accumulator set and one stream of B fragments (both tiles at once cost 16 more
                // registers than the 128-channel instantiation has: scratch in the slab loop's prologue)
                double sq = 0.0;
                if (tb0 < T) {  // (wave-uniform; a wave beyond the end of the trial contributes nothing and stores nothing)
                    double4_t acc[NT];
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) acc[tt] = double4_t{0.0, 0.0, 0.0, 0.0};  // (folds into the first MFMA's zero C operand)
                    // B fragments D k-steps ahead of their MFMAs, the order pinned: left alone the scheduler hoists as many of the
                    // KS reads as the register budget holds -- and at 128 channels (64 VGPRs of membrane fragments) then spills
                    const double *w0 = sbuf + (gl ? (boff0 ^ 16) : boff0);
                    constexpr int D = WANT_Y ? 1 : 3;  // (with y stored the launch is HBM-write bound and short of registers: one read ahead)
                    double wk[KS];
#pragma unroll
                    for (int k = 0; k < D && k < KS; ++k) wk[k] = w0[4 * k * GEN_COLS];
                    __builtin_amdgcn_sched_group_barrier(0x100, D < KS ? D : KS, 0);
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        if (k + D < KS) {
                            wk[k + D] = w0[4 * (k + D) * GEN_COLS];
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
                            acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vf[tt][k >> 2][k & 3], wk[k], acc[tt], 0, 0, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                    }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sq = __builtin_fma(acc[tt][r], acc[tt][r], sq);
                if (WANT_Y) {
                    // (the planar source is the complex Beamformer's: y interleaved (re, im), columns [0, Ghp) real, [Ghp, Gp) imaginary;
                    //  the spike source is the SNN beamformer's real y) -- one row pointer per time tile, stepping four rows per store
                    const int gcol = 16 * (2 * sl + gl) + lc;
                    const int part = gcol >= Ghp;
                    const int gc = gcol - (part ? Ghp : 0);
                    const bool colok = SRC_SPIKES ? gcol < G : (gcol < Gp && gc < (G >> 1));
                    const size_t rowlen = (size_t)G;  // doubles per row of y (complex: G counts real columns = 2 x the DoA count)
                    const size_t coloff = SRC_SPIKES ? (size_t)gcol : (size_t)2 * gc + part;
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const int t0r = tb0 + 16 * tt + q;
                        double *yp = y + ((size_t)b * T + t0r) * rowlen + coloff;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (colok && t0r + 4 * r < T) *yp = acc[tt][r];
                            yp += 4 * rowlen;
                        }
                    }
                }
                }
                sq2[gl] = sq;
            }
            {
                // the column sums of both tiles over the wave's rows in one paired exchange: lanes 0-15 tile 0, lanes 16-31 tile 1
                const double cs2 = row_sum4_pair(sq2[0], sq2[1]);
                if (l < GEN_COLS) red[(buf * BF_WAVES + wv) * GEN_COLS + l] = cs2;
            }
            slabs_landed();
            __syncthreads();
            if (pout && tid < GEN_COLS) {
                double sm = 0.0;
#pragma unroll
                for (int w8 = 0; w8 < BF_WAVES; ++w8) sm += red[(buf * BF_WAVES + w8) * GEN_COLS + tid];
                if (NSUB > 1)
                    accl[GEN_COLS * sl + tid] = s > 0 ? accl[GEN_COLS * sl + tid] + sm : sm;  // (the same thread every time)
                else if (GEN_COLS * sl + tid < Gp)
                    pout[GEN_COLS * sl + tid] = sm;
            }
        }
    }
    if (NSUB > 1 && pout) {
        __syncthreads();
        for (int g = tid; g < Gp; g += BF_THREADS) pout[g] = accl[g];
    }
}

template <int CT, bool SRC_SPIKES>
static hipError_t launch_gen(const BeamformW &W, const NeuronTab *nt, const int8_t *spikes, const double *pre, int B,
                             int T, int Ts, double *y, int y_complex, double *partial, hipStream_t stream)
{
    const int NK = SRC_SPIKES ? nt->NK : 0;
    const int Gp = 16 * W.GT;
    const GenLds L = gen_lds(CT, NK, SRC_SPIKES, Gp);
    if (L.total_bytes > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid((T + gen_chunk(CT) - 1) / gen_chunk(CT), B), block(BF_THREADS);
    const double *tab = SRC_SPIKES ? nt->tab : nullptr;
    const int vec_stage = SRC_SPIKES && W.C == 16 * CT && (reinterpret_cast<uintptr_t>(spikes) & 15) == 0;
    hipError_t e;
#define GEN_LAUNCH(WY)                                                                                              \
    do {                                                                                                            \
        auto k = &beamform_gen_kernel<CT, SRC_SPIKES, WY>;                                                          \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                160 * 1024);                                                                        \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL(k, grid, block, L.total_bytes, stream, spikes, pre, tab, NK, W.Wp, W.GT, W.C, W.G, T, Ts, y, \
                           y_complex, partial, W.chunk_range, (int)L.u_doubles, vec_stage);                         \
    } while (0)
    if (y)
        GEN_LAUNCH(true);
    else
        GEN_LAUNCH(false);
#undef GEN_LAUNCH
    return hipGetLastError();
}

template <bool SRC_SPIKES>
static hipError_t dispatch_ct(const BeamformW &W, const NeuronTab *nt, const int8_t *spikes, const double *pre, int B,
                              int T, int Ts, double *y, int y_complex, double *partial, hipStream_t stream)
{
    switch (W.CT) {
        case 1: return launch_gen<1, SRC_SPIKES>(W, nt, spikes, pre, B, T, Ts, y, y_complex, partial, stream);
        case 2: return launch_gen<2, SRC_SPIKES>(W, nt, spikes, pre, B, T, Ts, y, y_complex, partial, stream);
        case 3: return launch_gen<3, SRC_SPIKES>(W, nt, spikes, pre, B, T, Ts, y, y_complex, partial, stream);
        case 4: return launch_gen<4, SRC_SPIKES>(W, nt, spikes, pre, B, T, Ts, y, y_complex, partial, stream);
        case 8: return launch_gen<8, SRC_SPIKES>(W, nt, spikes, pre, B, T, Ts, y, y_complex, partial, stream);
        default: return hipErrorInvalidValue;
    }
}


hipError_t launch_lif_beamform(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T, double *y,
                               double *partial, hipStream_t stream, int *nchunks)
{
    if ((partial || y) && ws_eligible(W, nt, y != nullptr)) return launch_ws(W, nt, spikes, B, T, partial, y, stream, nchunks);
    *nchunks = beamform_nchunks_ct(T, W.CT);
    return dispatch_ct<true>(W, &nt, spikes, nullptr, B, T, 0, y, 0, partial, stream);
}

int lif_beamform_chunk_frames(const BeamformW &W, const NeuronTab &nt)
{
    if (ws_eligible(W, nt, false)) return BF_WAVES * WS_NT * 16;
    return W.CT > 4 ? gen_chunk(8) : gen_chunk(1);
}

hipError_t launch_planar_beamform(const BeamformW &W, const double *pre, int B, int T, int Ts, double *y,
                                  int y_complex, double *partial, hipStream_t stream, int *nchunks)
{
    if (y_complex && wsc_eligible(W)) {  // up to 8 microphones: the bf_mat-stationary form
        *nchunks = (T + WSC_CHUNK - 1) / WSC_CHUNK;
        return launch_wsc(W, pre, B, T, Ts, y, partial, stream);
    }
    *nchunks = beamform_nchunks_ct(T, W.CT);
    return dispatch_ct<false>(W, nullptr, nullptr, pre, B, T, Ts, y, y_complex, partial, stream);
}

// ---- chunk reduction, mean over time, arg-max (first maximum, like np.argmax) --------------------------------
// The order of the time reduction is fixed and STREAMABLE: chunk sums are added in ascending order inside blocks of
// PA_BLOCK consecutive chunks, block sums in ascending order onto the total.  A recording that arrives tile by tile can
// apply exactly this order with O(1) state (stream_accumulate_kernel below) and ends with the bits of the one-shot call.
// Up to PA_BLOCK chunks (config 2: 19) that is the plain ascending sum.  Long recordings (speech: 1298 chunks, 125 trials)
// would leave one thread per DoA latency bound, so with S = 4 four blocks are summed side by side and then added to the
// total in their order -- the same additions, S of them in flight.
constexpr int PA_BLOCK = STREAM_BLOCK_CHUNKS;

template <int S>
__global__ __launch_bounds__(256 * S) void power_argmax_kernel(const double *__restrict__ partial, int T, int nchunks,
                                                                int Gp, int G, int complex_pairs, int Ghp,
                                                                double *__restrict__ power, int32_t *__restrict__ argmax)
{
    constexpr int U = S > 1 ? 4 : 1;  // blocks per slice and round: S * U block sums in flight between two barriers
    __shared__ double sv[256];
    __shared__ int si[256];
    __shared__ double ps[U][S][256];
    const int b = blockIdx.x;
    const int col = threadIdx.x & 255;
    const int slice = threadIdx.x >> 8;
    const double *pb = partial + (size_t)b * nchunks * Gp;
    const int nblocks = (nchunks + PA_BLOCK - 1) / PA_BLOCK;
    double best = -1.0;
    int bi = 0x7fffffff;
    for (int g0 = 0; g0 < G; g0 += 256) {
        const int g = g0 + col;
        double total = 0.0;
        for (int blk0 = 0; blk0 < nblocks; blk0 += S * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int blk = blk0 + S * u + slice;
                double s = 0.0;
                if (g < G && blk < nblocks) {
                    const int c_hi = (blk + 1) * PA_BLOCK < nchunks ? (blk + 1) * PA_BLOCK : nchunks;
                    for (int ch = blk * PA_BLOCK; ch < c_hi; ++ch) {
                        s += pb[(size_t)ch * Gp + g];
                        if (complex_pairs) s += pb[(size_t)ch * Gp + Ghp + g];
                    }
                }
                if (S > 1)
                    ps[u][slice][col] = s;
                else
                    total += s;
            }
            if (S > 1) {
                __syncthreads();
                if (slice == 0) {
                    // the block sums onto the total in ascending block order: blk0 + S u + v
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int v = 0; v < S; ++v)
                            if (blk0 + S * u + v < nblocks) total += ps[u][v][col];
                }
                __syncthreads();
            }
        }
        if (slice == 0 && g < G) {
            const double p = total / (double)T;
            if (power) power[(size_t)b * G + g] = p;
            if (p > best) {
                best = p;
                bi = g;
            }
        }
    }
    if (slice == 0) {
        sv[col] = best;
        si[col] = bi;
    }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (slice == 0 && col < s) {
            const double ov = sv[col + s];
            const int oi = si[col + s];
            if (ov > sv[col] || (ov == sv[col] && oi < si[col])) {
                sv[col] = ov;
                si[col] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && argmax) argmax[b] = si[0] == 0x7fffffff ? 0 : si[0];
}

// Long recordings (speech: 1298 chunk rows of 368 doubles per trial, 125 trials = 478 MB of partial sums): one workgroup per trial
// leaves half the chip idle and every thread with a chain of dependent row reads (0.33 ms per launch, 1.4 TB/s).  The reduction is
// independent per DoA column, so the columns of a trial are split over workgroups of 64 columns x 4 slices (the SAME order of
// additions per column: chunk sums ascending inside blocks of PA_BLOCK, block sums ascending onto the total), and a second, tiny
// kernel takes the first maximum of every trial's row of `power`.
__global__ __launch_bounds__(256) void power_columns_kernel(const double *__restrict__ partial, int T, int nchunks, int Gp, int G,
                                                             int complex_pairs, int Ghp, double *__restrict__ power)
{
    constexpr int S = 4, U = 4, W = 64;
    __shared__ double ps[U][S][W];
    const int b = blockIdx.y;
    const int col = threadIdx.x & (W - 1);
    const int slice = threadIdx.x >> 6;
    const int g = blockIdx.x * W + col;
    const double *pb = partial + (size_t)b * nchunks * Gp;
    const int nblocks = (nchunks + PA_BLOCK - 1) / PA_BLOCK;
    double total = 0.0;
    for (int blk0 = 0; blk0 < nblocks; blk0 += S * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int blk = blk0 + S * u + slice;
            double s = 0.0;
            if (g < G && blk < nblocks) {
                const int c_hi = (blk + 1) * PA_BLOCK < nchunks ? (blk + 1) * PA_BLOCK : nchunks;
                for (int ch = blk * PA_BLOCK; ch < c_hi; ++ch) {
                    s += pb[(size_t)ch * Gp + g];
                    if (complex_pairs) s += pb[(size_t)ch * Gp + Ghp + g];
                }
            }
            ps[u][slice][col] = s;
        }
        __syncthreads();
        if (slice == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < S; ++v)
                    if (blk0 + S * u + v < nblocks) total += ps[u][v][col];
        }
        __syncthreads();
    }
    if (slice == 0 && g < G) power[(size_t)b * G + g] = total / (double)T;
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const double *__restrict__ power, int G, int32_t *__restrict__ argmax)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x, col = threadIdx.x;
    double best = -1.0;
    int bi = 0x7fffffff;
    for (int g = col; g < G; g += 256) {  // ascending per thread: its first maximum
        const double p = power[(size_t)b * G + g];
        if (p > best) {
            best = p;
            bi = g;
        }
    }
    sv[col] = best;
    si[col] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (col < s) {
            const double ov = sv[col + s];
            const int oi = si[col + s];
            if (ov > sv[col] || (ov == sv[col] && oi < si[col])) {
                sv[col] = ov;
                si[col] = oi;
            }
        }
        __syncthreads();
    }
    if (col == 0) argmax[b] = si[0] == 0x7fffffff ? 0 : si[0];
}

// ---- streaming: the same reduction with O(1) state -----------------------------------------------------------------
// acc [B][2][G]: running total and the sum of the open block; ctl: {chunks done, chunks of the open block}.  The chunk
// range [lo, hi) of this call (window-relative, decided on the device by the caller's horizon kernel) is read from
// `range`; rows lo .. hi - 1 of `partial` [B][nwin][Gp] are added in order.  power / argmax (may be NULL) = the mean over
// the `frames` frames beamformed so far (frames_ptr[0]; the last chunk of a recording may be ragged).
__global__ __launch_bounds__(256) void stream_accumulate_kernel(const double *__restrict__ partial, int nwin, int Gp, int G,
                                                                 const int *__restrict__ range, const int *__restrict__ ctl,
                                                                 double *__restrict__ acc, const int *__restrict__ frames_ptr,
                                                                 double *__restrict__ power, int32_t *__restrict__ argmax)
{
    __shared__ double sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x;
    const int col = threadIdx.x;
    const int lo = range[0], hi = range[1];
    const int open0 = ctl[1];  // chunks already in the open block
    const double *pb = partial + (size_t)b * nwin * Gp;
    double *tot = acc + (size_t)b * 2 * G, *blk = tot + G;
    const double frames = (double)frames_ptr[0];
    double best = -1.0;
    int bi = 0x7fffffff;
    for (int g = col; g < G; g += 256) {
        double total = tot[g], s = blk[g];
        int open = open0;
        for (int ch = lo; ch < hi; ++ch) {
            s += pb[(size_t)ch * Gp + g];
            if (++open == PA_BLOCK) {
                total += s;
                s = 0.0;
                open = 0;
            }
        }
        tot[g] = total;
        blk[g] = s;
        // the value the one-shot reduction would hold after these chunks: the open block is added last (if it holds any)
        const double now = open > 0 ? total + s : total;
        const double p = frames > 0.0 ? now / frames : 0.0;
        if (power) power[(size_t)b * G + g] = p;
        if (p > best) {
            best = p;
            bi = g;
        }
    }
    sv[col] = best;
    si[col] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (col < s) {
            const double ov = sv[col + s];
            const int oi = si[col + s];
            if (ov > sv[col] || (ov == sv[col] && oi < si[col])) {
                sv[col] = ov;
                si[col] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && argmax) argmax[b] = si[0] == 0x7fffffff ? 0 : si[0];
}

hipError_t launch_stream_accumulate(const double *partial, int B, int nwin, int Gp, int G, const int *range, const int *ctl,
                                    double *acc, const int *frames_ptr, double *power, int32_t *argmax, hipStream_t stream)
{
    hipLaunchKernelGGL(stream_accumulate_kernel, dim3(B), dim3(256), 0, stream, partial, nwin, Gp, G, range, ctl, acc, frames_ptr, power,
                       argmax);
    return hipGetLastError();
}

hipError_t launch_power_argmax(const double *partial, int B, int T, int nchunks, int Gp, int G, int complex_pairs,
                               int Ghalf_pad, double *power, int32_t *argmax, hipStream_t stream)
{
    if (nchunks >= 128 && power && argmax && B <= 65535) {
        hipLaunchKernelGGL(power_columns_kernel, dim3((G + 63) / 64, B), dim3(256), 0, stream, partial, T, nchunks, Gp, G, complex_pairs,
                           Ghalf_pad, power);
        hipLaunchKernelGGL(argmax_rows_kernel, dim3(B), dim3(256), 0, stream, power, G, argmax);
    } else if (nchunks >= 128)
        hipLaunchKernelGGL(power_argmax_kernel<4>, dim3(B), dim3(1024), 0, stream, partial, T, nchunks, Gp, G, complex_pairs, Ghalf_pad,
                           power, argmax);
    else
        hipLaunchKernelGGL(power_argmax_kernel<1>, dim3(B), dim3(256), 0, stream, partial, T, nchunks, Gp, G, complex_pairs, Ghalf_pad,
                           power, argmax);
    return hipGetLastError();
}

}  // namespace micloc
