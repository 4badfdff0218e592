// Band-pass IIR + robust zero-crossing (RZCC) spike encoder for gfx950.
// Reference: micloc/snn_beamformer.py:330-338 (lfilter(b, a, .) then spk_encoder.evolve) and
// micloc/spike_encoder.py:115-137 (cumsum -> scipy.signal.find_peaks(distance=robust_width)).
//
// The encoder contract is bit-exact, and both the DF2T recurrence and np.cumsum are strictly
// sequential in time, so the time axis is NOT parallelised: one lane owns one (trial, channel) stream
// and walks it with the exact operation order of oracle/micloc_oracle.c; parallelism comes from the
// B x 2M independent streams (one wave = 64 streams, one wave per workgroup so the streams spread
// over all CUs).  Input is the planar [stream][Ts] layout written by the STHT kernel, so each lane
// reads its own 64-byte line per 8 steps (16-byte vector loads, next chunk prefetched).
//
// Phase 1 (stream): y = IIR(x); c += y; local maxima / minima of c with scipy's plateau rule
//   (_local_maxima_1d: strict rise, flat run, strict fall -> midpoint, edges never peaks) are appended
//   to per-lane candidate lists kept slot-major in scratch so the lock-step appends/reads coalesce.
// Phase 2 (select): scipy's _select_by_peak_distance is a greedy by descending priority; peaks only
//   interact within `distance`, so the list splits into independent clusters at every gap >= distance
//   and the greedy is run per cluster (equal priority: the later peak wins, i.e. a stable sort).
//   Kept peaks are scattered as +1 / -1 bytes into the zero-initialised [B][T][C] spike tensor.
#include "micloc_internal.h"

namespace micloc {

constexpr int RZ_CHUNK = 8;

struct PeakState {
    int left;   // start of the current candidate plateau, -1 if none
    double pv;  // its value
};

template <int N>
__global__ __launch_bounds__(64) void bandpass_rzcc_kernel(const double *__restrict__ h, double *__restrict__ pre,
                                                            int8_t *__restrict__ spikes, int *__restrict__ plist,
                                                            double *__restrict__ vlist, IirCoef coef, int nlanes,
                                                            int C, int T, int Ts, int w, int bipolar, int cap)
{
    const int lane_g = blockIdx.x * 64 + threadIdx.x;
    if (lane_g >= nlanes) return;
    const double *src = h + (size_t)lane_g * Ts;
    double *dst = pre ? pre + (size_t)lane_g * Ts : nullptr;

    double z[N > 1 ? N - 1 : 1];
#pragma unroll
    for (int i = 0; i < (N > 1 ? N - 1 : 1); ++i) z[i] = 0.0;

    double c = 0.0, prev = 0.0;
    PeakState smax{-1, 0.0}, smin{-1, 0.0};
    int cnt_max = 0, cnt_min = 0;
    const size_t pol_stride = (size_t)cap * nlanes;
    const bool want_spikes = spikes != nullptr;

    double2 nxt[RZ_CHUNK / 2];
#pragma unroll
    for (int i = 0; i < RZ_CHUNK / 2; ++i) nxt[i] = reinterpret_cast<const double2 *>(src)[i];

    for (int t0 = 0; t0 < T; t0 += RZ_CHUNK) {
        double xv[RZ_CHUNK], yv[RZ_CHUNK];
#pragma unroll
        for (int i = 0; i < RZ_CHUNK / 2; ++i) {
            xv[2 * i] = nxt[i].x;
            xv[2 * i + 1] = nxt[i].y;
        }
        if (t0 + RZ_CHUNK < T) {
#pragma unroll
            for (int i = 0; i < RZ_CHUNK / 2; ++i)
                nxt[i] = reinterpret_cast<const double2 *>(src + t0 + RZ_CHUNK)[i];
        }
#pragma unroll
        for (int jj = 0; jj < RZ_CHUNK; ++jj) {
            const int t = t0 + jj;
            if (t < T) {  // wave-uniform
                const double xin = xv[jj];
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
                yv[jj] = y;
                if (want_spikes) {
                    c = c + y;
                    if (t > 0) {
                        // ---- local maxima of c (scipy _local_maxima_1d, streaming form) ----
                        if (smax.left >= 0) {
                            if (c < smax.pv) {
                                if (cnt_max < cap) {
                                    plist[(size_t)cnt_max * nlanes + lane_g] = (smax.left + t - 1) >> 1;
                                    vlist[(size_t)cnt_max * nlanes + lane_g] = smax.pv;
                                }
                                ++cnt_max;
                                smax.left = -1;
                            } else if (c > smax.pv) {
                                smax.left = t;
                                smax.pv = c;
                            }
                        } else if (prev < c) {
                            smax.left = t;
                            smax.pv = c;
                        }
                        // ---- local minima of c == local maxima of -c ----
                        if (bipolar) {
                            if (smin.left >= 0) {
                                if (c > smin.pv) {
                                    if (cnt_min < cap) {
                                        plist[pol_stride + (size_t)cnt_min * nlanes + lane_g] = (smin.left + t - 1) >> 1;
                                        vlist[pol_stride + (size_t)cnt_min * nlanes + lane_g] = -smin.pv;
                                    }
                                    ++cnt_min;
                                    smin.left = -1;
                                } else if (c < smin.pv) {
                                    smin.left = t;
                                    smin.pv = c;
                                }
                            } else if (prev > c) {
                                smin.left = t;
                                smin.pv = c;
                            }
                        }
                    }
                    prev = c;
                }
            } else {
                yv[jj] = 0.0;
            }
        }
        if (dst) {
#pragma unroll
            for (int i = 0; i < RZ_CHUNK / 2; ++i)
                reinterpret_cast<double2 *>(dst + t0)[i] = make_double2(yv[2 * i], yv[2 * i + 1]);
        }
    }
    if (!want_spikes) return;

    // ---- phase 2: min-distance selection, cluster by cluster -----------------------------------------
    const int b = lane_g / C;
    const int ch = lane_g - b * C;
    int8_t *sp = spikes + (size_t)b * T * C + ch;
    for (int pol = 0; pol < (bipolar ? 2 : 1); ++pol) {
        int *P = plist + pol * pol_stride + lane_g;
        const double *V = vlist + pol * pol_stride + lane_g;
        int n = pol ? cnt_min : cnt_max;
        if (n > cap) n = cap;  // cannot happen: peaks are >= 2 samples apart and cap = T/2 + 1
        const int8_t mark = pol ? -1 : 1;
        int s = 0;
        int plast = n > 0 ? P[0] : 0;
        for (int i = 1; i <= n; ++i) {
            int pi = 0;
            bool closes = true;
            if (i < n) {
                pi = P[(size_t)i * nlanes];
                closes = (pi - plast) >= w;
            }
            if (closes) {
                const int e = i;
                if (e - s == 1) {
                    sp[(size_t)plast * C] = mark;
                } else {
                    // greedy by descending priority inside the cluster [s, e); a decided entry is
                    // flagged by complementing its position (positions are >= 0).
                    int remaining = e - s;
                    while (remaining > 0) {
                        int best = -1;
                        double bv = 0.0;
                        for (int k = s; k < e; ++k) {
                            const int pk = P[(size_t)k * nlanes];
                            if (pk < 0) continue;
                            const double vk = V[(size_t)k * nlanes];
                            if (best < 0 || vk >= bv) {  // >= : equal priority -> later index wins
                                best = k;
                                bv = vk;
                            }
                        }
                        const int pb = P[(size_t)best * nlanes];
                        sp[(size_t)pb * C] = mark;
                        P[(size_t)best * nlanes] = ~pb;
                        --remaining;
                        for (int k = best - 1; k >= s; --k) {
                            int pk = P[(size_t)k * nlanes];
                            const int pk_abs = pk < 0 ? ~pk : pk;
                            if (pb - pk_abs >= w) break;
                            if (pk >= 0) {
                                P[(size_t)k * nlanes] = ~pk;
                                --remaining;
                            }
                        }
                        for (int k = best + 1; k < e; ++k) {
                            int pk = P[(size_t)k * nlanes];
                            const int pk_abs = pk < 0 ? ~pk : pk;
                            if (pk_abs - pb >= w) break;
                            if (pk >= 0) {
                                P[(size_t)k * nlanes] = ~pk;
                                --remaining;
                            }
                        }
                    }
                }
                s = e;
            }
            plast = pi;
        }
    }
}

size_t rzcc_scratch_bytes(int nlanes, int T)
{
    const size_t cap = (size_t)T / 2 + 1;
    size_t bytes = 2 * cap * (size_t)nlanes * sizeof(double);  // vlist
    bytes += 2 * cap * (size_t)nlanes * sizeof(int);           // plist
    return (bytes + 255) & ~(size_t)255;
}

template <int N>
static void launch_rz(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts, int w, int bipolar,
                      double *pre, int8_t *spikes, int *plist, double *vlist, int cap, hipStream_t stream)
{
    dim3 grid((nlanes + 63) / 64), block(64);
    hipLaunchKernelGGL(bandpass_rzcc_kernel<N>, grid, block, 0, stream, h, pre, spikes, plist, vlist, coef, nlanes,
                       C, T, Ts, w, bipolar, cap);
}

hipError_t launch_bandpass_rzcc(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts,
                                int robust_width, int bipolar, double *pre, int8_t *spikes, void *scratch,
                                hipStream_t stream)
{
    const int cap = T / 2 + 1;
    double *vlist = reinterpret_cast<double *>(scratch);
    int *plist = reinterpret_cast<int *>(vlist + 2 * (size_t)cap * nlanes);
    if (spikes) {
        hipError_t e = hipMemsetAsync(spikes, 0, (size_t)nlanes * T, stream);
        if (e != hipSuccess) return e;
    }
#define RZ_CASE(NN)                                                                                              \
    case NN:                                                                                                     \
        launch_rz<NN>(coef, h, nlanes, C, T, Ts, robust_width, bipolar, pre, spikes, plist, vlist, cap, stream); \
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
