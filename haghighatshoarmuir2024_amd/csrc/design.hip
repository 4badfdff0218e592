// Beamforming vectors from membrane covariances, batched over the DoA grid: the decomposition step of
// SNNBeamformer.design_from_template (reference micloc/snn_beamformer.py:176-203) and _find_dc_removed_sing_vec
// (:372-422), which the reference hands to LAPACK (np.linalg.svd) once per DoA.
//
//   unipolar  (:183-186)  U, D = svd(C);  theta = U^T 1;  root of sum_i theta_i^2 / (D_i - u) on (D_1, D_0) by bisection
//                         to rel_prec;  w = U (theta / (D - root)) / |.|                      -- independent of the signs of U
//   bipolar   (:191-203)  C_comp = (C11 + C22)/2 + 1j (C12 + C21^T)/2;  U = svd(C_comp)[0];  w = [Re U[:,0]; Im U[:,0]]
//
// C <= 32: one wave per DoA.  The symmetric eigenproblem (n <= 32) is solved by cyclic Jacobi rotations with the matrix and the
// accumulated rotations in LDS (lane k owns row k); the left singular vectors of the complex d x d matrix C_comp are the
// eigenvectors of the Hermitian C_comp C_comp^H = P + jQ, obtained from its real embedding [[P, -Q], [Q, P]] (every
// eigenvalue appears twice, the two eigenvectors [x; y], [-y; x] being u and j u).  A singular vector is defined up to a
// unit phase; the beam pattern |W^H W| does not depend on it, but the real-projected spectrum does a little (see the phase
// convention in the kernel): the first component is made real and negative, which is what LAPACK leaves on these matrices.
#include "micloc_internal.h"

namespace micloc {

constexpr int DS_N = 32;        // largest matrix order
constexpr int DS_LD = DS_N + 1; // padded row

__device__ __forceinline__ void jacobi_eig(double (*A)[DS_LD], double (*V)[DS_LD], int n, int lane)
{
    // V = I
    for (int j = 0; j < n; ++j)
        if (lane < n) V[lane][j] = lane == j ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 40; ++sweep) {
        // off-diagonal mass against the diagonal (every lane computes the same number: no broadcast needed)
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double a = A[i][j];
                if (i == j)
                    dg = __builtin_fma(a, a, dg);
                else
                    off = __builtin_fma(a, a, off);
            }
        if (off <= 1e-30 * dg || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p][q];
                if (apq != 0.0) {  // uniform
                    const double app = A[p][p], aqq = A[q][q];
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                    __syncthreads();
                    double akp = 0.0, akq = 0.0;
                    if (lane < n) {
                        akp = A[lane][p];
                        akq = A[lane][q];
                        const double vkp = V[lane][p], vkq = V[lane][q];
                        V[lane][p] = c * vkp - s * vkq;
                        V[lane][q] = s * vkp + c * vkq;
                    }
                    __syncthreads();
                    if (lane < n && lane != p && lane != q) {
                        const double nkp = c * akp - s * akq, nkq = s * akp + c * akq;
                        A[lane][p] = nkp;
                        A[p][lane] = nkp;
                        A[lane][q] = nkq;
                        A[q][lane] = nkq;
                    }
                    if (lane == 0) {
                        A[p][p] = app - t * apq;
                        A[q][q] = aqq + t * apq;
                        A[p][q] = 0.0;
                        A[q][p] = 0.0;
                    }
                    __syncthreads();
                }
            }
    }
    __syncthreads();
}

// cov [n_doa][C][C] (row-major, symmetric) -> column g0 + blockIdx.x of bf [C][G]
__global__ __launch_bounds__(64) void design_vec_kernel(const double *__restrict__ cov, int C, int bipolar, double rel_prec,
                                                         double *__restrict__ bf, int G, int g0)
{
    __shared__ double A[DS_N][DS_LD];
    __shared__ double V[DS_N][DS_LD];
    __shared__ double P[DS_N / 2][DS_LD], Q[DS_N / 2][DS_LD];
    __shared__ double Dv[DS_N], th[DS_N];
    __shared__ int order[DS_N];
    const int lane = threadIdx.x;
    const int n = C;
    const double *Cm = cov + (size_t)blockIdx.x * C * C;
    const int g = g0 + blockIdx.x;

    if (!bipolar) {
        for (int e = lane; e < n * n; e += 64) A[e / n][e % n] = 0.5 * (Cm[e] + Cm[(e % n) * n + e / n]);
    } else {
        const int d = C / 2;
        // C_comp = Pa + j Qb  (:195-199)
        for (int e = lane; e < d * d; e += 64) {
            const int i = e / d, j = e % d;
            P[i][j] = (Cm[(size_t)i * C + j] + Cm[(size_t)(d + i) * C + d + j]) / 2;
            Q[i][j] = (Cm[(size_t)i * C + d + j] + Cm[(size_t)(d + j) * C + i]) / 2;
        }
        __syncthreads();
        // H = C_comp C_comp^H = (Pa Pa^T + Qb Qb^T) + j (Qb Pa^T - Pa Qb^T); real embedding [[Re, -Im], [Im, Re]]
        for (int e = lane; e < d * d; e += 64) {
            const int i = e / d, j = e % d;
            double re = 0.0, im = 0.0;
            for (int k = 0; k < d; ++k) {
                re += P[i][k] * P[j][k] + Q[i][k] * Q[j][k];
                im += Q[i][k] * P[j][k] - P[i][k] * Q[j][k];
            }
            A[i][j] = re;
            A[d + i][d + j] = re;
            A[d + i][j] = im;
            A[i][d + j] = -im;
        }
    }
    __syncthreads();
    jacobi_eig(A, V, n, lane);
    if (lane < n) Dv[lane] = A[lane][lane];
    __syncthreads();
    // descending order of the eigenvalues (LAPACK's singular-value order); ties: lower index first
    if (lane < n) {
        int r = 0;
        for (int j = 0; j < n; ++j) r += (Dv[j] > Dv[lane]) || (Dv[j] == Dv[lane] && j < lane);
        order[r] = lane;
    }
    __syncthreads();

    if (!bipolar) {
        // theta = U^T 1 (:393), columns in descending-eigenvalue order
        if (lane < n) {
            const int col = order[lane];
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += V[k][col];
            th[lane] = s;
        }
        __syncthreads();
        double u_min = Dv[order[1]], u_max = Dv[order[0]];
        for (int it = 0; it < 200; ++it) {  // :399-411
            if ((u_max - u_min) / u_min < rel_prec) break;
            const double u_mid = (u_min + u_max) / 2;
            double val = 0.0;
            for (int i = 0; i < n; ++i) val += th[i] * th[i] / (Dv[order[i]] - u_mid);
            if (val < 0.0)
                u_min = u_mid;
            else
                u_max = u_mid;
        }
        const double root = (u_min + u_max) / 2.0;
        double w = 0.0;
        if (lane < n)
            for (int i = 0; i < n; ++i) w += V[lane][order[i]] * (th[i] / (Dv[order[i]] - root));  // :417
        __syncthreads();
        if (lane < n) th[lane] = w;
        __syncthreads();
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += th[i] * th[i];
        nrm = sqrt(nrm);
        if (lane < n) bf[(size_t)lane * G + g] = w / nrm;
    } else {
        const int d = C / 2;
        const int col = order[0];
        // u = x + j y with [x; y] the top eigenvector; rotate so that the largest component is real and positive
        int kmax = 0;
        double best = -1.0;
        for (int k = 0; k < d; ++k) {
            const double m2 = V[k][col] * V[k][col] + V[d + k][col] * V[d + k][col];
            if (m2 > best) {
                best = m2;
                kmax = k;
            }
        }
        // Phase convention.  The real projection y = Re(u^H z) of the bipolar beamformer is NOT invariant to the phase of u when z
        // is not perfectly analytic: over the reference's accuracy sweep a convention that jumps between DoAs (round 2: largest
        // component real and positive -- the largest component changes along the grid) moved 48 % of the arg-max decisions and
        // raised the MAE by 0.1-1.2 deg against the reference's matrix.  LAPACK's zgesdd leaves the FIRST component real and
        // negative (to ~2e-4 of its modulus on the reference's designs) -- a convention that is continuous along the grid; the same
        // one is used here (falling back to the largest component when the first one vanishes).
        const double m0 = V[0][col] * V[0][col] + V[d][col] * V[d][col];
        const int kref = m0 >= 1e-6 * best ? 0 : kmax;
        const double mag = sqrt(kref == 0 ? m0 : best);
        const double cr = -V[kref][col] / mag, ci = V[d + kref][col] / mag;  // -conj(u_k) / |u_k|: component k becomes real and negative
        double nrm = 0.0;
        for (int k = 0; k < n; ++k) nrm += V[k][col] * V[k][col];
        nrm = sqrt(nrm);
        if (lane < d) {
            const double x = V[lane][col] / nrm, y = V[d + lane][col] / nrm;
            bf[(size_t)lane * G + g] = x * cr - y * ci;
            bf[(size_t)(d + lane) * G + g] = x * ci + y * cr;
        }
    }
}

// ---- the same decompositions for 32 < C <= 128 (up to 64 microphones): one-sided Jacobi -----------------------------------
// A C x C matrix and its rotations no longer fit LDS twice.  Hestenes' one-sided form needs ONE matrix: rotate pairs of COLUMNS
// of G until all are mutually orthogonal, G J = U S -- the columns end up as the left singular vectors scaled by the singular
// values (for the symmetric positive semi-definite membrane covariance: its eigenvectors and eigenvalues, what np.linalg.svd
// returns at snn_beamformer.py:183).  The bipolar case decomposes the real embedding [[P, -Q], [Q, P]] of C_comp = P + jQ
// directly (no C_comp C_comp^H: no squared condition number): every singular value appears twice, its left vectors [x; y],
// [-y; x] being u and j u; any unit vector of that plane is u times a unit phase, which the phase convention removes.
// One workgroup of four waves per DoA; G is column-major in LDS (a wave reads a column with consecutive lanes: conflict-free);
// the n / 2 disjoint pairs of a round-robin step are shared out over the waves, a barrier between steps; the three dot
// products of a pair are butterfly sums (every lane gets the same bits: the rotation decision is wave-uniform).
constexpr int DW_N = 128;

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void design_vec_wide_kernel(const double *__restrict__ cov, int C, int bipolar, double rel_prec,
                                                              double *__restrict__ bf, int G, int g0)
{
    extern __shared__ double dyn[];
    const int n = C;
    double *Gc = dyn;          // [n][n]: column p at Gc + p n
    double *Dv = Gc + n * n;   // [n] singular values
    double *th = Dv + n;       // [n]
    int *order = reinterpret_cast<int *>(th + n);  // [n]
    __shared__ int rotated;
    __shared__ double psum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *Cm = cov + (size_t)blockIdx.x * C * C;
    const int g = g0 + blockIdx.x;

    if (!bipolar) {
        for (int e = tid; e < n * n; e += 256) {
            const int p = e / n, i = e - p * n;
            Gc[e] = 0.5 * (Cm[(size_t)i * n + p] + Cm[(size_t)p * n + i]);
        }
    } else {
        const int d = C / 2;
        for (int e = tid; e < d * d; e += 256) {
            const int i = e / d, j = e - i * d;
            const double P = (Cm[(size_t)i * C + j] + Cm[(size_t)(d + i) * C + d + j]) / 2;   // :195-199
            const double Q = (Cm[(size_t)i * C + d + j] + Cm[(size_t)(d + j) * C + i]) / 2;
            Gc[(size_t)j * n + i] = P;            // E[i][j]
            Gc[(size_t)(d + j) * n + i] = -Q;     // E[i][d + j]
            Gc[(size_t)j * n + d + i] = Q;        // E[d + i][j]
            Gc[(size_t)(d + j) * n + d + i] = P;  // E[d + i][d + j]
        }
    }
    if (tid == 0) rotated = 0;
    __syncthreads();

    const int half = n / 2;
    for (int sweep = 0; sweep < 40; ++sweep) {
        for (int step = 0; step < n - 1; ++step) {
            for (int k = wv; k < half; k += 4) {
                // round-robin tournament: player n - 1 stays, the others rotate
                const int p = k == 0 ? n - 1 : (step + k) % (n - 1);
                const int q = k == 0 ? step : (step - k + (n - 1)) % (n - 1);
                double *gp = Gc + (size_t)p * n, *gq = Gc + (size_t)q * n;
                const int i0 = lane, i1 = lane + 64;
                const double a0 = i0 < n ? gp[i0] : 0.0, a1 = i1 < n ? gp[i1] : 0.0;
                const double b0 = i0 < n ? gq[i0] : 0.0, b1 = i1 < n ? gq[i1] : 0.0;
                const double alpha = wave_sum(__builtin_fma(a0, a0, a1 * a1));
                const double beta = wave_sum(__builtin_fma(b0, b0, b1 * b1));
                const double gamma = wave_sum(__builtin_fma(a0, b0, a1 * b1));
                if (fabs(gamma) > 1e-15 * sqrt(alpha * beta) && gamma != 0.0) {  // (wave-uniform)
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                    if (i0 < n) {
                        gp[i0] = c * a0 - sn * b0;
                        gq[i0] = sn * a0 + c * b0;
                    }
                    if (i1 < n) {
                        gp[i1] = c * a1 - sn * b1;
                        gq[i1] = sn * a1 + c * b1;
                    }
                    if (lane == 0) rotated = 1;
                }
            }
            __syncthreads();
        }
        const int any = rotated;
        __syncthreads();
        if (tid == 0) rotated = 0;
        __syncthreads();
        if (!any) break;
    }

    // singular values = column norms; normalise the columns
    for (int p = wv; p < n; p += 4) {
        double *gp = Gc + (size_t)p * n;
        const int i0 = lane, i1 = lane + 64;
        const double a0 = i0 < n ? gp[i0] : 0.0, a1 = i1 < n ? gp[i1] : 0.0;
        const double nr = sqrt(wave_sum(__builtin_fma(a0, a0, a1 * a1)));
        const double inv = nr > 0.0 ? 1.0 / nr : 0.0;
        if (i0 < n) gp[i0] = a0 * inv;
        if (i1 < n) gp[i1] = a1 * inv;
        if (lane == 0) Dv[p] = nr;
    }
    __syncthreads();
    // descending order (LAPACK's singular-value order); ties: lower index first
    if (tid < n) {
        int r = 0;
        for (int j = 0; j < n; ++j) r += (Dv[j] > Dv[tid]) || (Dv[j] == Dv[tid] && j < tid);
        order[r] = tid;
    }
    __syncthreads();

    if (!bipolar) {
        // theta = U^T 1 (:393): column sums, in descending order of the singular values
        for (int r = wv; r < n; r += 4) {
            const double *gp = Gc + (size_t)order[r] * n;
            const double sm = wave_sum((lane < n ? gp[lane] : 0.0) + (lane + 64 < n ? gp[lane + 64] : 0.0));
            if (lane == 0) th[r] = sm;
        }
        __syncthreads();
        double u_min = Dv[order[1]], u_max = Dv[order[0]];
        const double th2 = tid < n ? th[tid] * th[tid] : 0.0, dmine = tid < n ? Dv[order[tid]] : 1.0;
        for (int it = 0; it < 200; ++it) {  // :399-411 (one term per thread; every thread sees the same sum)
            if ((u_max - u_min) / u_min < rel_prec) break;
            const double u_mid = (u_min + u_max) / 2;
            const double part = wave_sum(th2 / (dmine - u_mid));
            if (lane == 0) psum[wv] = part;
            __syncthreads();
            const double val = (psum[0] + psum[1]) + (psum[2] + psum[3]);
            __syncthreads();
            if (val < 0.0)
                u_min = u_mid;
            else
                u_max = u_mid;
        }
        const double root = (u_min + u_max) / 2.0;
        double w = 0.0;
        if (tid < n)
            for (int i = 0; i < n; ++i) w += Gc[(size_t)order[i] * n + tid] * (th[i] / (Dv[order[i]] - root));  // :417
        __syncthreads();
        if (tid < n) Dv[tid] = w;  // (the singular values are no longer needed)
        __syncthreads();
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += Dv[i] * Dv[i];
        nrm = sqrt(nrm);
        if (tid < n) bf[(size_t)tid * G + g] = w / nrm;
    } else {
        const int d = C / 2;
        const double *u = Gc + (size_t)order[0] * n;  // [x; y]
        // phase convention of design_vec_kernel: the first component real and negative (the largest one if the first vanishes)
        int kmax = 0;
        double best = -1.0;
        for (int k = 0; k < d; ++k) {
            const double m2 = u[k] * u[k] + u[d + k] * u[d + k];
            if (m2 > best) {
                best = m2;
                kmax = k;
            }
        }
        const double m0 = u[0] * u[0] + u[d] * u[d];
        const int kref = m0 >= 1e-6 * best ? 0 : kmax;
        const double mag = sqrt(kref == 0 ? m0 : best);
        const double cr = -u[kref] / mag, ci = u[d + kref] / mag;
        double nrm = 0.0;
        for (int k = 0; k < n; ++k) nrm += u[k] * u[k];
        nrm = sqrt(nrm);
        if (tid < d) {
            const double x = u[tid] / nrm, y = u[d + tid] / nrm;
            bf[(size_t)tid * G + g] = x * cr - y * ci;
            bf[(size_t)(d + tid) * G + g] = x * ci + y * cr;
        }
    }
}

hipError_t launch_design_vec(const double *cov, int n_doa, int C, int bipolar, double rel_prec, double *bf, int G, int g0,
                             hipStream_t stream)
{
    if (C > DW_N) return hipErrorInvalidValue;
    if (C <= DS_N) {
        hipLaunchKernelGGL(design_vec_kernel, dim3(n_doa), dim3(64), 0, stream, cov, C, bipolar, rel_prec, bf, G, g0);
        return hipGetLastError();
    }
    const size_t lds = ((size_t)C * C + 2 * (size_t)C) * sizeof(double) + (size_t)C * sizeof(int);
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(design_vec_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(design_vec_wide_kernel, dim3(n_doa), dim3(256), lds, stream, cov, C, bipolar, rel_prec, bf, G, g0);
    return hipGetLastError();
}

}  // namespace micloc
