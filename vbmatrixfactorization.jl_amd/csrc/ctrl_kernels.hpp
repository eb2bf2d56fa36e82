// ctrl_kernels.hpp -- the H x H "control" algebra of a sweep, fp64, one workgroup, LDS-resident.
//
//   ctrl_cov_kernel   SigmaA = sigma2*inv(B'B + L*SigmaB + sigma2*inv(CA))   src/vbmf.jl:96-97
//                     SigmaB = sigma2*inv(A'A + M*SigmaA + sigma2*inv(CB))   src/vbmf.jl:110-111
//                     (register-tiled Gauss-Jordan, one barrier per pivot; log-determinant from the pivots)
//   eig_kernel        lambda_max of the delta-Gram and of the B-Gram (repeated squaring + fp64 Rayleigh):
//                     Julia 0.5 `norm(::Matrix)` is the spectral norm (src/util.jl:27-29)
//   ctrl_end_kernel   updateCA!/updateCB! (src/vbmf.jl:129-146), updateSigma2! (src/vbmf.jl:153-157),
//                     d (src/vbmf.jl:211), loop test (src/vbmf.jl:193), build-defined ELBO, trace record.
//
// fp64 because ARD drives entries of C toward 0 and precisions to 1e10; on a
// 64..128-wide matrix the cost is negligible next to the streaming passes.
#pragma once
#include "common.hpp"

namespace vbmf {

// ---- layout of the device state block (doubles) ------------------------------------------------
struct StateLayout {
    int Hp;
    __host__ __device__ long long n2() const { return (long long)Hp * Hp; }
    __host__ __device__ long long GA() const { return 0; }
    __host__ __device__ long long GB() const { return n2(); }
    __host__ __device__ long long GD() const { return 2 * n2(); }     // must follow GB (one all-reduce)
    __host__ __device__ long long SA() const { return 3 * n2(); }
    __host__ __device__ long long SB() const { return 4 * n2(); }
    __host__ __device__ long long KB() const { return 5 * n2(); }     // matrix inverted for SigmaB
    __host__ __device__ long long W0() const { return 6 * n2(); }     // scratch (H > 128)
    __host__ __device__ long long W1() const { return 7 * n2(); }
    __host__ __device__ long long ca() const { return 8 * n2(); }
    __host__ __device__ long long cb() const { return 8 * n2() + Hp; }
    __host__ __device__ long long scal() const { return 8 * n2() + 2 * Hp; }
    __host__ __device__ long long total() const { return scal() + 32; }
};
enum : int { S_SIGMA2 = 0, S_TRYY, S_LOGDET_SA, S_LOGDET_SB, S_LAMB_PREV, S_LAMB_NEW, S_LAMD, S_D, S_ELBO,
             S_TRDOT, S_RESID, S_TRYBA };
enum : int { I_STOP = 0, I_ITERS = 1, I_ERR = 2, I_NITER = 3 };

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double block_max(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int i = 1; i < nw; ++i) s = fmax(s, red[i]);
    return s;
}

// In-place inverse of the SPD n x n matrix W (leading dim n) by Gauss-Jordan without pivoting.
// aux: 2n doubles.  Returns log det(W) (all threads).  Sets *err on a non-positive / non-finite pivot.
__device__ inline double gj_inverse_spd(double* W, int n, double* aux, int* err) {
    double* colk = aux;
    double* rowk = aux + n;
    double logdet = 0.0;
    for (int k = 0; k < n; ++k) {
        const double piv = W[(long long)k * n + k];
        if (!(piv > 0.0) || !isfinite(piv)) { if (threadIdx.x == 0) atomicExch(err, 1); }
        const double pinv = 1.0 / piv;
        logdet += log(piv);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            colk[i] = W[(long long)i * n + k];
            rowk[i] = (i == k ? 1.0 : W[(long long)k * n + i]) * pinv;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < n * n; t += blockDim.x) {
            const int i = t / n, j = t - i * n;
            double v;
            if (i == k) v = rowk[j];
            else v = (j == k ? 0.0 : W[t]) - colk[i] * rowk[j];
            W[t] = v;
        }
        __syncthreads();
    }
    return logdet;
}

// ---- register-tiled Gauss-Jordan ---------------------------------------------------------------
// T x T threads; thread (ty,tx) keeps the R x R entries {(ty + T*a, tx + T*b)} of the matrix in
// registers (cyclic distribution: every thread stays busy through all pivots).  Per pivot the owners
// of pivot row/column publish them through a double-buffered LDS strip, ONE barrier, then every
// thread updates its registers.  The matrix is padded with identity up to T*R.
template <int R, int T>
__device__ __forceinline__ void gj_tiled(double (&w)[R][R], int n, double* strip /* 2*2*T*R */, double* pivs) {
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    constexpr int NP = T * R;
#pragma unroll
    for (int a0 = 0; a0 < R; ++a0) {
        for (int t = 0; t < T; ++t) {
            const int k = a0 * T + t;
            if (k >= n) break;                                  // uniform
            double* row = strip + (k & 1) * (2 * NP);
            double* col = row + NP;
            if (ty == t) {
#pragma unroll
                for (int b = 0; b < R; ++b) row[tx + T * b] = w[a0][b];
            }
            if (tx == t) {
#pragma unroll
                for (int a = 0; a < R; ++a) col[ty + T * a] = w[a][a0];
            }
            __syncthreads();
            const double piv = row[k];
            const double pinv = 1.0 / piv;
            if (threadIdx.x == 0) pivs[k] = piv;
            double ci[R], rj[R];
#pragma unroll
            for (int a = 0; a < R; ++a) ci[a] = col[ty + T * a];
#pragma unroll
            for (int b = 0; b < R; ++b) rj[b] = row[tx + T * b] * pinv;
#pragma unroll
            for (int a = 0; a < R; ++a) {
                const bool is_i = (a == a0) && (ty == t);
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    const bool is_j = (b == a0) && (tx == t);
                    const double upd = w[a][b] - ci[a] * rj[b];
                    const double on_row = is_j ? pinv : rj[b];
                    const double on_col = -ci[a] * pinv;
                    w[a][b] = is_i ? on_row : (is_j ? on_col : upd);
                }
            }
        }
    }
}

// relaxed agent-scope read of the loop's stop flag (bypasses this CU's L1: the flag may have been
// raised a moment ago by ctrl_end in this very workgroup or by another one)
__device__ __forceinline__ int load_stop(const int* ints) {
    return __hip_atomic_load(ints + I_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// which = 0: SigmaA from (GB, SigmaB, ca), N = L_global.  which = 1: SigmaB from (GA, SigmaA, cb), N = M.
// Needs T*T threads and (4*T*R + T*R) doubles of LDS at `lds`; every thread of the block must call it.
template <int R, int T>
__device__ __forceinline__ void ctrl_cov_dev(double* __restrict__ st, StateLayout lay, int H, int which, double N,
                                             float* __restrict__ S32, int* __restrict__ ints, double* lds) {
    __shared__ double red[16];
    if (load_stop(ints)) return;
    constexpr int NP = T * R;
    const int Hp = lay.Hp;
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    const double* G = st + (which == 0 ? lay.GB() : lay.GA());
    const double* Sother = st + (which == 0 ? lay.SB() : lay.SA());
    double* Sself = st + (which == 0 ? lay.SA() : lay.SB());
    const double* cdiag = st + (which == 0 ? lay.ca() : lay.cb());
    double* scal = st + lay.scal();
    const double sigma2 = scal[S_SIGMA2];
    double w[R][R];
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const int i = ty + T * a, j = tx + T * b;
            double v = (i == j) ? 1.0 : 0.0;                    // identity padding
            if (i < H && j < H) {
                v = G[(long long)i * Hp + j] + N * Sother[(long long)i * Hp + j];
                if (i == j) v += sigma2 / cdiag[i];
                if (which == 1) st[lay.KB() + (long long)i * Hp + j] = v;
            }
            w[a][b] = v;
        }
    double* strip = lds;
    double* pivs = lds + 4 * NP;
    gj_tiled<R, T>(w, H, strip, pivs);
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const int i = ty + T * a, j = tx + T * b;
            if (i < Hp && j < Hp) {
                const double v = (i < H && j < H) ? w[a][b] : 0.0;
                Sself[(long long)i * Hp + j] = sigma2 * v;
                S32[(long long)i * Hp + j] = (float)v;          // Sigma/sigma2: what the post kernel multiplies by
            }
        }
    __syncthreads();
    double ld = 0.0;
    int bad = 0;
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        const double pv = pivs[k];
        if (!(pv > 0.0) || !isfinite(pv)) bad = 1;
        ld += log(pv);
    }
    ld = block_sum(ld, red);
    if (bad) atomicExch(ints + I_ERR, 1);
    if (threadIdx.x == 0) scal[which == 0 ? S_LOGDET_SA : S_LOGDET_SB] = (double)H * log(sigma2) - ld;
}

template <int R, int T>
__global__ __launch_bounds__(T * T) void ctrl_cov_kernel(double* __restrict__ st, StateLayout lay, int H, int which,
                                                         double N, float* __restrict__ S32, int* __restrict__ ints) {
    extern __shared__ __attribute__((aligned(16))) double lds_cov[];
    ctrl_cov_dev<R, T>(st, lay, H, which, N, S32, ints, lds_cov);
}

// lambda_max of a symmetric PSD H x H matrix: block 0 -> GD (S_LAMD), block 1 -> GB (S_LAMB_NEW).
// spectral = 0: Frobenius surrogate (trace) instead.
//
// Method: NSQ repeated squarings T <- T*T / ||T*T||_F in fp32 (register-tiled, LDS-resident) drive T
// to the dominant eigenprojector; its largest-diagonal column is then polished by two fp64 power
// steps on the ORIGINAL matrix and lambda = Rayleigh quotient in fp64.  The quotient's error is
// quadratic in the vector's error; the worst case over all spectra is <= n/(2e*2^(NSQ+1)) relative
// (~1e-4 at NSQ = 14, n = 64) and ~1e-12 whenever the top gap exceeds 0.1 %.  A cyclic Jacobi solve
// of the same matrix measured 2.9 ms on one CU; this is a few tens of microseconds.
constexpr int EIG_NSQ = 14;

// which = 0: GD -> S_LAMD, 1: GB -> S_LAMB_NEW.  256 threads, 2*NP*(NP+4) floats of LDS (NP = 16R).
template <int R>
__device__ __forceinline__ void eig_dev(double* __restrict__ st, StateLayout lay, int H, int spectral, int which,
                                        const int* __restrict__ ints, float* ldsf, int slot) {
    __shared__ double red[16];
    __shared__ int s_arg;
    if (load_stop(ints)) return;
    const int Hp = lay.Hp;
    const double* G = st + (which == 0 ? lay.GD() : lay.GB());
    double* scal = st + lay.scal();
    double tr = 0.0;
    for (int i = threadIdx.x; i < H; i += blockDim.x) tr += G[(long long)i * Hp + i];
    tr = block_sum(tr, red);
    if (!spectral || !(tr > 0.0) || !isfinite(tr)) {        // zero matrix -> 0; NaN propagates (loop exit on NaN d)
        if (threadIdx.x == 0) scal[slot] = tr;
        return;
    }
    // rows padded to LD = NP + 4 floats: 16-byte aligned row starts for ds_read_b128, and the 16 rows a
    // wave reads side by side start 4 banks apart (conflict-free)
    constexpr int T = 16, NP = T * R, LD = NP + 4;
    float* A0 = ldsf;
    float* A1 = ldsf + NP * LD;
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    for (int t = threadIdx.x; t < NP * NP; t += blockDim.x) {
        const int i = t / NP, j = t - i * NP;
        A0[i * LD + j] = (i < H && j < H) ? (float)(G[(long long)i * Hp + j] / tr) : 0.f;   // |G_ij| <= tr: no overflow
    }
    __syncthreads();
    float* cur = A0;
    float* nxt = A1;
    for (int sq = 0; sq < EIG_NSQ; ++sq) {
        float c[R][R];
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b) c[a][b] = 0.f;
        for (int k = 0; k < NP; k += 4) {                    // C = T * T' (T symmetric): both operands are row reads
            float4 ai[R], bj[R];
#pragma unroll
            for (int a = 0; a < R; ++a) ai[a] = *reinterpret_cast<const float4*>(cur + (ty + T * a) * LD + k);
#pragma unroll
            for (int b = 0; b < R; ++b) bj[b] = *reinterpret_cast<const float4*>(cur + (tx + T * b) * LD + k);
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    c[a][b] = fmaf(ai[a].x, bj[b].x, c[a][b]);
                    c[a][b] = fmaf(ai[a].y, bj[b].y, c[a][b]);
                    c[a][b] = fmaf(ai[a].z, bj[b].z, c[a][b]);
                    c[a][b] = fmaf(ai[a].w, bj[b].w, c[a][b]);
                }
        }
        double ss = 0.0;
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b) ss += (double)c[a][b] * (double)c[a][b];
        ss = block_sum(ss, red);                             // ||T^2||_F^2 with ||T||_F = 1: -> 1 iff T is a rank-1 projector
        const float sc = (ss > 0.0) ? (float)(1.0 / sqrt(ss)) : 0.f;
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b) nxt[(ty + T * a) * LD + tx + T * b] = c[a][b] * sc;
        __syncthreads();
        float* tmp = cur; cur = nxt; nxt = tmp;
        if (sq > 0 && fabs(ss - 1.0) < 1e-7) break;          // uniform: converged to the dominant projector
    }
    // dominant column = the one with the largest diagonal entry of the (near) projector
    if (threadIdx.x < 64) {
        float best = -1.f; int arg = 0;
        for (int i = threadIdx.x; i < H; i += 64) { const float v = cur[i * LD + i]; if (v > best) { best = v; arg = i; } }
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_down(best, off); const int oa = __shfl_down(arg, off);
            if (ob > best) { best = ob; arg = oa; }
        }
        if (threadIdx.x == 0) s_arg = arg;
    }
    __syncthreads();
    // lambda = v'Gv / v'v in fp64 on the ORIGINAL matrix (error quadratic in v's error).  4 threads per
    // row, each a contiguous quarter of it.
    double num = 0.0, den = 0.0;
    {
        const int i = threadIdx.x >> 2, q = threadIdx.x & 3;
        for (int i0 = 0; i0 < H; i0 += 64) {
            const int row = i0 + i;
            if (row < H) {
                const int jn = (H + 3) / 4, j0 = q * jn, j1 = min(H, j0 + jn);
                double acc = 0.0;
                for (int j = j0; j < j1; ++j) acc += G[(long long)row * Hp + j] * (double)cur[j * LD + s_arg];
                const double vi = (double)cur[row * LD + s_arg];
                num += vi * acc;
                if (q == 0) den += vi * vi;
            }
        }
    }
    num = block_sum(num, red);
    den = block_sum(den, red);
    if (threadIdx.x == 0) scal[slot] = den > 0.0 ? num / den : 0.0;
    __syncthreads();                                         // LDS is reused by whatever the block runs next
}

template <int R>
__global__ __launch_bounds__(256) void eig_kernel(double* __restrict__ st, StateLayout lay, int H, int spectral,
                                                  int do_d, int do_b, const int* __restrict__ ints) {
    extern __shared__ __attribute__((aligned(16))) float lds_eig[];
    const int which = blockIdx.x;           // 0: GD, 1: GB
    if ((which == 0 && !do_d) || (which == 1 && !do_b)) return;
    eig_dev<R>(st, lay, H, spectral, which, ints, lds_eig, which == 0 ? S_LAMD : S_LAMB_NEW);
}

// Fallback for H > 128 (the squaring kernel's two LDS tiles do not fit): 1024 threads, fp64 power iteration
// on the matrix in global memory (L2-resident) with a Rayleigh quotient at the end.  Converges like
// (lambda_2/lambda_1)^(2k) in the quotient; EIG_POWER_ITERS fixed, so near-degenerate top eigenvalues are
// only resolved to ~1e-3 -- accepted for this slow path (d is a stopping heuristic), stated in DESIGN.md.
constexpr int EIG_POWER_ITERS = 96;
__global__ __launch_bounds__(1024) void eig_power_kernel(double* __restrict__ st, StateLayout lay, int H, int spectral,
                                                         int do_d, int do_b, const int* __restrict__ ints) {
    __shared__ double red[16];
    __shared__ double v0[256], v1[256];
    if (load_stop(ints)) return;
    const int which = blockIdx.x;           // 0: GD, 1: GB
    if ((which == 0 && !do_d) || (which == 1 && !do_b)) return;
    const int Hp = lay.Hp;
    const double* G = st + (which == 0 ? lay.GD() : lay.GB());
    double* scal = st + lay.scal();
    const int slot = which == 0 ? S_LAMD : S_LAMB_NEW;
    double tr = 0.0;
    for (int i = threadIdx.x; i < H; i += blockDim.x) tr += G[(long long)i * Hp + i];
    tr = block_sum(tr, red);
    if (!spectral || !(tr > 0.0) || !isfinite(tr)) {
        if (threadIdx.x == 0) scal[slot] = tr;
        return;
    }
    // start from the diagonal (a positive vector correlated with the dominant eigenvector of a PSD matrix)
    for (int i = threadIdx.x; i < H; i += blockDim.x) v0[i] = G[(long long)i * Hp + i] / tr + 1e-3;
    __syncthreads();
    const int row = threadIdx.x >> 2, q = threadIdx.x & 3;      // 4 threads per row, 256 rows
    double lam = 0.0;
    for (int it = 0; it < EIG_POWER_ITERS; ++it) {
        double acc = 0.0;
        if (row < H) {
            const int jn = (H + 3) / 4, j0 = q * jn, j1 = min(H, j0 + jn);
            for (int j = j0; j < j1; ++j) acc += G[(long long)row * Hp + j] * v0[j];
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        double num = 0.0, den = 0.0;
        if (row < H && q == 0) { v1[row] = acc; num = v0[row] * acc; den = v0[row] * v0[row]; }
        num = block_sum(num, red);
        den = block_sum(den, red);
        lam = den > 0.0 ? num / den : 0.0;
        double n1 = 0.0;
        if (row < H && q == 0) n1 = acc * acc;
        n1 = block_sum(n1, red);
        const double sc = n1 > 0.0 ? 1.0 / sqrt(n1) : 0.0;
        __syncthreads();
        for (int i = threadIdx.x; i < H; i += blockDim.x) v0[i] = v1[i] * sc;
        __syncthreads();
    }
    if (threadIdx.x == 0) scal[slot] = lam;
}

// flags: bit0 est_covs->CA, bit1 est_covs->CB, bit2 est_var, bit3 compute d + loop bookkeeping,
//        bit4 tr(Y'BA') from the Gram identity tr(KB o GB) (else from scal[S_TRDOT]),
//        bit5 S_LAMB_PREV already holds lambda_max of the old B'B (no rotation from S_LAMB_NEW)
__device__ __forceinline__ void ctrl_end_dev(double* __restrict__ st, StateLayout lay, int H, double Lg, double M,
                                             int flags, double eps, double* __restrict__ trace,
                                             int* __restrict__ ints) {
    __shared__ double red[16];
    if (load_stop(ints)) return;
    const int Hp = lay.Hp;
    const double* GA = st + lay.GA();
    const double* GB = st + lay.GB();
    const double* SA = st + lay.SA();
    const double* SB = st + lay.SB();
    const double* KB = st + lay.KB();
    double* ca = st + lay.ca();
    double* cb = st + lay.cb();
    double* scal = st + lay.scal();

    // tr(Y'BA') and tr((A'A + M SigmaA)(B'B + L SigmaB))
    double t1 = 0.0, t2 = 0.0;
    for (int t = threadIdx.x; t < H * H; t += blockDim.x) {
        const int i = t / H, j = t - i * H;
        const long long ij = (long long)i * Hp + j;
        t1 += KB[ij] * GB[ij];
        t2 += (GA[ij] + M * SA[ij]) * (GB[ij] + Lg * SB[ij]);
    }
    t1 = block_sum(t1, red);
    t2 = block_sum(t2, red);
    const double trYBA = (flags & 16) ? t1 : scal[S_TRDOT];
    const double resid = scal[S_TRYY] - 2.0 * trYBA + t2;
    __syncthreads();
    if (flags & 1) for (int h = threadIdx.x; h < H; h += blockDim.x) ca[h] = GA[(long long)h * Hp + h] / M + SA[(long long)h * Hp + h];
    if (flags & 2) for (int h = threadIdx.x; h < H; h += blockDim.x) cb[h] = GB[(long long)h * Hp + h] / Lg + SB[(long long)h * Hp + h];
    __syncthreads();
    double sigma2 = scal[S_SIGMA2];
    if (flags & 4) sigma2 = resid / (Lg * M);

    // build-defined ELBO, with the post-update CA, CB, sigma2
    double e = 0.0;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        const long long hh = (long long)h * Hp + h;
        e += -(M / 2.0) * log(ca[h]) - 0.5 * (GA[hh] + M * SA[hh]) / ca[h];
        e += -(Lg / 2.0) * log(cb[h]) - 0.5 * (GB[hh] + Lg * SB[hh]) / cb[h];
    }
    e = block_sum(e, red);
    if (threadIdx.x == 0) {
        const double PI2 = 6.283185307179586476925286766559;
        double F = -(Lg * M / 2.0) * log(PI2 * sigma2) - resid / (2.0 * sigma2) + e;
        F += (M / 2.0) * scal[S_LOGDET_SA] + M * H / 2.0 + (Lg / 2.0) * scal[S_LOGDET_SB] + Lg * H / 2.0;
        scal[S_SIGMA2] = sigma2;
        scal[S_ELBO] = F;
        scal[S_RESID] = resid;
        scal[S_TRYBA] = trYBA;
        if (flags & 8) {
            const double d = sqrt(scal[S_LAMD] / scal[S_LAMB_PREV]);
            scal[S_D] = d;
            if (!(flags & 32)) scal[S_LAMB_PREV] = scal[S_LAMB_NEW];   // (32: lambda_max of the old B is written there directly)
            const int it = ints[I_ITERS];
            if (trace) { trace[4 * it + 0] = d; trace[4 * it + 1] = sigma2; trace[4 * it + 2] = F; trace[4 * it + 3] = resid; }
            ints[I_ITERS] = it + 1;
            if (!(d > eps) || it + 1 >= ints[I_NITER])                     // src/vbmf.jl:193 (NaN d exits too)
                __hip_atomic_store(ints + I_STOP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __threadfence();
    __syncthreads();
}

__global__ __launch_bounds__(256) void ctrl_end_kernel(double* __restrict__ st, StateLayout lay, int H, double Lg,
                                                       double M, int flags, double eps,
                                                       double* __restrict__ trace, int* __restrict__ ints) {
    ctrl_end_dev(st, lay, H, Lg, M, flags, eps, trace, ints);
}

// ---- the control chain as run by workgroup 0 of a streaming-pass launch ----------------------------
// The H x H algebra of a sweep depends only on Grams that are complete before the pass is launched
// and produces what the kernels AFTER the pass need, so one extra workgroup of the pass runs it while
// the others stream: one stream, no events, kernel boundaries give all ordering and visibility.
struct CtrlArgs {
    double* st; StateLayout lay; int* ints; double* trace;
    float* S32;            // SigmaA/sigma2 (pass 1) or SigmaB/sigma2 (pass 2) table for the post kernel
    double Lg, M, eps;
    int H, spectral, end_flags;
    int mode;              // CTRL_* bits; 0: no control workgroup in this launch
};
// Schedule inside vbmf_run (sweep j; B_{j-1} is the factor the sweep starts from):
//   pass 1 of sweep j : [lambda_max(dB'dB) of sweep j-1, ctrl_end of sweep j-1]  then  SigmaA of sweep j
//   pass 2 of sweep j : SigmaB of sweep j  and  lambda_max(B_{j-1}'B_{j-1})  (the denominator of d_j; the
//                       Gram of B_{j-1} is still in place: post(B) of sweep j runs after this launch)
// so each pass carries a chain of similar length (~75 / ~65 us at H = 64).
enum : int { CTRL_PREV_END = 1, CTRL_COV_A = 2, CTRL_COV_B = 4, CTRL_EIG_BOLD = 8 };

template <int R>
__device__ __forceinline__ void ctrl_chain(const CtrlArgs& a, void* lds) {
    if (a.mode & CTRL_PREV_END) {
        eig_dev<R>(a.st, a.lay, a.H, a.spectral, 0, a.ints, reinterpret_cast<float*>(lds), S_LAMD);
        ctrl_end_dev(a.st, a.lay, a.H, a.Lg, a.M, a.end_flags | 32, a.eps, a.trace, a.ints);
    }
    if (a.mode & CTRL_COV_A) ctrl_cov_dev<R, 16>(a.st, a.lay, a.H, 0, a.Lg, a.S32, a.ints, reinterpret_cast<double*>(lds));
    if (a.mode & CTRL_COV_B) ctrl_cov_dev<R, 16>(a.st, a.lay, a.H, 1, a.M, a.S32, a.ints, reinterpret_cast<double*>(lds));
    if (a.mode & CTRL_EIG_BOLD) {
        __syncthreads();
        eig_dev<R>(a.st, a.lay, a.H, a.spectral, 1, a.ints, reinterpret_cast<float*>(lds), S_LAMB_PREV);
    }
}

__global__ void copy_scalar_kernel(double* st, StateLayout lay, int dst, int src) {
    if (threadIdx.x == 0 && blockIdx.x == 0) st[lay.scal() + dst] = st[lay.scal() + src];
}

}  // namespace vbmf
