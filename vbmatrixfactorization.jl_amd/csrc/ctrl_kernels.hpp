// ctrl_kernels.hpp -- the H x H "control" algebra of a sweep, fp64, one workgroup, LDS-resident.
//
//   ctrl_cov_kernel   SigmaA = sigma2*inv(B'B + L*SigmaB + sigma2*inv(CA))   src/vbmf.jl:96-97
//                     SigmaB = sigma2*inv(A'A + M*SigmaA + sigma2*inv(CB))   src/vbmf.jl:110-111
//                     (register-tiled Gauss-Jordan, one barrier per pivot; log-determinant from the pivots)
//   eig_kernel        lambda_max of the delta-Gram and of the B-Gram (Lanczos on the register-resident matrix):
//                     Julia 0.5 `norm(::Matrix)` is the spectral norm (src/util.jl:27-29)
//   ctrl_end_kernel   updateCA!/updateCB! (src/vbmf.jl:129-146), updateSigma2! (src/vbmf.jl:153-157),
//                     d (src/vbmf.jl:211), loop test (src/vbmf.jl:193), build-defined ELBO, trace record.
//
// fp64 because ARD drives entries of C toward 0 and precisions to 1e10; on a
// 64..128-wide matrix the cost is negligible next to the streaming passes.
#pragma once
#include "common.hpp"
#include "blk_inverse.hpp"

namespace vbmf {

// ---- layout of the device state block (doubles) ------------------------------------------------
struct StateLayout {
    int Hp;
    __host__ __device__ long long n2() const { return (long long)Hp * Hp; }
    __host__ __device__ long long GA() const { return 0; }
    __host__ __device__ long long GB() const { return n2(); }
    __host__ __device__ long long GD() const { return 2 * n2(); }     // must follow GB (one all-reduce)
    // GX: 8 doubles right behind GD, so that [GB | GD | GX] is ONE contiguous all-reduce target; GX[0] = tr(B'YA) = the
    // direct sum  sum_{l,h} (Y A)[l,h] * BHat[l,h]  over this rank's rows, formed where BHat is produced
    __host__ __device__ long long GX() const { return 3 * n2(); }
    __host__ __device__ long long SA() const { return 3 * n2() + 8; }
    __host__ __device__ long long SB() const { return 4 * n2() + 8; }
    __host__ __device__ long long KB() const { return 5 * n2() + 8; }     // matrix inverted for SigmaB
    __host__ __device__ long long W0() const { return 6 * n2() + 8; }     // scratch (H > 128)
    __host__ __device__ long long W1() const { return 7 * n2() + 8; }
    __host__ __device__ long long W2() const { return 8 * n2() + 8; }
    __host__ __device__ long long ca() const { return 9 * n2() + 8; }
    __host__ __device__ long long cb() const { return 9 * n2() + 8 + Hp; }
    __host__ __device__ long long scal() const { return 9 * n2() + 8 + 2 * Hp; }
    __host__ __device__ long long total() const { return scal() + 32; }
};
enum : int { S_SIGMA2 = 0, S_TRYY, S_LOGDET_SA, S_LOGDET_SB, S_LAMB_PREV, S_LAMB_NEW, S_LAMD, S_D, S_ELBO,
             S_TRDOT, S_RESID, S_TRYBA,
             S_LOGDET_SA_SHADOW = 20 };      // 12..18: sparse_kernels.hpp
enum : int { I_STOP = 0, I_ITERS = 1, I_ERR = 2, I_NITER = 3, I_SREADY = 4, I_DBG_SB_PPM = 5 };   // (5: test hook, vbmf_debug_set)

// Threads of the workgroup that run the control algebra: the whole block -- except in the 512-thread launch of the H >= 128
// streaming kernel (stream_gemm.hpp, stream_lds8_kernel), whose control workgroups run on their first 256 threads (waves 4-7
// leave at once; a barrier counts live waves only).  No control kernel of its own is ever launched with 512 threads.
__device__ __forceinline__ int ctrl_nthreads() { return blockDim.x == 512u ? 256 : (int)blockDim.x; }

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int nw = (ctrl_nthreads() + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double block_max(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
    const int nw = (ctrl_nthreads() + 63) >> 6;
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
        for (int i = threadIdx.x; i < n; i += ctrl_nthreads()) {
            colk[i] = W[(long long)i * n + k];
            rowk[i] = (i == k ? 1.0 : W[(long long)k * n + i]) * pinv;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < n * n; t += ctrl_nthreads()) {
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
// 1/x to fp64 accuracy without the IEEE division sequence (v_rcp_f64 + two Newton steps): the pivots of an SPD sweep are
// positive, finite and far from the denormal range
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// In-place Gauss-Jordan inverse of the SPD matrix held as R x R register tiles (thread (ty, tx) owns rows ty + T*a,
// columns tx + T*b); one barrier per pivot, pivot row / column handed over through double-buffered LDS strips.
// The reciprocal of pivot k+1 is computed during step k's update (every thread on its own diagonal-tile element, so
// there is no branch and the division chain overlaps the update's FMAs) and travels in the column strip's slot k+1 --
// col[k] would otherwise duplicate row[k].  (It used to be divided out after the barrier: ~25 % of a step.)
template <int R, int T>
__device__ __forceinline__ void gj_tiled(double (&w)[R][R], int n, double* strip /* 2*2*T*R */, double* pivs) {
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    constexpr int NP = T * R;
    double pinv_next = fast_rcp(w[0][0]);                       // the owner of (0, 0) holds the first pivot here
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
                for (int a = 0; a < R; ++a) col[ty + T * a] = (a == a0 && ty == t) ? pinv_next : w[a][a0];
            }
            __syncthreads();
            const double piv = row[k];
            const double pinv = col[k];
            if (threadIdx.x == 0) pivs[k] = piv;
            double ci[R], rj[R];
#pragma unroll
            for (int a = 0; a < R; ++a) ci[a] = col[ty + T * a];       // ci of row k itself is the reciprocal: unused there
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
            // reciprocal of the next pivot, element (k+1, k+1): tile (a0, a0) for t + 1 < T, else tile (a0+1, a0+1)
            double cand = w[a0][a0];
            if (a0 + 1 < R) cand = (t + 1 < T) ? cand : w[a0 + 1][a0 + 1];
            pinv_next = fast_rcp(cand);
        }
    }
}

// The same sweep on NB independent matrices at once (same n): one barrier per pivot serves all of them and their
// dependent LDS round trips overlap.
// strip: NB * 4*T*R doubles, pivs: NB * T*R doubles (matrix q's pivots at pivs + q*T*R).
template <int R, int T, int NB>
__device__ __forceinline__ void gj_tiled_n(double (&w)[NB][R][R], int n, double* strip, double* pivs) {
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    constexpr int NP = T * R;
    double pinv_next[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) pinv_next[q] = fast_rcp(w[q][0][0]);
#pragma unroll
    for (int a0 = 0; a0 < R; ++a0) {
        for (int t = 0; t < T; ++t) {
            const int k = a0 * T + t;
            if (k >= n) break;                                  // uniform
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                double* row = strip + q * (4 * NP) + (k & 1) * (2 * NP);
                double* col = row + NP;
                if (ty == t) {
#pragma unroll
                    for (int b = 0; b < R; ++b) row[tx + T * b] = w[q][a0][b];
                }
                if (tx == t) {
#pragma unroll
                    for (int a = 0; a < R; ++a) col[ty + T * a] = (a == a0 && ty == t) ? pinv_next[q] : w[q][a][a0];
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const double* row = strip + q * (4 * NP) + (k & 1) * (2 * NP);
                const double* col = row + NP;
                const double piv = row[k];
                const double pinv = col[k];
                if (threadIdx.x == 0) pivs[q * NP + k] = piv;
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
                        const double upd = w[q][a][b] - ci[a] * rj[b];
                        const double on_row = is_j ? pinv : rj[b];
                        const double on_col = -ci[a] * pinv;
                        w[q][a][b] = is_i ? on_row : (is_j ? on_col : upd);
                    }
                }
                double cand = w[q][a0][a0];
                if (a0 + 1 < R) cand = (t + 1 < T) ? cand : w[q][a0 + 1][a0 + 1];
                pinv_next[q] = fast_rcp(cand);
            }
        }
    }
}

// ---- 129 <= H <= 256: the blocked sweep with the matrix in registers (inv256_blk, blk_inverse.hpp) ----------------------------------
// A 256 x 256 fp64 matrix is 512 KB -- the whole register file of a CU -- so the LDS-image sweep of the smaller ranks cannot hold it.  Rounds
// 1-2 split K = [A B; B' D] and took the Schur complement (two register-tiled 128 x 128 Gauss-Jordans with a barrier per pivot + four VALU
// GEMMs through global scratch: 0.38 ms alone, 0.46-0.85 beside a pass); late in round 3 the 136 upper 16 x 16 blocks live in the registers
// of the 16 waves and the sweep's row panel travels through LDS: 0.14 ms alone, 0.22-0.28 beside a pass, 3e-15 from a long-double reference.

// relaxed agent-scope read of the loop's stop flag (bypasses this CU's L1: the flag may have been
// raised a moment ago by ctrl_end in this very workgroup or by another one)
__device__ __forceinline__ int load_stop(const int* ints) {
    return __hip_atomic_load(ints + I_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Row-sharded runs: the ranks' error flags travel as one more number of the packed Gram message (post_kernels.hpp,
// publish_err_flag) and arrive summed at st[GX + 1]; every rank's loop test reads the same sum, so a hand-off timeout or a bad
// pivot on ONE rank stops EVERY rank at the same sweep (without it the other ranks ran on into the all-reduce with that rank's
// garbage partials).  Returns 0, or the error code this rank records if it has none of its own: 2 (hand-off) / 1 (numeric).
__device__ __forceinline__ int remote_error_code(const double* __restrict__ st, StateLayout lay) {
    const double f = st[lay.GX() + 1];
    return f >= 1000.0 ? 2 : (f > 0.0 ? 1 : 0);
}
__device__ __forceinline__ bool stop_on_remote_error(const double* __restrict__ st, StateLayout lay, int* __restrict__ ints) {
    const int rc = remote_error_code(st, lay);
    if (rc == 0) return false;
    if (atomicCAS(ints + I_ERR, 0, rc) == 0) atomicOr(ints + I_ERR, 0x200);   // 0x200: no error of this rank's own -- another rank's
    return true;
}

// Inverse of an SPD matrix of order H <= 16 NB in LDS by the FOUR waves of a 256-thread control workgroup (blk_inverse.hpp:
// blocked symmetric sweep, the 16 x 16 diagonal blocks inside a wavefront, rank-16 updates on the fp64 MFMA, two barriers per
// BLOCK step where round 2's register-tiled Gauss-Jordan had a barrier and two LDS round trips per pivot).
//   elem(i, j): the matrix (i, j < H);  W: 16 NB x (16 NB + 2) doubles of LDS;  on return W's upper triangle holds -inverse
//   (read it through spd_inv_at), *logdet = log det of the matrix, *bad != 0 if a pivot was not positive / finite.
// Every thread of the control workgroup (ctrl_nthreads() = 256 of them) must call it.
template <int NB, class ElemF>
__device__ __forceinline__ void spd_inverse_lds4(double* W, int H, ElemF elem, double* logdet, int* bad) {
    constexpr int NP = 16 * NB, LD = NP + 2;
    for (int t = threadIdx.x; t < NP * NP; t += 256) {
        const int i = t / NP, j = t % NP;
        W[i * LD + j] = (i < H && j < H) ? elem(i, j) : (i == j ? 1.0 : 0.0);      // identity padding
    }
    __syncthreads();
    PivAcc pv;
    blk_sweep<NB, 4>(W, LD, (H + 15) >> 4, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, pv);
    *logdet = pv.logdet();
    *bad = pv.bad;
}
template <int NB>
__device__ __forceinline__ double spd_inv_at(const double* W, int i, int j) {          // element (i, j) of the inverse
    constexpr int LD = 16 * NB + 2;
    return -(i <= j ? W[i * LD + j] : W[j * LD + i]);
}
constexpr size_t spd_inverse_lds_bytes(int NB) { return (size_t)(16 * NB) * (16 * NB + 2) * sizeof(double); }

// which = 0: SigmaA from (GB, SigmaB, ca), N = L_global.  which = 1: SigmaB from (GA, SigmaA, cb), N = M.
// Needs T*T threads and (4*T*R + T*R) doubles of LDS at `lds`; every thread of the block must call it.
// shadow (which = 0, H <= 128 only): SigmaA and its log-determinant go to the W0 scratch / the shadow scalar instead
// of the state; commit_cov_a_kernel copies them over once it is known that the loop continues (ctrl_chain).
template <int R, int T>
__device__ __forceinline__ void ctrl_cov_dev(double* __restrict__ st, StateLayout lay, int H, int which, double N,
                                             float* __restrict__ S32, int* __restrict__ ints, double* lds,
                                             bool shadow = false) {
    __shared__ double red[16];
    if (load_stop(ints)) return;
    constexpr int NP = T * R;
    const int Hp = lay.Hp;
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    const double* G = st + (which == 0 ? lay.GB() : lay.GA());
    const double* Sother = st + (which == 0 ? lay.SB() : lay.SA());
    double* Sself = st + (which == 0 ? (shadow ? lay.W0() : lay.SA()) : lay.SB());
    const double* cdiag = st + (which == 0 ? lay.ca() : lay.cb());
    double* scal = st + lay.scal();
    const double sigma2 = scal[S_SIGMA2];
    double* pivs;
    if constexpr (R == 8 && T == 32) {
        // 129 <= H <= 256 (Hp = 256): the blocked sweep with the matrix in registers (inv256_blk, blk_inverse.hpp)
        double* Kg = st + lay.W0();
        double* Ki = st + lay.W1();
        for (int t = threadIdx.x; t < 256 * 256; t += 1024) {
            const int i = t >> 8, j = t & 255;
            double v = (i == j) ? 1.0 : 0.0;                    // identity padding
            if (i < H && j < H) {
                v = G[(long long)i * Hp + j] + N * Sother[(long long)i * Hp + j];
                if (i == j) v += sigma2 / cdiag[i];
            }
            Kg[t] = v;
        }
        __syncthreads();
        pivs = lds + INV256_LDS_DOUBLES;
        inv256_blk(Kg, Ki, lds, pivs, (H + 15) >> 4);
        for (int t = threadIdx.x; t < 256 * 256; t += 1024) {
            const int i = t >> 8, j = t & 255;
            const double v = (i < H && j < H) ? Ki[t] : 0.0;
            Sself[(long long)i * Hp + j] = sigma2 * v;
            S32[(long long)i * Hp + j] = (float)v;
        }
    } else {
        // H <= 128: blocked sweep in LDS (spd_inverse_lds4)
        static_assert(T == 16, "four waves");
        constexpr int NBc = NP / 16;
        double ldK; int badK;
        spd_inverse_lds4<NBc>(lds, H, [&](int i, int j) {
            double v = G[(long long)i * Hp + j] + N * Sother[(long long)i * Hp + j];
            if (i == j) v += sigma2 / cdiag[i];
            return v;
        }, &ldK, &badK);
        const double dbg = which == 1 ? 1.0 + 1e-6 * (double)ints[I_DBG_SB_PPM] : 1.0;      // test hook (normally exactly 1)
        for (int t = threadIdx.x; t < Hp * Hp; t += 256) {
            const int i = t / Hp, j = t % Hp;
            const double v = (i < H && j < H) ? spd_inv_at<NBc>(lds, i, j) : 0.0;
            Sself[(long long)i * Hp + j] = sigma2 * v;
            S32[(long long)i * Hp + j] = (float)(v * dbg);          // Sigma/sigma2: what the post kernel multiplies by
        }
        if (badK) atomicExch(ints + I_ERR, 1);
        if (threadIdx.x == 0)
            scal[which == 0 ? (shadow ? S_LOGDET_SA_SHADOW : S_LOGDET_SA) : S_LOGDET_SB] = (double)H * log(sigma2) - ldK;
        return;
    }
    __syncthreads();
    double ld = 0.0;
    int bad = 0;
    for (int k = threadIdx.x; k < H; k += ctrl_nthreads()) {
        const double pv = pivs[k];
        if (!(pv > 0.0) || !isfinite(pv)) bad = 1;
        ld += log(pv);
    }
    ld = block_sum(ld, red);
    if (bad) atomicExch(ints + I_ERR, 1);
    if (threadIdx.x == 0)
        scal[which == 0 ? (shadow ? S_LOGDET_SA_SHADOW : S_LOGDET_SA) : S_LOGDET_SB] = (double)H * log(sigma2) - ld;
}

// the loop continues: make the speculative SigmaA of ctrl_chain the state's
__global__ __launch_bounds__(256) void commit_cov_a_kernel(double* __restrict__ st, StateLayout lay, const int* __restrict__ ints) {
    if (load_stop(ints)) return;
    const long long n = lay.n2();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        st[lay.SA() + i] = st[lay.W0() + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) st[lay.scal() + S_LOGDET_SA] = st[lay.scal() + S_LOGDET_SA_SHADOW];
}

template <int R, int T>
__global__ __launch_bounds__(T * T) void ctrl_cov_kernel(double* __restrict__ st, StateLayout lay, int H, int which,
                                                         double N, float* __restrict__ S32, int* __restrict__ ints) {
    extern __shared__ __attribute__((aligned(16))) double lds_cov[];
    ctrl_cov_dev<R, T>(st, lay, H, which, N, S32, ints, lds_cov);
}

// ---- lambda_max of a symmetric PSD matrix by LANCZOS on the register-resident matrix (H > 64) -------------------------------------
// History.  H <= 128, rounds 1-3: ten repeated squarings of the LDS-resident fp32 matrix towards the dominant projector, then an fp64
// Rayleigh quotient -- exact (1e-12) when the top gap exceeds 0.5 %, but up to 1.8e-4 off inside a cluster (measured 2.4e-4 on a spectrum
// flat to 0.1 %: tests/test_gpu_lambda_max.py), 25 us at H = 64, 217 us at H = 128.  H > 128: power iteration from registers, 128 fixed
// steps (round 2: ~1e-3 in a cluster), then to a tolerance capped at 2048 steps (round 3) -- which a FLAT spectrum (the delta-Gram of a
// rank-256 fit to rank-16 data) runs into: 3.1 ms per call beside a 0.75 ms pass, and still only ~1e-4, because the power method's error
// in a cluster falls like 1/k.  The Krylov space of the same matrix-vector products does far better (error ~ cosh(2 k sqrt(gap))^-2, exact
// after H steps), for two block-wide dot products more per step:
//     w = G q_j;  alpha_j = q_j'w;  w -= alpha_j q_j + beta_j q_(j-1);  beta_(j+1) = ||w||;  q_(j+1) = w / beta_(j+1)
// and lambda_max(G) ~ the top eigenvalue of the tridiagonal T_k = tridiag(beta, alpha, beta), found by Sturm-count multisection
// (256 shifts per round, 4 rounds: 2e-10 of the trace) at k = 12, 24, 36, 48, 64, 96, 128, 192, 256; the iteration ends when that
// value grew by <= EIG_LANCZOS_TOL since the previous look, at an invariant subspace (beta = 0), or at k = H.  No re-orthogonalisation:
// the copies of converged Ritz values that its absence breeds never exceed lambda_max.  Measured (tests/test_gpu_lambda_max.py: flat,
// Marchenko-Pastur, near-degenerate pair, rank-deficient, dominant, 2^-k, identity): <= 8e-8 relative at H = 128, 130, 200, 256 (and at H = 40 when
// routed here; H <= 64 keeps the squaring kernel below for its speed inside short pass launches).
// Layout: NT threads (256 inside a pass launch and for 64 < H <= 128, 1024 for H > 128), matrix padded to NP = 16, 32, 64, 128 or 256.  Thread
// (group = t / NCQ, cq = t % NCQ), NCQ = NP / 16, holds the RB x 16 block G[RB group .. +RB][16 cq .. +16] / tr in registers as fp32
// pairs (the products go out as v_pk_fma_f32: a step is VALU-bound), reads its 16 entries of q from LDS (four 16-byte reads), forms RB
// partial dot products and folds them over the NCQ chunks of its group with DPP moves.  Lane cq < RB of a group OWNS row RB group + cq:
// it keeps q_j and q_(j-1) of that row in fp64.
typedef float f32x2v __attribute__((ext_vector_type(2)));
// sum over the N (1, 2, 4, 8, 16; aligned) neighbouring lanes of a DPP row, the result in each of them: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror (cross-lane moves inside the VALU; __shfl_xor would be ds_bpermute round trips through LDS)
template <int N>
__device__ __forceinline__ float rowN_sum(float x) {
    if constexpr (N >= 2) x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    if constexpr (N >= 4) x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    if constexpr (N >= 8) x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    if constexpr (N >= 16) x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
    return x;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double lane_f64(double x, int l) {          // wave-uniform value of lane l
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum over the first NT threads of the workgroup (whole waves), the wave stage on DPP + readlane (block_sum's six fp64 __shfl_down steps
// are twelve ds_bpermute round trips); fixed order, every thread gets the same value.  red: NT / 64 doubles.
template <int NT>
__device__ __forceinline__ double block_sum_dpp(double x, double* red) {
    x += dpp_f64<0xB1>(x);
    x += dpp_f64<0x4E>(x);
    x += dpp_f64<0x141>(x);
    x += dpp_f64<0x140>(x);
    const double wsum = (lane_f64(x, 0) + lane_f64(x, 16)) + (lane_f64(x, 32) + lane_f64(x, 48));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wsum;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}
// top eigenvalue of the k x k tridiagonal (diagonal al[0..k), squared off-diagonals be2[1..k)) inside [lo, hi] (hi an upper bound of the
// spectrum): `rounds` rounds of 256-way multisection on the Sturm count.  Threads 0..255 count, every thread returns the same value.
__device__ __forceinline__ double tri_lambda_max(const double* al, const double* be2, int k, double lo, double hi, int* imin, int rounds) {
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) *imin = 255;
        __syncthreads();
        const int t = threadIdx.x;
        const double step = (hi - lo) * (1.0 / 256.0);
        if (t < 256) {
            const double x = t == 255 ? hi : lo + step * (double)(t + 1);
            double d = al[0] - x;
            int cnt = d < 0.0 ? 1 : 0;
            for (int i = 1; i < k; ++i) {
                if (fabs(d) < 1e-290) d = -1e-290;
                d = al[i] - x - be2[i] * __builtin_amdgcn_rcp(d);
                cnt += d < 0.0 ? 1 : 0;
            }
            if (cnt >= k) atomicMin(imin, t);                   // every eigenvalue lies below x
        }
        __syncthreads();
        const int tm = *imin;
        const double nhi = tm == 255 ? hi : lo + step * (double)(tm + 1);
        const double nlo = tm == 0 ? lo : lo + step * (double)tm;
        __syncthreads();                                        // (imin is reset at the top of the next round)
        lo = nlo; hi = nhi;
    }
    return 0.5 * (lo + hi);
}
constexpr double EIG_LANCZOS_TOL = 2e-7;                        // growth of the top Ritz value between two looks, relative
// LDS of a call: NP floats (q) + 2 NP + 1 doubles (alpha, beta^2) + 16 doubles + 1 int; the caller hands over >= EIG_LDS_BYTES
constexpr int EIG_LDS_BYTES = 256 * 4 + (2 * 256 + 2) * 8 + 16 * 8 + 16;
// lambda_max(G) / tr for the H x H matrix G (row stride Hp, H <= NP, tr = its trace > 0).  Every one of the NT threads gets the value.
template <int NP, int NT>
__device__ __forceinline__ double lanczos_lambda_max(const double* __restrict__ G, int Hp, int H, double tr, void* lds) {
    constexpr int NCQ = NP / 16;
    constexpr int RB = (NP * NCQ + NT - 1) / NT;                // rows per thread: 4 at NP = 128 / NT = 256 and NP = 256 / NT = 1024, else 1
    static_assert(RB == 1 || RB == 4, "register block");
    static_assert(NCQ >= RB, "one owner lane per row of the block");
    float* vbuf = reinterpret_cast<float*>(lds);
    double* al = reinterpret_cast<double*>(vbuf + 256);
    double* be2 = al + 256;
    double* red = be2 + 258;
    int* imin = reinterpret_cast<int*>(red + 16);
    const int t = threadIdx.x;
    const int grp = t / NCQ, cq = t % NCQ;
    const int row0 = grp * RB;
    const bool active = row0 < NP;
    const bool owner = active && cq < RB;
    const int myrow = row0 + (cq % RB);
    f32x2v g[RB][8];
    {
        const double inv = 1.0 / tr;                            // |G_ij| <= tr: entries in [-1, 1], lambda_max in (0, 1]
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = row0 + r, col = cq * 16 + j;
                g[r][j >> 1][j & 1] = (active && row < H && col < H) ? (float)(G[(long long)row * Hp + col] * inv) : 0.f;
            }
    }
    // start from the diagonal (a positive vector correlated with the dominant eigenvector of a PSD matrix)
    double q = (owner && myrow < H) ? G[(long long)myrow * Hp + myrow] / tr + 1e-3 : 0.0, qp = 0.0, beta = 0.0;
    {
        const double n0 = block_sum_dpp<NT>(q * q, red);
        q *= 1.0 / sqrt(n0);
    }
    if (owner) vbuf[myrow] = (float)q;
    __syncthreads();
    const int kmax = H < NP ? H : NP;
    int next_look = 12;
    double theta = 0.0, theta_prev = 0.0;
    for (int j = 0; j < kmax; ++j) {
        const float4* v4 = reinterpret_cast<const float4*>(&vbuf[cq * 16]);
        f32x2v a2[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) a2[r] = f32x2v{0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float4 v = v4[jj];
            const f32x2v va = {v.x, v.y}, vb = {v.z, v.w};
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                a2[r] = __builtin_elementwise_fma(g[r][2 * jj], va, a2[r]);
                a2[r] = __builtin_elementwise_fma(g[r][2 * jj + 1], vb, a2[r]);
            }
        }
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = rowN_sum<NCQ>(a2[r][0] + a2[r][1]);
        float mine = acc[0];
        if constexpr (RB == 4) mine = (cq & 3) == 0 ? acc[0] : ((cq & 3) == 1 ? acc[1] : ((cq & 3) == 2 ? acc[2] : acc[3]));
        double wv = owner ? (double)mine : 0.0;                 // (G q)[myrow]
        const double alpha = block_sum_dpp<NT>(q * wv, red);    // (q = 0 outside the owner lanes)
        wv = owner ? wv - alpha * q - beta * qp : 0.0;
        const double b2 = block_sum_dpp<NT>(wv * wv, red);
        const int k = j + 1;
        if (t == 0) { al[j] = alpha; be2[k] = b2; }
        const bool brk = !(b2 > 1e-28);                         // invariant subspace: T_k carries lambda_max of everything q_0 touches
        qp = q;
        beta = sqrt(b2);
        q = brk ? 0.0 : wv / beta;
        if (owner) vbuf[myrow] = (float)q;
        __syncthreads();                                        // q_(j+1), alpha_j, beta_(j+1) are in LDS; every read of q_j is over
        if (brk || k == next_look || k == kmax) {
            theta = tri_lambda_max(al, be2, k, theta_prev > 1e-9 ? theta_prev - 1e-9 : 0.0, 1.0 + 1e-6, imin, 4);
            if (brk || k == kmax || theta - theta_prev <= EIG_LANCZOS_TOL * theta) break;
            theta_prev = theta;
            next_look = k < 48 ? k + 12 : (k < 64 ? 64 : (k < 128 ? k + 32 : k + 64));
        }
    }
    __syncthreads();                                            // the LDS is reused by whatever the block runs next
    return theta;
}

// H <= 64: lambda_max by repeated squaring (kept for the small ranks: inside a 20-60 us pass launch of a narrow problem or a short row
// shard its ~11 us (H = 32) / ~25 us (H = 64) are on the critical path, and the Lanczos iteration below, with its three barriers per step,
// measured 2x slower there -- config 2 75 -> 126 us per sweep, the 8-way shard 163 -> 221).  spectral = 0: the trace instead.
//
// Method: NSQ repeated squarings T <- T*T / ||T*T||_F in fp32 (register-tiled, LDS-resident) drive T
// to the dominant eigenprojector; its largest-diagonal column is then polished by two fp64 power
// steps on the ORIGINAL matrix and lambda = Rayleigh quotient in fp64.  The quotient's error is
// quadratic in the vector's error; the worst case over all spectra is <= n/(2e*2^(NSQ+1)) relative
// (~2e-3 at NSQ = 10, n = 64; (1 - r) r^(2^(NSQ+1)) <= 1/(2e 2^NSQ) = 1.8e-4 for two competing eigenvalues of ratio r)
// and ~1e-12 whenever the top gap exceeds 0.5 %.  A cyclic Jacobi solve
// of the same matrix measured 2.9 ms on one CU; this is a few tens of microseconds.
constexpr int EIG_NSQ = 10;

// which = 0: GD -> S_LAMD, 1: GB -> S_LAMB_NEW.  256 threads, 2*NP*(NP+4) floats of LDS (NP = 16R).
template <int R>
__device__ __forceinline__ void eig_squaring_dev(double* __restrict__ st, StateLayout lay, int H, int spectral, int which,
                                        const int* __restrict__ ints, float* ldsf, int slot) {
    __shared__ double red[16];
    __shared__ int s_arg;
    if (load_stop(ints)) return;
    const int Hp = lay.Hp;
    const double* G = st + (which == 0 ? lay.GD() : lay.GB());
    double* scal = st + lay.scal();
    double tr = 0.0;
    for (int i = threadIdx.x; i < H; i += ctrl_nthreads()) tr += G[(long long)i * Hp + i];
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
    for (int t = threadIdx.x; t < NP * NP; t += ctrl_nthreads()) {
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
        if (sq > 0 && fabs(ss - 1.0) < 1e-5) break;          // uniform: converged to the dominant projector (the fp64
                                                             // Rayleigh quotient below squares what is left of the error)
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

// lambda_max of a symmetric PSD H x H matrix: which = 0: GD -> S_LAMD, 1: GB -> S_LAMB_NEW (the slot is the caller's).
// spectral = 0: Frobenius surrogate (trace) instead.  256 threads (the first 256 of a 512-thread pass launch).  H <= 64: repeated squaring
// (2 * NP * (NP + 4) floats of LDS); 64 < H <= 128: Lanczos (EIG_LDS_BYTES).
template <int R>
__device__ __forceinline__ void eig_dev(double* __restrict__ st, StateLayout lay, int H, int spectral, int which,
                                        const int* __restrict__ ints, float* ldsf, int slot) {
    if (R <= 4 && !(spectral & 2)) {                            // (uniform over the launch)
        eig_squaring_dev<R>(st, lay, H, spectral & 1, which, ints, ldsf, slot);
    } else {
        __shared__ double red[16];
        if (load_stop(ints)) return;
        const int Hp = lay.Hp;
        const double* G = st + (which == 0 ? lay.GD() : lay.GB());
        double* scal = st + lay.scal();
        double tr = 0.0;
        for (int i = threadIdx.x; i < H; i += ctrl_nthreads()) tr += G[(long long)i * Hp + i];
        tr = block_sum(tr, red);
        if (!(spectral & 1) || !(tr > 0.0) || !isfinite(tr)) {  // zero matrix -> 0; NaN propagates (loop exit on NaN d)
            if (threadIdx.x == 0) scal[slot] = tr;
            return;
        }
        const double theta = lanczos_lambda_max<16 * R, 256>(G, Hp, H, tr, ldsf);
        if (threadIdx.x == 0) scal[slot] = tr * theta;
    }
}

template <int R>
__global__ __launch_bounds__(256) void eig_kernel(double* __restrict__ st, StateLayout lay, int H, int spectral,
                                                  int do_d, int do_b, const int* __restrict__ ints) {
    extern __shared__ __attribute__((aligned(16))) float lds_eig[];
    const int which = blockIdx.x;           // 0: GD, 1: GB
    if ((which == 0 && !do_d) || (which == 1 && !do_b)) return;
    eig_dev<R>(st, lay, H, spectral, which, ints, lds_eig, which == 0 ? S_LAMD : S_LAMB_NEW);
}

// H > 128: the same iteration on 1024 threads (NP = 256)
__global__ __launch_bounds__(1024) void eig_lanczos_kernel(double* __restrict__ st, StateLayout lay, int H, int spectral,
                                                           int do_d, int do_b, const int* __restrict__ ints) {
    __shared__ double red[16];
    __shared__ __attribute__((aligned(16))) unsigned char lds_lz[EIG_LDS_BYTES];
    if (load_stop(ints)) return;
    const int which = blockIdx.x;           // 0: GD, 1: GB
    if ((which == 0 && !do_d) || (which == 1 && !do_b)) return;
    const int Hp = lay.Hp;
    const double* G = st + (which == 0 ? lay.GD() : lay.GB());
    double* scal = st + lay.scal();
    const int slot = which == 0 ? S_LAMD : S_LAMB_NEW;
    double tr = 0.0;
    for (int i = threadIdx.x; i < H; i += (int)blockDim.x) tr += G[(long long)i * Hp + i];
    tr = block_sum(tr, red);
    if (!(spectral & 1) || !(tr > 0.0) || !isfinite(tr)) {
        if (threadIdx.x == 0) scal[slot] = tr;
        return;
    }
    const double theta = lanczos_lambda_max<256, 1024>(G, Hp, H, tr, lds_lz);
    if (threadIdx.x == 0) scal[slot] = tr * theta;
}

// flags: bit0 est_covs->CA, bit1 est_covs->CB, bit2 est_var, bit3 compute d + loop bookkeeping,
//        bit4 (unused; tr(Y'BA') is always the direct sum st[GX] = sum (Y A) o BHat left by the kernel that produced BHat,
//             or by prepare_trYBA -- the Gram identity tr(KB o GB) it replaces is exact only for an un-rounded BHat),
//        bit5 S_LAMB_PREV already holds lambda_max of the old B'B (no rotation from S_LAMB_NEW)
//        bit6 split schedule (ctrl_chain): the d / loop part (bit3) is done by ctrl_loop_dev in another workgroup;
//             this call only files sigma2 / ELBO / residual in trace row it_row
__device__ __forceinline__ void ctrl_end_dev(double* __restrict__ st, StateLayout lay, int H, double Lg, double M,
                                             int flags, double eps, double* __restrict__ trace,
                                             int* __restrict__ ints, int it_row = -1) {
    // Inside a pass launch every dependent round trip to memory costs microseconds (the chip is streaming at full HBM
    // bandwidth around this workgroup), so everything the function needs is requested up front in ONE round -- the stop
    // flag, the scalars, the old CA/CB, the H x H arrays -- and the later stages work from registers and LDS.
    __shared__ double red[16];
    __shared__ double sc_s[32];
    __shared__ double dg[4][256];                                // diagonals of GA, SA, GB, SB
    const int Hp = lay.Hp;
    const double* GA = st + lay.GA();
    const double* GB = st + lay.GB();
    const double* SA = st + lay.SA();
    const double* SB = st + lay.SB();
    double* ca = st + lay.ca();
    double* cb = st + lay.cb();
    double* scal = st + lay.scal();
    // Split schedule (flags & 64): the loop test of THIS sweep runs beside this function in another workgroup of the same
    // launch (ctrl_loop_dev) and marks the stop flag with it_row + 1; CA/CB/sigma2 of the sweep it stops at are still due, so
    // that value does not count as "stopped" here -- whichever of the two workgroups gets there first.
    int stopped = load_stop(ints);
    if ((flags & 64) && stopped == it_row + 1) stopped = 0;
    const int hme = threadIdx.x;                                 // H <= 256 <= ctrl_nthreads(): one diagonal entry per thread
    const double scv = hme < 32 ? scal[hme] : 0.0;
    const double trdot = st[lay.GX()];                           // tr(Y'BA') = sum (Y A) o BHat, summed where BHat was produced
    double ca_h = hme < H ? ca[hme] : 1.0, cb_h = hme < H ? cb[hme] : 1.0;

    // tr((A'A + M SigmaA)(B'B + L SigmaB)); 8 elements per thread and round, all loads before the first use
    double t2 = 0.0;
    {
        constexpr int U = 8;
        const int sh = 31 - __clz(Hp), total = Hp * Hp;          // Hp is a power of two
        for (int e0 = threadIdx.x; e0 < total; e0 += U * ctrl_nthreads()) {
            double gb[U], ga[U], sa[U], sb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * ctrl_nthreads();
                const bool ok = e < total && (e >> sh) < H && (e & (Hp - 1)) < H;
                const int ee = ok ? e : 0;
                gb[u] = GB[ee]; ga[u] = GA[ee]; sa[u] = SA[ee]; sb[u] = SB[ee];
                if (!ok) { ga[u] = 0.0; sa[u] = 0.0; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * ctrl_nthreads();
                const int i = e >> sh, j = e & (Hp - 1);
                if (e < total && i == j && i < H) { dg[0][i] = ga[u]; dg[1][i] = sa[u]; dg[2][i] = gb[u]; dg[3][i] = sb[u]; }
                t2 += (ga[u] + M * sa[u]) * (gb[u] + Lg * sb[u]);
            }
        }
    }
    if (stopped) return;                                         // uniform
    if (hme < 32) sc_s[hme] = scv;
    t2 = block_sum(t2, red);                                     // (its barriers publish sc_s and dg)
    const double trYBA = trdot;
    const double resid = sc_s[S_TRYY] - 2.0 * trYBA + t2;
    double e = 0.0;
    if (hme < H) {
        const double gaa = dg[0][hme], saa = dg[1][hme], gbb = dg[2][hme], sbb = dg[3][hme];
        if (flags & 1) { ca_h = gaa / M + saa; ca[hme] = ca_h; }
        if (flags & 2) { cb_h = gbb / Lg + sbb; cb[hme] = cb_h; }
        // build-defined ELBO, with the post-update CA, CB, sigma2
        e += -(M / 2.0) * log(ca_h) - 0.5 * (gaa + M * saa) / ca_h;
        e += -(Lg / 2.0) * log(cb_h) - 0.5 * (gbb + Lg * sbb) / cb_h;
    }
    double sigma2 = sc_s[S_SIGMA2];
    if (flags & 4) sigma2 = resid / (Lg * M);
    e = block_sum(e, red);
    if (threadIdx.x == 0) {
        const double PI2 = 6.283185307179586476925286766559;
        double F = -(Lg * M / 2.0) * log(PI2 * sigma2) - resid / (2.0 * sigma2) + e;
        F += (M / 2.0) * sc_s[S_LOGDET_SA] + M * H / 2.0 + (Lg / 2.0) * sc_s[S_LOGDET_SB] + Lg * H / 2.0;
        scal[S_SIGMA2] = sigma2;
        scal[S_ELBO] = F;
        scal[S_RESID] = resid;
        scal[S_TRYBA] = trYBA;
        if (flags & 64) {
            if (trace) { trace[4 * it_row + 1] = sigma2; trace[4 * it_row + 2] = F; trace[4 * it_row + 3] = resid; }
        } else if (flags & 8) {
            const double d = sqrt(sc_s[S_LAMD] / sc_s[S_LAMB_PREV]);
            scal[S_D] = d;
            if (!(flags & 32)) scal[S_LAMB_PREV] = sc_s[S_LAMB_NEW];   // (32: lambda_max of the old B is written there directly)
            const int it = ints[I_ITERS];
            if (trace) { trace[4 * it + 0] = d; trace[4 * it + 1] = sigma2; trace[4 * it + 2] = F; trace[4 * it + 3] = resid; }
            ints[I_ITERS] = it + 1;
            const bool rerr = stop_on_remote_error(st, lay, ints);
            if (!(d > eps) || it + 1 >= ints[I_NITER] || rerr)            // src/vbmf.jl:193 (NaN d exits too)
                __hip_atomic_store(ints + I_STOP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // (no device-scope fence here: the consumers are later kernels -- the launch boundary publishes these stores -- or this
    //  same workgroup after the barrier; on gfx950 an agent-scope release writes back the XCD's whole L2, which costs
    //  several microseconds inside a pass that is filling that L2 with its own output)
    __syncthreads();
}

__global__ __launch_bounds__(1024) void ctrl_end_kernel(double* __restrict__ st, StateLayout lay, int H, double Lg,
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
    int mode;              // CTRL_* bits; 0: no control workgroups in this launch
    int it_row;            // sweep index (trace row) the CTRL_PREV_END part finalises
    int* sready; int sready_val;   // release flag for the Sigma table of part 0 (the pass's register epilogue waits on it)
};
// Schedule inside vbmf_run (sweep j; B_{j-1} is the factor the sweep starts from):
//   pass 1 of sweep j : [lambda_max(dB'dB) of sweep j-1, ctrl_end of sweep j-1]  then  SigmaA of sweep j
//   pass 2 of sweep j : SigmaB of sweep j  and  lambda_max(B_{j-1}'B_{j-1})  (the denominator of d_j; the
//                       Gram of B_{j-1} is still in place: post(B) of sweep j runs after this launch)
// so each pass carries a chain of similar length (~75 / ~65 us at H = 64).
enum : int { CTRL_PREV_END = 1, CTRL_COV_A = 2, CTRL_COV_B = 4, CTRL_EIG_BOLD = 8 };

// d = ||B_old - B_new||_2 / ||B_old||_2 (src/util.jl:27-29) and the loop test (src/vbmf.jl:193) of sweep it_row
__device__ __forceinline__ void ctrl_loop_dev(double* __restrict__ st, StateLayout lay, double eps,
                                              double* __restrict__ trace, int* __restrict__ ints, int it_row) {
    if (load_stop(ints)) return;
    if (threadIdx.x == 0) {
        double* scal = st + lay.scal();
        const double d = sqrt(scal[S_LAMD] / scal[S_LAMB_PREV]);
        scal[S_D] = d;
        if (trace) trace[4 * it_row + 0] = d;
        ints[I_ITERS] = it_row + 1;
        const bool rerr = stop_on_remote_error(st, lay, ints);
        if (!(d > eps) || it_row + 1 >= ints[I_NITER] || rerr)              // NaN d exits too
            __hip_atomic_store(ints + I_STOP, it_row + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // non-zero = stop; the
                                                                            // value names the sweep (see ctrl_end_dev)
    }
    // (see ctrl_end_dev: no device-scope fence; the stop flag itself is an agent-scope atomic)
    __syncthreads();
}

// The chain is split over the launch's first TWO workgroups (what bounds a pass launch on a short row shard is this
// chain, not the streaming: 83 us per launch on a 12.5k-row shard before the split):
//   part 0:  [CA, CB, sigma2, ELBO of the previous sweep]  ->  SigmaA (speculative: into the shadow, committed by
//            commit_cov_a_kernel after the launch iff the loop continues)            |  SigmaB
//   part 1:  lambda_max(dB'dB) -> d, trace, iteration count, stop                    |  lambda_max(B_old'B_old)
// Part 1's stop decision does not wait for part 0 and vice versa; CA/CB/sigma2 belong to the finished sweep and are
// due whether or not the loop stops, SigmaA belongs to the next one and is not.
// Durations of the chain's parts in 10 ns ticks of the constant 100 MHz clock, kept in ints[8..15] as four 64-bit slots
// (vbmf_debug_peek(VBMF_PEEK_CHAIN)): [0] ctrl_end, [1] SigmaA, [2] lambda_max(dB'dB) + loop test, [3] SigmaB
template <int R>
__device__ __forceinline__ void ctrl_chain(const CtrlArgs& a, void* lds, int part) {
    unsigned long long* stamp = reinterpret_cast<unsigned long long*>(a.ints + 8);
    const unsigned long long t0 = wall_clock64();
    if (part == 0) {
        if (a.mode & CTRL_PREV_END)
            ctrl_end_dev(a.st, a.lay, a.H, a.Lg, a.M, (a.end_flags & ~8) | 64, a.eps, a.trace, a.ints, a.it_row);
        const unsigned long long t1 = wall_clock64();
        if (a.mode & CTRL_COV_A)
            ctrl_cov_dev<R, 16>(a.st, a.lay, a.H, 0, a.Lg, a.S32, a.ints, reinterpret_cast<double*>(lds), true);
        if (a.mode & CTRL_COV_B) ctrl_cov_dev<R, 16>(a.st, a.lay, a.H, 1, a.M, a.S32, a.ints, reinterpret_cast<double*>(lds));
        if (threadIdx.x == 0) {
            const unsigned long long t2 = wall_clock64();
            if (a.mode & CTRL_PREV_END) stamp[0] = t1 - t0;
            if (a.mode & CTRL_COV_A) stamp[1] = t2 - t1;
            if (a.mode & CTRL_COV_B) stamp[3] = t2 - t1;
        }
        if (a.sready) {                                     // set unconditionally: a waiting epilogue must never hang
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(a.sready, a.sready_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        if (a.mode & CTRL_PREV_END) {
            eig_dev<R>(a.st, a.lay, a.H, a.spectral, 0, a.ints, reinterpret_cast<float*>(lds), S_LAMD);
            ctrl_loop_dev(a.st, a.lay, a.eps, a.trace, a.ints, a.it_row);
            if (threadIdx.x == 0) stamp[2] = wall_clock64() - t0;
        }
        if (a.mode & CTRL_EIG_BOLD)
            eig_dev<R>(a.st, a.lay, a.H, a.spectral, 1, a.ints, reinterpret_cast<float*>(lds), S_LAMB_PREV);
    }
}

// ---- vbls! as H x H algebra (examples/mil_util.jl:179-203, vbmf_parameters branch, no label mask) ----------------------------
// With B, SigmaB, CB frozen, P = Y'B is fixed, and so is S = P'P.  One iteration of  updateA!; updateCA!; updateSigma2!  then needs
// no M- or L-sized work at all:
//     K      = B'B + L SigmaB + sigma2 inv(CA)          SigmaA = sigma2 inv(K)                       (src/vbmf.jl:96-97)
//     A'A    = SigmaA S SigmaA / sigma2^2               (A = P SigmaA / sigma2, src/vbmf.jl:98)
//     CA_hh  = (A'A)_hh / M + SigmaA_hh                                                               (src/vbmf.jl:129-134)
//     tr(Y'BA') = tr(S SigmaA) / sigma2
//     sigma2 = (||Y||^2 - 2 tr(Y'BA') + tr((A'A + M SigmaA)(B'B + L SigmaB))) / (L M)                 (src/vbmf.jl:153-157)
// so the whole loop is ONE launch of one workgroup: the inverse by the blocked sweep, the two H x H products on the fp64 MFMA,
// everything LDS-resident, and A itself is formed once at the end from the last inv(K).  The MIL classifier calls vbls! with 150
// iterations on bags of a few tens of columns and H <= 10 (examples/mil_util.jl:473-479): per call 5.6-7.6 ms with one launch
// group per iteration (profiles/r03_f_vbls_mil.txt: break-even with NumPy on the host), most of it launch latency.
// 256 threads; LDS: four 16 NB x (16 NB + 2) fp64 images (W, S, SigmaA, T).  Blocks (I, J) are owned by wave (I NB + J) % 4.
// Reads GB, SB, ca, sigma2, ||Y||^2 and S (at st + lay.W1()); writes SA, ca, sigma2, log det SigmaA and the fp32 table inv(K).
template <int NB>
__global__ __launch_bounds__(256) void vbls_loop_kernel(double* __restrict__ st, StateLayout lay, int H, double Lg, double M,
                                                        int niter, float* __restrict__ SA32, int* __restrict__ ints) {
    extern __shared__ __attribute__((aligned(16))) double lds_vl[];
    __shared__ double red[16];
    __shared__ double ca_s[16 * NB], dg_s[16 * NB];
    constexpr int NP = 16 * NB, LD = NP + 2, IMG = NP * LD;
    constexpr int NOWN = (NB * NB + 3) / 4;                    // blocks per wave
    double* W = lds_vl;
    double* Sm = lds_vl + IMG;
    double* SAm = lds_vl + 2 * IMG;
    double* Tm = lds_vl + 3 * IMG;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, q16 = lane >> 4;
    const int Hp = lay.Hp, nbu = (H + 15) >> 4;
    double* scal = st + lay.scal();
    double sigma2 = scal[S_SIGMA2];
    const double trYY = scal[S_TRYY];
    const double* Sg = st + lay.W1();
    // K0 = B'B + L SigmaB of this wave's blocks (C/D layout), S into LDS, CA into LDS
    f64x4 k0[NOWN];
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int b = w + 4 * o, I = b / NB, J = b % NB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + q16 + 4 * r, j = 16 * J + c16;
            k0[o][r] = (b < NB * NB && i < H && j < H) ? st[lay.GB() + (long long)i * Hp + j] + Lg * st[lay.SB() + (long long)i * Hp + j] : 0.0;
        }
    }
    for (int t = threadIdx.x; t < NP * NP; t += 256) {
        const int i = t / NP, j = t % NP;
        Sm[i * LD + j] = (i < H && j < H) ? Sg[(long long)i * Hp + j] : 0.0;
    }
    if (threadIdx.x < NP) ca_s[threadIdx.x] = threadIdx.x < H ? st[lay.ca() + threadIdx.x] : 1.0;
    __syncthreads();
    double ldK = 0.0;
    int bad = 0;
    for (int it = 0; it < niter; ++it) {
        // K into the sweep image (every block: the sweep reads the upper ones)
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int b = w + 4 * o, I = b / NB, J = b % NB;
            if (b >= NB * NB) continue;
            f64x4 x = k0[o];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * I + q16 + 4 * r, j = 16 * J + c16;
                if (i == j) x[r] = i < H ? x[r] + sigma2 / ca_s[i] : 1.0;       // + sigma2 inv(CA); identity padding
            }
            blk_st_rows(W, LD, I, J, lane, x);
        }
        __syncthreads();
        PivAcc pv;
        blk_sweep<NB, 4>(W, LD, nbu, w, lane, pv);               // upper blocks of W = -inv(K)
        ldK = pv.logdet();
        bad |= pv.bad;
        // SigmaA = sigma2 inv(K), full symmetric image
        for (int t = threadIdx.x; t < NP * NP; t += 256) {
            const int i = t / NP, j = t % NP;
            SAm[i * LD + j] = (i < H && j < H) ? -sigma2 * (i <= j ? W[i * LD + j] : W[j * LD + i]) : 0.0;
        }
        __syncthreads();
        // T = SigmaA S  (blocks of this wave; SigmaA symmetric: the A operand of block (I, K) is the row read of block (K, I))
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int b = w + 4 * o, I = b / NB, J = b % NB;
            if (b >= NB * NB) continue;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int K = 0; K < NB; ++K) acc = blk_mma(blk_ld_rows(SAm, LD, K, I, lane), blk_ld_rows(Sm, LD, K, J, lane), acc);
            blk_st_rows(Tm, LD, I, J, lane, acc);
        }
        __syncthreads();
        // A'A = T SigmaA / sigma2^2 and the three reductions, block by block in registers
        const double is4 = 1.0 / (sigma2 * sigma2);
        double t2 = 0.0, trs = 0.0;
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int b = w + 4 * o, I = b / NB, J = b % NB;
            if (b >= NB * NB) continue;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int K = 0; K < NB; ++K) acc = blk_mma(blk_ld_cols(Tm, LD, I, K, lane), blk_ld_rows(SAm, LD, K, J, lane), acc);
            const f64x4 sa = blk_ld_rows(SAm, LD, I, J, lane), ss = blk_ld_rows(Sm, LD, I, J, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * I + q16 + 4 * r, j = 16 * J + c16;
                const double ata = acc[r] * is4;
                t2 += (ata + M * sa[r]) * k0[o][r];
                trs += ss[r] * sa[r];
                if (i == j && i < H) dg_s[i] = ata / M + sa[r];             // the new CA_hh
            }
        }
        t2 = block_sum(t2, red);
        trs = block_sum(trs, red);                               // (its barriers publish dg_s)
        const bool last = it + 1 == niter;
        if (last) {                                              // what the last updateA! leaves: SigmaA, inv(K) as the post kernel's table
            for (int t = threadIdx.x; t < Hp * Hp; t += 256) {
                const int i = t / Hp, j = t % Hp;
                const double v = (i < H && j < H) ? SAm[i * LD + j] : 0.0;
                st[lay.SA() + t] = v;
                SA32[t] = (float)(v / sigma2);
            }
            if (threadIdx.x == 0) scal[S_LOGDET_SA] = (double)H * log(sigma2) - ldK;
        }
        __syncthreads();
        if (threadIdx.x < NP) ca_s[threadIdx.x] = threadIdx.x < H ? dg_s[threadIdx.x] : 1.0;
        sigma2 = (trYY - 2.0 * trs / sigma2 + t2) / (Lg * M);
        __syncthreads();
    }
    if (threadIdx.x < H) st[lay.ca() + threadIdx.x] = ca_s[threadIdx.x];
    if (threadIdx.x == 0) scal[S_SIGMA2] = sigma2;
    if (bad) atomicExch(ints + I_ERR, 1);
}

// identity (H x H, zero-padded to Hp) as a post kernel's table: A = P I, whose Gram is S = P'P
__global__ void identity_table_kernel(float* __restrict__ S32, int H, int Hp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Hp * Hp) S32[t] = ((t / Hp) == (t % Hp) && (t / Hp) < H) ? 1.f : 0.f;
}
__global__ void copy_doubles_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[t];
}

__global__ void copy_scalar_kernel(double* st, StateLayout lay, int dst, int src) {
    if (threadIdx.x == 0 && blockIdx.x == 0) st[lay.scal() + dst] = st[lay.scal() + src];
}

}  // namespace vbmf
