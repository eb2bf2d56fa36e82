// sparse_kernels.hpp -- the ARD-sparse variant (src/vbmf_sparse.jl, full_cov=false, diag_var=false).
//
// The two streaming passes (Y'B, Y*A) and the B-side post/Gram kernels are shared with the basic
// model; what differs is element-wise over the M x H entries of vec(A') plus the H x H control algebra:
//   sparse_update_a    diagSigma = 1/(v_spread + CA); vec(A') = sigmaHat * diagSigma .* vec(B'Y); mask
//                      (src/vbmf_sparse.jl:214-232,244-246; QS1 `repeat(v, inner=M-1)` layout, QS2)
//   sparse_update_ca   beta = beta0 + (A.^2 + diagSigma)/2; CA = alpha ./ beta            (:284-288)
//   colsum             SigmaA = diag(sum_m diagSigma[m,:])                                 (:236-239)
//   sparse_cov_b       SigmaB = inv(diag(CB) + sigmaHat (A'A + SigmaA))                    (:263-265)
//   sparse_ctrl_end    CB/delta (:295-300), zeta/sigmaHat (:317-321), d and the loop test (:389,368)
//   sparse_lb_sums     the M*H-long sums lowerBound needs                                  (:442-443,453-455,461,467)
//   group_priors       the grouped models' hyper-prior fits (src/vbmf_dual.jl:393-434, src/vbmf_trial.jl:442-507)
// State block reuse: S_SIGMA2 holds sigmaHat (a PRECISION here), the `ca` strip holds delta, `cb` holds CB.
#pragma once
#include "common.hpp"
#include "ctrl_kernels.hpp"
#include "post_kernels.hpp"

namespace vbmf {

enum : int { S_ZETA = 12, S_ALPHA = 13, S_GAMMA = 14, S_ETA = 15, S_BETA0 = 16, S_DELTA0 = 17, S_ZETA0 = 18 };

// v[h] = sigmaHat * ||B[:,h]||^2 + L * SigmaB[h,h]   (:217; sigmaHat does NOT multiply L*SigmaB: QS2)
// diag_var (vsq != nullptr): v[h] = sum_l (sigma_l B[l,h])^2 + L * mean(sigma) * SigmaB[h,h]   (:211; S_SIGMA2 holds the mean)
__global__ void sparse_v_kernel(const double* __restrict__ st, StateLayout lay, int H, double Lg,
                                double* __restrict__ v, const double* __restrict__ vsq, const int* __restrict__ stop) {
    if (stop && *stop) return;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const long long hh = (long long)h * lay.Hp + h;
    if (vsq) v[h] = vsq[h] + Lg * st[lay.scal() + S_SIGMA2] * st[lay.SB() + hh];
    else v[h] = st[lay.scal() + S_SIGMA2] * st[lay.GB() + hh] + Lg * st[lay.SB() + hh];
}

// P: [Hp][ldP] fp32 (x fastest); A32/dS32/CA32: [Mp][Hp] fp32 row-major.  compat: QS1 layout.
__global__ __launch_bounds__(256) void sparse_update_a_kernel(const float* __restrict__ P, long long ldP,
                                                              const float* __restrict__ CA32,
                                                              const double* __restrict__ v,
                                                              const double* __restrict__ st, StateLayout lay,
                                                              float* __restrict__ A32, float* __restrict__ dS32,
                                                              const unsigned char* __restrict__ mask, int hmask_start,
                                                              long long M, int H, int Hp, int compat, int unit_sigma,
                                                              const int* __restrict__ stop) {
    if (stop && *stop) return;
    const double sig = unit_sigma ? 1.0 : st[lay.scal() + S_SIGMA2];      // diag_var: sigma is inside P (:230)
    const long long total = M * Hp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / Hp;
        const int h = (int)(i - m * Hp);
        float a = 0.f, ds = 0.f;
        if (h < H) {
            const long long p = m * H + h;                         // 0-based position in vec(A')
            long long vi = h;
            if (compat) vi = (p < H) ? p : (p - H) / (M - 1);      // repeat(v, inner = M-1) after the first H
            const double prec = v[vi] + (double)CA32[i];
            const double d = 1.0 / prec;
            ds = (float)d;
            a = (float)(sig * d * (double)P[(long long)h * ldP + m]);
            if (mask != nullptr && h >= hmask_start && mask[m]) a = 0.f;
        }
        A32[i] = a;
        dS32[i] = ds;
    }
}

// The same update with the operand tiles of the next pass written in the same kernel (sparse_update_a_kernel + retile_kernel in one, value for
// value): one wave per 32-row tile of A, the 32 x 32 block of the product brought in as whole 128-byte rows of P and turned through the wave's
// LDS tile (sparse_update_a_kernel's threads walk P down its columns, 40 KB apart: 8 x the bytes), the block formed in the accumulator-tile
// layout write_factor_tiles takes (lane = column h, registers = rows), A32 stored as the value the tiles encode.  At 10k x 256 (config 5):
// update_a 22-26 us + retile 16 us -> one launch of ~12 us.
template <int MODE, int NH>
__global__ __launch_bounds__(256) void sparse_update_a_tiles_kernel(const float* __restrict__ P, long long ldP,
                                                                    const float* __restrict__ CA32, const double* __restrict__ v,
                                                                    const double* __restrict__ st, StateLayout lay,
                                                                    float* __restrict__ A32, float* __restrict__ dS32,
                                                                    uint4* __restrict__ Ft,
                                                                    const unsigned char* __restrict__ mask, int hmask_start,
                                                                    long long M, int H, int compat, int unit_sigma, int XT,
                                                                    const int* __restrict__ stop) {
    constexpr int Hp = NH * 32;
    __shared__ float tbuf[4][32 * TB_LD];
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    // one wave per (row tile, column tile) block: XT * NH waves (one wave per row tile walking its NH blocks in turn was 2 % SLOWER than the
    // two kernels it replaces at 10k x 256 -- 313 waves for 2.5 M fp64 divisions)
    const int wv = blockIdx.x * 4 + wib;
    const int xt = wv / NH, h = wv % NH;
    if (xt >= XT) return;                                   // (no workgroup barrier below: the LDS tile is the wave's own)
    float* tb = tbuf[wib];
    const int c = lane & 31, half = lane >> 5;
    const int hr = lane >> 1, seg = lane & 1;               // the load's view: row hr of the block, 16 columns from 16 seg
    const long long x0 = (long long)xt * 32;
    const double sig = unit_sigma ? 1.0 : st[lay.scal() + S_SIGMA2];
    {
        const float4* src = reinterpret_cast<const float4*>(P + (long long)(h * 32 + hr) * ldP + x0 + seg * 16);
        float4 pv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = src[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* d = tb + hr * TB_LD + seg * 16 + 4 * j;
            d[0] = pv[j].x; d[1] = pv[j].y; d[2] = pv[j].z; d[3] = pv[j].w;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        const int hcol = h * 32 + c;
        f32x16 a, ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int xr = rho(r, half);
            const long long m = x0 + xr;
            float av = 0.f, dv = 0.f;
            if (m < M && hcol < H) {
                const long long p = m * H + hcol;                              // 0-based position in vec(A')
                long long vi = hcol;
                if (compat) vi = (p < H) ? p : (p - H) / (M - 1);              // repeat(v, inner = M-1) after the first H
                const double prec = v[vi] + (double)CA32[m * Hp + hcol];
                const double dd = 1.0 / prec;
                dv = (float)dd;
                av = (float)(sig * dd * (double)tb[c * TB_LD + xr]);
                if (mask != nullptr && hcol >= hmask_start && mask[m]) av = 0.f;
            }
            a[r] = av; ds[r] = dv;
        }
        write_factor_tiles<MODE, NH>(Ft, a, xt, h, lane);                       // a := the value the tiles encode
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = x0 + rho(r, half);
            if (m < M) { A32[m * Hp + hcol] = a[r]; dS32[m * Hp + hcol] = ds[r]; }
        }
    }
}

// dst[x][h] = sqrt(w[x]) * src[x][h]: B' diag(w) B is then the plain Gram of dst (full_cov with diag_var, :180-182)
__global__ void sqrt_rowscale_kernel(const float* __restrict__ src, const float* __restrict__ w, float* __restrict__ dst,
                                     long long n, int Hp, const int* __restrict__ stop) {
    if (stop && *stop) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = sqrtf(w[i / Hp]) * src[i];
}

// ---- full_cov = true (src/vbmf_sparse.jl:178-202, src/vbmf_dual.jl:218-243, src/vbmf_trial.jl:252-277) ------------------
// The reference builds invSigmaATVec = sigmaHat * kron(I_M, B'B + L SigmaB) + diag(CA) as a dense MH x MH matrix and
// inverts it.  That matrix is BLOCK DIAGONAL: M independent H x H blocks  K_m = sigmaHat (B'B + L SigmaB) + diag(CA[m,:]),
// so  Sigma_m = inv(K_m),  vec(A')[m,:] = sigmaHat Sigma_m (B'Y)[:,m],  diagSigmaATVec[m,:] = diag(Sigma_m),
// SigmaA = sum_m Sigma_m  (a full H x H matrix here) -- the per-column posterior update with the small covariance
// inverted in LDS/registers by one workgroup per column (gj_tiled).  The dense matrix is never formed.
// Block b walks columns m = b, b + grid, ...; its sum of Sigma_m goes to part[b] (folded in fixed order afterwards).
// Two columns per round and workgroup (gj_tiled_n<R, T, 2>, one barrier per pivot for both).  Measured: only +2.5 % over one
// column per round -- with two workgroups per CU the sweep is bound by instruction issue (~70 instructions per pivot step and
// wave), not by the LDS round trips; an fp64-MFMA rank-4 update per 16 x 16 tile is what would change that.
// NB columns per round and workgroup: 2 up to H = 64; 1 for 64 < H <= 128 (R = 8: a second 128 x 128 fp64 block would not fit
// the register file beside the running sum of the blocks).
template <int R, int T, int NB = 2>
__global__ __launch_bounds__(T * T) void sparse_update_a_full_kernel(const float* __restrict__ P, long long ldP,
                                                                     const float* __restrict__ CA32,
                                                                     const double* __restrict__ st, StateLayout lay,
                                                                     float* __restrict__ A32, float* __restrict__ dS32,
                                                                     const unsigned char* __restrict__ mask, int hmask_start,
                                                                     long long M, int H, int Hp, double Lg,
                                                                     double* __restrict__ part, int* __restrict__ ints,
                                                                     const double* __restrict__ Gw = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds_full[];
    if (load_stop(ints)) return;
    constexpr int NP = T * R;
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    double* strip = lds_full;                      // NB * 4 * NP
    double* pivs = lds_full + NB * 4 * NP;         // NB * NP
    double* pv = lds_full + NB * 5 * NP;           // NB * NP: (B'Y)[:, m] of the two columns
    // Gw != nullptr: heteroscedastic rows (:180-182, :192-193) -- K_m = B' diag(sigmaVec) B + L mean(sigmaVec) SigmaB + diag(CA[m,:])
    // with the weighted Gram in Gw ([Hp][Hp]), S_SIGMA2 = mean(sigmaVec), P = B' diag(sigmaVec) Y, and no sigmaHat on the mean
    const double sig = st[lay.scal() + S_SIGMA2];
    const double gsc = Gw != nullptr ? 1.0 : sig, msc = Gw != nullptr ? 1.0 : sig;
    const double* G = Gw != nullptr ? Gw : st + lay.GB();
    double k0[R][R], acc[R][R];
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const int i = ty + T * a, j = tx + T * b;
            k0[a][b] = (i < H && j < H) ? gsc * G[(long long)i * lay.Hp + j] + sig * Lg * st[lay.SB() + (long long)i * lay.Hp + j] : 0.0;
            acc[a][b] = 0.0;
        }
    int bad = 0;
    for (long long mb = blockIdx.x; mb < M; mb += (long long)NB * gridDim.x) {
        long long mm[NB];
        bool live[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) { mm[q] = mb + (long long)q * gridDim.x; live[q] = mm[q] < M; if (!live[q]) mm[q] = mb; }
        double w[NB][R][R];
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    const int i = ty + T * a, j = tx + T * b;
                    double v = (i == j) ? 1.0 : 0.0;                    // identity padding
                    if (i < H && j < H) v = k0[a][b] + ((i == j) ? (double)CA32[mm[q] * Hp + i] : 0.0);
                    w[q][a][b] = v;
                }
        for (int h = threadIdx.x; h < NB * NP; h += T * T) {
            const int q = h / NP, hh = h - q * NP;
            pv[h] = hh < H ? (double)P[(long long)hh * ldP + mm[q]] : 0.0;
        }
        gj_tiled_n<R, T, NB>(w, H, strip, pivs);
        __syncthreads();                                                // pv and pivs complete
        for (int k = threadIdx.x; k < NB * NP; k += T * T) {
            const int q = k / NP, kk = k - q * NP;
            if (kk < H) { const double pq = pivs[k]; if (!(pq > 0.0) || !isfinite(pq)) bad = 1; }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (!live[q]) continue;                                     // uniform (a dummy copy of column mb when M runs out)
            const long long m = mm[q];
#pragma unroll
            for (int a = 0; a < R; ++a) {
                const int i = ty + T * a;
                double sm = 0.0;
#pragma unroll
                for (int b = 0; b < R; ++b) sm += w[q][a][b] * pv[q * NP + tx + T * b];
                for (int off = T / 2; off > 0; off >>= 1) sm += __shfl_xor(sm, off);   // the T lanes of a row are contiguous
                if (tx == 0 && i < H) {
                    float av = (float)(msc * sm);
                    if (mask != nullptr && i >= hmask_start && mask[m]) av = 0.f;
                    A32[m * Hp + i] = av;
                }
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    const int j = tx + T * b;
                    if (i < H && j < H) {
                        acc[a][b] += w[q][a][b];
                        if (i == j) dS32[m * Hp + i] = (float)w[q][a][b];
                    }
                }
            }
        }
        __syncthreads();                                                // LDS is rewritten by the next round
    }
    if (bad) atomicExch(ints + I_ERR, 1);
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const int i = ty + T * a, j = tx + T * b;
            if (i < Hp && j < Hp) part[(long long)blockIdx.x * Hp * Hp + (long long)i * Hp + j] = acc[a][b];
        }
}
// Round 3, H <= 64: ONE COLUMN OF Y PER WAVEFRONT -- the column's H x H block staged in the wave's own LDS image and inverted
// there by the blocked symmetric sweep of blk_inverse.hpp (16 x 16 diagonal blocks swept inside the wave, rank-16 updates on
// the fp64 MFMA): no s_barrier anywhere in the loop over the columns, NW independent waves per workgroup.  What north_star
// describes ("the small H x H covariance staged and inverted in LDS per wavefront").  The part of K_m that does not depend on
// m, k0 = gsc G + sigma L SigmaB, stays in registers as the upper blocks in the MFMA's C/D layout and is written into the
// image with the column's diag(CA[m,:]) added on the way; so does the running sum of the Sigma_m.
//   NBK: 16-blocks per side (H <= 16 NBK); NW: waves per workgroup (NW images of 16 NBK x (16 NBK + 2) doubles of LDS).
template <int NBK, int NW>
__global__ __launch_bounds__(NW * 64) void sparse_update_a_full_wave_kernel(const float* __restrict__ P, long long ldP,
                                                                            const float* __restrict__ CA32,
                                                                            const double* __restrict__ st, StateLayout lay,
                                                                            float* __restrict__ A32, float* __restrict__ dS32,
                                                                            const unsigned char* __restrict__ mask, int hmask_start,
                                                                            long long M, int H, int Hp, double Lg,
                                                                            double* __restrict__ part, int* __restrict__ ints,
                                                                            const double* __restrict__ Gw = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds_fw[];
    if (load_stop(ints)) return;
    constexpr int NP = 16 * NBK, LD = NP + 2, NUP = NBK * (NBK + 1) / 2;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, q16 = lane >> 4;
    double* W = lds_fw + (size_t)w * NP * LD;
    double* pvec = lds_fw + (size_t)NW * NP * LD + w * NP;          // (B'Y)[:, m] of this wave's column
    const int nbu = (H + 15) >> 4;
    // (see sparse_update_a_full_kernel for the heteroscedastic form: Gw, gsc, msc)
    const double sig = st[lay.scal() + S_SIGMA2];
    const double gsc = Gw != nullptr ? 1.0 : sig, msc = Gw != nullptr ? 1.0 : sig;
    const double* G = Gw != nullptr ? Gw : st + lay.GB();
    f64x4 k0[NUP], acc[NUP];
    {
        int u = 0;
#pragma unroll
        for (int I = 0; I < NBK; ++I)
#pragma unroll
            for (int J = I; J < NBK; ++J, ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * I + q16 + 4 * r, j = 16 * J + c16;
                    k0[u][r] = (i < H && j < H) ? gsc * G[(long long)i * lay.Hp + j] + sig * Lg * st[lay.SB() + (long long)i * lay.Hp + j]
                                                : (i == j ? 1.0 : 0.0);                       // identity padding
                    acc[u][r] = 0.0;
                }
            }
    }
    int bad = 0;
    // the column's own inputs -- CA[m, :] (one value per lane and diagonal block) and (B'Y)[:, m] (one per lane) -- are requested one
    // column AHEAD: both are dependent trips to global memory (~2 us each) that would otherwise sit in front of every sweep
    const long long mstep = (long long)gridDim.x * NW;
    double ca_n[NBK], p_n = 0.0;
    auto fetch = [&](long long m) __attribute__((always_inline)) {
#pragma unroll
        for (int I = 0; I < NBK; ++I) {
            const int i = 16 * I + c16;
            ca_n[I] = (m < M && i < H) ? (double)CA32[m * Hp + i] : 0.0;
        }
        p_n = (m < M && lane < H) ? (double)P[(long long)lane * ldP + m] : 0.0;
    };
    long long m = (long long)blockIdx.x * NW + w;
    fetch(m);
    for (; m < M; m += mstep) {
        double ca_c[NBK];
#pragma unroll
        for (int I = 0; I < NBK; ++I) ca_c[I] = ca_n[I];
        const double p_c = p_n;
        fetch(m + mstep);
        {   // K_m = k0 + diag(CA[m,:]) into the image (upper blocks; the diagonal element of row 16 I + c sits in lane row c & 3, register c >> 2)
            int u = 0;
#pragma unroll
            for (int I = 0; I < NBK; ++I)
#pragma unroll
                for (int J = I; J < NBK; ++J, ++u) {
                    f64x4 x = k0[u];
                    if (I == J) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[r] += (q16 == (c16 & 3) && r == (c16 >> 2)) ? ca_c[I] : 0.0;
                    }
                    if (I < nbu && J < nbu) blk_st_rows(W, LD, I, J, lane, x);
                }
        }
        if (lane < NP) pvec[lane] = p_c;                                // (H <= 64 = lanes: one value per lane; zero beyond H)
        PivAcc pv;
        blk_sweep<NBK, 1>(W, LD, nbu, 0, lane, pv);                     // W's upper blocks = -Sigma_m
        bad |= pv.bad;
        // vec(A')[m,:] = msc Sigma_m (B'Y)[:, m]: lane i takes row i of the symmetric matrix (upper storage); a loop of fixed
        // length with the triangle chosen per element, so that the reads pipeline (bounds that depend on the lane do not)
        {
            const int i = lane < H ? lane : 0;
            double sm = 0.0;
            const int nj = 16 * nbu;                                    // (blocks beyond nbu were never written)
#pragma unroll 8
            for (int j = 0; j < nj; ++j) {
                const int lo = j < i ? j : i, hi = j < i ? i : j;
                sm += W[lo * LD + hi] * pvec[j];                        // (pvec is zero beyond H; W is the identity padding there)
            }
            if (lane < H) {
                float av = (float)(-msc * sm);
                if (mask != nullptr && lane >= hmask_start && mask[m]) av = 0.f;
                A32[m * Hp + lane] = av;
                dS32[m * Hp + lane] = (float)(-W[lane * LD + lane]);
            }
        }
        {
            int u = 0;
#pragma unroll
            for (int I = 0; I < NBK; ++I)
#pragma unroll
                for (int J = I; J < NBK; ++J, ++u)
                    if (I < nbu && J < nbu) acc[u] -= blk_ld_rows(W, LD, I, J, lane);
        }
    }
    if (bad) atomicExch(ints + I_ERR, 1);
    // fold the NW waves' sums through their images (fixed order), one dense Hp x Hp partial per workgroup
    {
        int u = 0;
#pragma unroll
        for (int I = 0; I < NBK; ++I)
#pragma unroll
            for (int J = I; J < NBK; ++J, ++u) blk_st_rows(W, LD, I, J, lane, acc[u]);
    }
    __syncthreads();
    double* mypart = part + (long long)blockIdx.x * Hp * Hp;
    for (int t = threadIdx.x; t < Hp * Hp; t += NW * 64) {
        const int i = t / Hp, j = t % Hp;
        double sm = 0.0;
        if (i < H && j < H) {
            const int a = i <= j ? i : j, b = i <= j ? j : i;
            for (int ww = 0; ww < NW; ++ww) sm += lds_fw[(size_t)ww * NP * LD + a * LD + b];
        }
        mypart[t] = sm;
    }
}

// 128 < H <= 256 (Hp = 256): a 256 x 256 fp64 block is the whole register file of a CU: the column's block is inverted by the control
// chain's register-resident blocked sweep (inv256_blk, blk_inverse.hpp) from / into a per-workgroup GLOBAL workspace ws[b] = [K | inv(K)]
// (1 MiB, L2).  One 1024-thread workgroup per column and round; the running sum of the blocks lives in part[b] (read-modify-write: the
// 64 values per thread would not fit beside the inverse's blocks).  ~0.15-0.2 ms per column (0.4 with the Schur-complement inverse of rounds
// 1-2): a correctness path for small M (the MIL callers gate full_cov by M H <= 3200; the reference's own route inverts the dense MH x MH matrix).
constexpr long long FULL256_WS = 2 * 256 * 256;                           // doubles per workgroup: the matrix and its inverse
__global__ __launch_bounds__(1024) void sparse_update_a_full256_kernel(const float* __restrict__ P, long long ldP,
                                                                       const float* __restrict__ CA32,
                                                                       const double* __restrict__ st, StateLayout lay,
                                                                       float* __restrict__ A32, float* __restrict__ dS32,
                                                                       const unsigned char* __restrict__ mask, int hmask_start,
                                                                       long long M, int H, double Lg, double* __restrict__ part,
                                                                       int* __restrict__ ints, const double* __restrict__ Gw,
                                                                       double* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) double lds_f256[];
    if (load_stop(ints)) return;
    constexpr int Hp = 256;
    double* Kg = ws + (long long)blockIdx.x * FULL256_WS;
    double* Ki = Kg + Hp * Hp;
    double* pivs = lds_f256 + INV256_LDS_DOUBLES;
    double* pv = pivs + 256;
    double* mypart = part + (long long)blockIdx.x * Hp * Hp;
    const double sig = st[lay.scal() + S_SIGMA2];
    const double gsc = Gw != nullptr ? 1.0 : sig, msc = Gw != nullptr ? 1.0 : sig;   // (see sparse_update_a_full_kernel)
    const double* G = Gw != nullptr ? Gw : st + lay.GB();
    for (int t = threadIdx.x; t < Hp * Hp; t += 1024) mypart[t] = 0.0;
    int bad = 0;
    for (long long m = blockIdx.x; m < M; m += gridDim.x) {
        for (int t = threadIdx.x; t < Hp * Hp; t += 1024) {
            const int i = t >> 8, j = t & 255;
            double v = (i == j) ? 1.0 : 0.0;                             // identity padding
            if (i < H && j < H) {
                v = gsc * G[(long long)i * lay.Hp + j] + sig * Lg * st[lay.SB() + (long long)i * lay.Hp + j];
                if (i == j) v += (double)CA32[m * Hp + i];
            }
            Kg[t] = v;
        }
        if (threadIdx.x < Hp) pv[threadIdx.x] = threadIdx.x < H ? (double)P[(long long)threadIdx.x * ldP + m] : 0.0;
        __syncthreads();
        inv256_blk(Kg, Ki, lds_f256, pivs, (H + 15) >> 4);
        __syncthreads();
        if (threadIdx.x < H) { const double pq = pivs[threadIdx.x]; if (!(pq > 0.0) || !isfinite(pq)) bad = 1; }
        {   // vec(A')[m,:] = msc * inv(K_m) (B'Y)[:, m]: four threads per row, 64 columns each
            const int i = threadIdx.x >> 2, q = threadIdx.x & 3;
            double sm = 0.0;
            for (int j = q * 64; j < q * 64 + 64; ++j) sm += Ki[i * Hp + j] * pv[j];
            sm += __shfl_xor(sm, 1);
            sm += __shfl_xor(sm, 2);
            if (q == 0 && i < H) {
                float av = (float)(msc * sm);
                if (mask != nullptr && i >= hmask_start && mask[m]) av = 0.f;
                A32[m * Hp + i] = av;
                dS32[m * Hp + i] = (float)Ki[i * Hp + i];
            }
        }
        for (int t = threadIdx.x; t < Hp * Hp; t += 1024) {
            const int i = t >> 8, j = t & 255;
            if (i < H && j < H) mypart[t] += Ki[t];
        }
        __syncthreads();                                                // the workspace and pv are rewritten by the next round
    }
    if (bad) atomicExch(ints + I_ERR, 1);
}

// SigmaA = sum over the blocks' partial sums, fixed order
__global__ __launch_bounds__(256) void full_sa_fold_kernel(const double* __restrict__ part, int nblocks, int Hp,
                                                           double* __restrict__ st, StateLayout lay, const int* __restrict__ stop) {
    if (stop && *stop) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Hp * Hp) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[(long long)b * Hp * Hp + e];
    st[lay.SA() + e] = s;
}

// Grouped models (src/vbmf_dual.jl:322-351, src/vbmf_trial.jl:357-400): the entries of vec(A') fall into up to three
// groups with their own Gamma hyper-prior -- g = 0: columns h < H0 (all rows); g = 1: h >= H0, rows m < M0; g = 2:
// h >= H0, rows m >= M0 (vbmf_dual: M0 = M, two groups).  Priors (alpha0g, beta0g) at scal[S_GPRI + 2g, +1], read from
// the state block (`grp` = scal: they change on the device every sweep under est_priors); the posterior shapes
// alpha0g + 1/2 (:324-325 / :359-361) are recorded at scal[S_GPOST + g] (what lowerBound reads).
// gpart != nullptr: per-block [sum log(beta), sum CA] of each group, what the hyper-prior fits need.
enum : int { S_GPRI = 21, S_GPOST = 27 };
__device__ __forceinline__ int ca_group(long long m, int h, int H0, long long M0) { return h < H0 ? 0 : (m < M0 ? 1 : 2); }
__global__ __launch_bounds__(256) void sparse_update_ca_kernel(const float* __restrict__ A32,
                                                               const float* __restrict__ dS32,
                                                               float* __restrict__ beta32, float* __restrict__ CA32,
                                                               double alpha, double beta0, long long M, int H, int Hp,
                                                               const int* __restrict__ stop,
                                                               double* __restrict__ grp, int H0, long long M0,
                                                               double* __restrict__ gpart) {
    __shared__ double sh[4][6];
    if (stop && *stop) return;
    double al[3] = {alpha, alpha, alpha}, b0[3] = {beta0, beta0, beta0};
    if (grp) {
#pragma unroll
        for (int g = 0; g < 3; ++g) { al[g] = grp[S_GPRI + 2 * g] + 0.5; b0[g] = grp[S_GPRI + 2 * g + 1]; }
        if (blockIdx.x == 0 && threadIdx.x == 0) { grp[S_GPOST] = al[0]; grp[S_GPOST + 1] = al[1]; grp[S_GPOST + 2] = al[2]; }
    }
    double s[6] = {0, 0, 0, 0, 0, 0};
    const long long total = M * Hp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / Hp;
        const int h = (int)(i - m * Hp);
        if (h >= H) continue;
        const int g = ca_group(m, h, H0, M0);
        const double a = (double)A32[i];
        const double b = (g == 0 ? b0[0] : (g == 1 ? b0[1] : b0[2])) + 0.5 * (a * a + (double)dS32[i]);
        const double ca = (g == 0 ? al[0] : (g == 1 ? al[1] : al[2])) / b;
        beta32[i] = (float)b;
        CA32[i] = (float)ca;
        if (gpart) {
            const double lb = log(b);
            if (g == 0) { s[0] += lb; s[1] += ca; } else if (g == 1) { s[2] += lb; s[3] += ca; } else { s[4] += lb; s[5] += ca; }
        }
    }
    if (!gpart) return;
#pragma unroll
    for (int k = 0; k < 6; ++k)
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) sh[w][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 6)
        gpart[(long long)blockIdx.x * 6 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// digamma / trigamma in fp64: recurrence up to x >= 8, then the asymptotic series
__device__ inline double digamma_dev(double x) {
    double r = 0.0;
    while (x < 8.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x
           - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132 - f * (691.0 / 32760 - f / 12))))));
}
__device__ inline double trigamma_dev(double x) {
    double r = 0.0;
    while (x < 8.0) { r += 1.0 / (x * x); x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + 1.0 / x + 0.5 * f
           + f / x * (1.0 / 6 - f * (1.0 / 30 - f * (1.0 / 42 - f * (1.0 / 30 - f * (5.0 / 66 - f * (691.0 / 2730 - f * 7.0 / 6))))));
}
// x with digamma(x) = y (Newton from Minka's start; digamma is concave increasing, so the iteration is monotone)
__device__ inline double digamma_inv_dev(double y) {
    double x = (y >= -2.22) ? exp(y) + 0.5 : -1.0 / (y + 0.57721566490153286);
    for (int it = 0; it < 12; ++it) {
        const double step = (digamma_dev(x) - y) / trigamma_dev(x);
        double xn = x - step;
        if (!(xn > 0.0)) xn = 0.5 * x;
        if (xn == x) break;
        x = xn;
    }
    return x;
}

// est_priors (src/vbmf_dual.jl:491-495, src/vbmf_trial.jl:565-572): alpha0g <- root of
//   n_g log(beta0g) - n_g digamma(x) + sum_i gammaELn(alpha_g, beta_i)   on [1e-10, 1e10]
// (dual :394-400; Roots.jl's fzero restated as the exact root, unchanged when the bracket holds no sign change -- the
// reference's `try ... end`), then beta0g <- n_g alpha0g / sum(CA_g) (:408-410).  One block.
__global__ __launch_bounds__(256) void group_priors_kernel(const double* __restrict__ gpart, int nblocks,
                                                           double* __restrict__ st, StateLayout lay, double M, int H, int H0,
                                                           double M0, const int* __restrict__ stop) {
    __shared__ double red[16];
    __shared__ double tot[6];
    if (stop && *stop) return;
    for (int k = 0; k < 6; ++k) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) s += gpart[(long long)b * 6 + k];
        s = block_sum(s, red);
        if (threadIdx.x == 0) tot[k] = s;
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double* scal = st + lay.scal();
    const double ng[3] = {M * (double)H0, M0 * (double)(H - H0), (M - M0) * (double)(H - H0)};
    for (int g = 0; g < 3; ++g) {
        const double n = ng[g];
        if (!(n > 0.0)) continue;                                  // empty group: nothing to fit
        const int ia = S_GPRI + 2 * g, ib = ia + 1;
        const double a_post = scal[S_GPOST + g];                   // = alpha0g + 1/2, set by this sweep's updateCA!
        const double y = log(scal[ib]) + digamma_dev(a_post) - tot[2 * g] / n;
        double a_new = scal[ia];
        // f(1e-10) > 0 always (digamma(1e-10) ~ -1e10); the bracket has a sign change iff y < digamma(1e10)
        if (isfinite(y) && y < 23.025850929890457 && y > -1e10) {
            a_new = digamma_inv_dev(y);
            a_new = fmin(fmax(a_new, 1e-10), 1e10);
        }
        scal[ia] = a_new;
        scal[ib] = n * a_new / tot[2 * g + 1];
    }
}

// SigmaA[h][h] = sum_m X[m][h] (src/vbmf_sparse.jl:236-239), fp64, fixed order, two stages:
// grid (Hp/32, COLSUM_CHUNKS) partial sums over row chunks, then one block per 32 columns folds the chunks.
// (One block per 32 columns walking all M rows was latency-bound: 321 us at M = 10 000, H = 256.)
constexpr int COLSUM_CHUNKS = 32;
// rs != nullptr: sum of (rs[m] * X[m][h])^2 instead (diag_var: sum_l (sigma_l B[l,h])^2, :211)
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ X, long long M, int H, int Hp,
                                                          double* __restrict__ part /* [COLSUM_CHUNKS][Hp] */,
                                                          const float* __restrict__ rs, const int* __restrict__ stop) {
    __shared__ double sh[8][32];
    if (stop && *stop) return;
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int h = blockIdx.x * 32 + c;
    const long long rows = (M + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS;
    const long long m0 = (long long)blockIdx.y * rows, m1 = m0 + rows < M ? m0 + rows : M;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (h < H) {
        long long m = m0 + g;
        if (rs) {
            for (; m < m1; m += 8) { const double t = (double)rs[m] * (double)X[m * Hp + h]; a0 += t * t; }
        } else {
            for (; m + 24 < m1; m += 32) {
                a0 += (double)X[m * Hp + h];
                a1 += (double)X[(m + 8) * Hp + h];
                a2 += (double)X[(m + 16) * Hp + h];
                a3 += (double)X[(m + 24) * Hp + h];
            }
            for (; m < m1; m += 8) a0 += (double)X[m * Hp + h];
        }
    }
    sh[g][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0) {
        double s = 0.0;
        for (int q = 0; q < 8; ++q) s += sh[q][c];
        part[(long long)blockIdx.y * Hp + h] = s;
    }
}
// vec_out != nullptr: the folded sums go to vec_out[h] only (no SigmaA write)
__global__ __launch_bounds__(256) void colsum_fold_kernel(const double* __restrict__ part, int H, int Hp,
                                                          double* __restrict__ st, StateLayout lay,
                                                          double* __restrict__ vec_out, const int* __restrict__ stop) {
    if (stop && *stop) return;
    // block b owns rows [32b, 32b+32) of SigmaA: the diagonal from the chunk partials, zeros elsewhere
    // (SigmaA is diagonal here: write full rows so stale off-diagonals never survive)
    __shared__ double diag[32];
    if (threadIdx.x < 32) {
        const int h = blockIdx.x * 32 + threadIdx.x;
        double s = 0.0;
        for (int q = 0; q < COLSUM_CHUNKS; ++q) s += part[(long long)q * Hp + h];
        diag[threadIdx.x] = h < H ? s : 0.0;
        if (vec_out) vec_out[h] = diag[threadIdx.x];
    }
    if (vec_out) return;
    __syncthreads();
    for (int t = threadIdx.x; t < 32 * Hp; t += blockDim.x) {
        const int r = t / Hp, j = t - r * Hp, h = blockIdx.x * 32 + r;
        st[lay.SA() + (long long)h * Hp + j] = (j == h) ? diag[r] : 0.0;
    }
}

// SigmaB = inv(diag(CB) + sigmaHat*(GA + SigmaA));  S32 = sigmaHat*SigmaB (so that B = Q*S32).
// s32_unit (diag_var, :256-261): S32 = SigmaB itself; the per-row sigma_l is applied to the rows of B afterwards
template <int R, int T>
__global__ __launch_bounds__(T * T) void sparse_cov_b_kernel(double* __restrict__ st, StateLayout lay, int H,
                                                             float* __restrict__ S32, int* __restrict__ ints, int s32_unit) {
    extern __shared__ __attribute__((aligned(16))) double lds_scb[];
    __shared__ double red[16];
    if (load_stop(ints)) return;
    constexpr int NP = T * R;
    const int Hp = lay.Hp;
    const int tx = threadIdx.x % T, ty = threadIdx.x / T;
    double* scal = st + lay.scal();
    const double sig = scal[S_SIGMA2];
    const double* cb = st + lay.cb();
    double* pivs;
    if constexpr (R == 8 && T == 32) {
        // 129 <= H <= 256 (Hp = 256): the blocked sweep with the matrix in registers (inv256_blk, blk_inverse.hpp)
        double* Kg = st + lay.W0();
        double* Ki = st + lay.W1();
        for (int t = threadIdx.x; t < 256 * 256; t += 1024) {
            const int i = t >> 8, j = t & 255;
            double v = (i == j) ? 1.0 : 0.0;
            if (i < H && j < H) {
                v = sig * (st[lay.GA() + (long long)i * Hp + j] + st[lay.SA() + (long long)i * Hp + j]);
                if (i == j) v += cb[i];
            }
            Kg[t] = v;
        }
        __syncthreads();
        pivs = lds_scb + INV256_LDS_DOUBLES;
        inv256_blk(Kg, Ki, lds_scb, pivs, (H + 15) >> 4);
        for (int t = threadIdx.x; t < 256 * 256; t += 1024) {
            const int i = t >> 8, j = t & 255;
            const double v = (i < H && j < H) ? Ki[t] : 0.0;
            st[lay.SB() + (long long)i * Hp + j] = v;
            S32[(long long)i * Hp + j] = (float)(s32_unit ? v : sig * v);
        }
    } else {
        // H <= 128: blocked sweep in LDS (ctrl_kernels.hpp, spd_inverse_lds4)
        static_assert(T == 16, "four waves");
        constexpr int NBc = NP / 16;
        double ldK; int badK;
        spd_inverse_lds4<NBc>(lds_scb, H, [&](int i, int j) {
            double v = sig * (st[lay.GA() + (long long)i * Hp + j] + st[lay.SA() + (long long)i * Hp + j]);
            if (i == j) v += cb[i];
            return v;
        }, &ldK, &badK);
        for (int t = threadIdx.x; t < Hp * Hp; t += 256) {
            const int i = t / Hp, j = t % Hp;
            const double v = (i < H && j < H) ? spd_inv_at<NBc>(lds_scb, i, j) : 0.0;
            st[lay.SB() + (long long)i * Hp + j] = v;
            S32[(long long)i * Hp + j] = (float)(s32_unit ? v : sig * v);
        }
        if (badK) atomicExch(ints + I_ERR, 1);
        if (threadIdx.x == 0) scal[S_LOGDET_SB] = -ldK;     // log det SigmaB
        return;
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
    if (threadIdx.x == 0) scal[S_LOGDET_SB] = -ld;          // log det SigmaB
}

// flags: bit1 est_cb -> CB/delta, bit2 sigma update, bit3 d + loop bookkeeping (bit4 unused: tr(B'Q) is always st[GX])
// t2part[b] = this block's share of sum (GA + SigmaA) o (GB + L SigmaB)  (:319): every input is final once the B update's Gram
// reduction has run, so the 4 x Hp^2 fp64 values (2 MiB at H = 256) are read by T2_BLOCKS workgroups beside the next Y'B pass
// instead of by sparse_ctrl_end_kernel's single workgroup on the critical path after it (one workgroup pulls ~40 GB/s: 54 us).
constexpr int T2_BLOCKS = 64;
__global__ __launch_bounds__(256) void sparse_t2_kernel(const double* __restrict__ st, StateLayout lay, int H, double Lg,
                                                        double* __restrict__ t2part, const int* __restrict__ ints) {
    __shared__ double red[16];
    if (load_stop(ints)) return;
    const int Hp = lay.Hp;
    const double* GA = st + lay.GA();
    const double* GB = st + lay.GB();
    const double* SA = st + lay.SA();
    const double* SB = st + lay.SB();
    const int hsh = 31 - __clz(Hp);                              // Hp is a power of two (32 .. 256)
    double t2 = 0.0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Hp * Hp; t += gridDim.x * blockDim.x)
        if ((t >> hsh) < H && (t & (Hp - 1)) < H) t2 += (GA[t] + SA[t]) * (GB[t] + Lg * SB[t]);   // SigmaA already summed over m (QS3)
    t2 = block_sum(t2, red);
    if (threadIdx.x == 0) t2part[blockIdx.x] = t2;
}

__global__ __launch_bounds__(1024) void sparse_ctrl_end_kernel(double* __restrict__ st, StateLayout lay, int H,
                                                              double Lg, int flags, double eps,
                                                              double* __restrict__ trace, int* __restrict__ ints,
                                                              const double* __restrict__ t2part = nullptr) {
    __shared__ double red[16];
    if (load_stop(ints)) return;
    const int Hp = lay.Hp;
    const double* GA = st + lay.GA();
    const double* GB = st + lay.GB();
    const double* SA = st + lay.SA();
    const double* SB = st + lay.SB();
    double* delta = st + lay.ca();
    double* cb = st + lay.cb();
    double* scal = st + lay.scal();
    double t2 = 0.0;
    if (t2part != nullptr) {                                     // sparse_t2_kernel's shares, fixed order
        if (threadIdx.x == 0)
            for (int b = 0; b < T2_BLOCKS; ++b) t2 += t2part[b];
    } else {
        for (int t = threadIdx.x; t < H * H; t += blockDim.x) {
            const int i = t / H, j = t - i * H;
            const long long ij = (long long)i * Hp + j;
            t2 += (GA[ij] + SA[ij]) * (GB[ij] + Lg * SB[ij]);    // SigmaA already summed over m (QS3)
        }
    }
    t2 = block_sum(t2, red);
    __syncthreads();
    if (flags & 2)
        for (int h = threadIdx.x; h < H; h += blockDim.x) {
            const long long hh = (long long)h * Hp + h;
            const double dl = scal[S_DELTA0] + 0.5 * GB[hh] + 0.5 * SB[hh];    // :297
            delta[h] = dl;
            cb[h] = scal[S_GAMMA] / dl;                                         // :298
        }
    if (threadIdx.x == 0) {
        const double trBQ = st[lay.GX()];                        // sum B o (Y A): direct, from the kernel that produced B
        scal[S_TRYBA] = trBQ;
        if (flags & 4) {
            const double zeta = scal[S_ZETA0] + 0.5 * scal[S_TRYY] - trBQ + 0.5 * t2;   // :317-319
            scal[S_ZETA] = zeta;
            scal[S_SIGMA2] = scal[S_ETA] / zeta;                                         // :321
        }
        if (flags & 8) {
            const double d = sqrt(scal[S_LAMD] / scal[S_LAMB_PREV]);
            scal[S_D] = d;
            scal[S_LAMB_PREV] = scal[S_LAMB_NEW];
            const int it = ints[I_ITERS];
            if (trace) { trace[4 * it + 0] = d; trace[4 * it + 1] = scal[S_SIGMA2]; trace[4 * it + 2] = 0.0; trace[4 * it + 3] = scal[S_ZETA]; }
            ints[I_ITERS] = it + 1;
            const bool rerr = stop_on_remote_error(st, lay, ints);       // row-sharded: another rank's (or this rank's) device error
            if (!(d > eps) || it + 1 >= ints[I_NITER] || rerr)
                __hip_atomic_store(ints + I_STOP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// partials[b][0..15] = [ sum log(beta) g0, g1, g2, sum' CA*(A^2 + dS), sum CA g0, g1, g2, sum' log(dS),
//                        count', sum' log(beta), sum' CA, 0... ] over the valid M x H entries (groups as in
// sparse_update_ca_kernel; the one-group sparse model passes H0 = H).  sum' / count' run over the entries that
// lowerBoundTrimmed keeps, |A| > trim (src/vbmf_sparse.jl:481); trim < 0 keeps everything (lowerBound).
constexpr int LB_NS = 16;
__global__ __launch_bounds__(256) void sparse_lb_sums_kernel(const float* __restrict__ A32, const float* __restrict__ dS32,
                                                             const float* __restrict__ CA32, const float* __restrict__ beta32,
                                                             long long M, int H, int Hp, int H0, long long M0,
                                                             double* __restrict__ partials, double trim) {
    __shared__ double sh[4][LB_NS];
    double s[LB_NS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long total = M * Hp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / Hp;
        const int h = (int)(i - m * Hp);
        if (h >= H) continue;
        const double a = A32[i], ds = dS32[i], ca = CA32[i], be = beta32[i];
        const int g = ca_group(m, h, H0, M0);
        const double lb = log(be);
        if (g == 0) { s[0] += lb; s[4] += ca; } else if (g == 1) { s[1] += lb; s[5] += ca; } else { s[2] += lb; s[6] += ca; }
        if (trim < 0.0 || fabs(a) > trim) {
            s[3] += ca * (a * a + ds); s[7] += log(ds);
            s[8] += 1.0; s[9] += lb; s[10] += ca;
        }
    }
#pragma unroll
    for (int k = 0; k < LB_NS; ++k)
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off);
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < LB_NS; ++k) sh[w][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < LB_NS)
        partials[(long long)blockIdx.x * LB_NS + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// ---- heteroscedastic rows (diag_var = true, src/vbmf_sparse.jl:207-212, 229-230, 256-261, 308-315) ----------
// ||Y[l,:]||^2 of the STORED matrix, from the pass-2 tiling (x = row l; both lane halves hold the same 32 rows)
template <int MODE>
__global__ __launch_bounds__(256) void row_sumsq_kernel(const uint4* __restrict__ Y2, int XT, int KS, long long L,
                                                        double* __restrict__ yrow) {
    const int lane = threadIdx.x & 63;
    const int xt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (xt >= XT) return;
    const uint4* p = Y2 + (long long)xt * KS * 64 + lane;
    double s = 0.0;
    for (int k = 0; k < KS; ++k) {
        const uint4 q = p[(long long)k * 64];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (MODE == MODE_F32) { const double v = (double)__uint_as_float(w[e]); s += v * v; }
            else {
                const double lo = (double)__uint_as_float(w[e] << 16), hi = (double)__uint_as_float(w[e] & 0xffff0000u);
                s += lo * lo + hi * hi;
            }
        }
    }
    s += __shfl_xor(s, 32);
    const long long l = (long long)xt * 32 + (lane & 31);
    if (lane < 32 && l < L) yrow[l] = s;
}

// G32 = A'A + SigmaA as an fp32 table, scal[S_TRDOT] = tr(G SigmaB)
__global__ __launch_bounds__(1024) void hetero_g_kernel(double* __restrict__ st, StateLayout lay, int H,
                                                        float* __restrict__ G32, const int* __restrict__ stop) {
    __shared__ double red[16];
    if (stop && *stop) return;
    const int Hp = lay.Hp;
    double t = 0.0;
    for (int i = threadIdx.x; i < Hp * Hp; i += blockDim.x) {
        const int r = i / Hp, c = i - r * Hp;
        const double g = (r < H && c < H) ? st[lay.GA() + i] + st[lay.SA() + i] : 0.0;
        G32[i] = (float)g;
        if (r < H && c < H) t += g * st[lay.SB() + i];
    }
    t = block_sum(t, red);
    if (threadIdx.x == 0) st[lay.scal() + S_TRDOT] = t;
}

// zeta_l = zeta0 + ||Y_l||^2/2 - Q[l,:].B[l,:] + (B_l' G B_l + tr(G SigmaB))/2;  sigma_l = etaVec / zeta_l  (:309-314)
// Q: [Hp][ldQ] (x fastest), B32: [Lp][Hp].  One thread per row; G32 is read as broadcast from L1/L2.
__global__ __launch_bounds__(256) void hetero_sigma_kernel(const float* __restrict__ Q, long long ldQ,
                                                           const float* __restrict__ B32, const float* __restrict__ G32,
                                                           const double* __restrict__ yrow, const double* __restrict__ st,
                                                           StateLayout lay, double etaVec, long long L, int H, int Hp,
                                                           double* __restrict__ zetav, double* __restrict__ sigv,
                                                           float* __restrict__ sig32, double* __restrict__ part,
                                                           const int* __restrict__ stop) {
    __shared__ double red[16];
    if (stop && *stop) return;
    const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double sg = 0.0;
    if (l < L) {
        const float* b = B32 + l * Hp;
        double qb = 0.0, quad = 0.0;
        for (int h = 0; h < H; ++h) {
            const float bh = b[h];
            qb += (double)Q[(long long)h * ldQ + l] * (double)bh;
            float acc = 0.f;
            const float* g = G32 + (long long)h * Hp;
            for (int k = 0; k < H; ++k) acc += g[k] * b[k];
            quad += (double)bh * (double)acc;
        }
        const double z = st[lay.scal() + S_ZETA0] + 0.5 * yrow[l] - qb + 0.5 * (quad + st[lay.scal() + S_TRDOT]);
        zetav[l] = z;
        sg = etaVec / z;
        sigv[l] = sg;
        sig32[l] = (float)sg;
    }
    sg = block_sum(sg, red);
    if (threadIdx.x == 0) part[blockIdx.x] = sg;
}
// scal[S_SIGMA2] = mean(sigmaVecHat)  (what :211 and :257 use)
__global__ __launch_bounds__(256) void hetero_mean_kernel(const double* __restrict__ part, int n, double L,
                                                          double* __restrict__ out, const int* __restrict__ stop) {
    __shared__ double red[16];
    if (stop && *stop) return;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s / L;
}

// vec(A') (index m*H + h, fp64 host order) <-> [Mp][Hp] fp32
__global__ void pack_vec_kernel(const double* __restrict__ src, long long M, int H, int Hp, long long Mp,
                                float* __restrict__ dst, float fill) {
    const long long total = Mp * Hp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / Hp; const int h = (int)(i % Hp);
        dst[i] = (m < M && h < H) ? (float)src[m * H + h] : fill;
    }
}
__global__ void unpack_vec_kernel(const float* __restrict__ src, long long M, int H, int Hp, double* __restrict__ dst) {
    const long long total = M * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / H; const int h = (int)(i % H);
        dst[i] = (double)src[m * Hp + h];
    }
}

}  // namespace vbmf
