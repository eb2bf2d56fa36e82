// blk_inverse.hpp -- fp64 inverse of a small SPD matrix (the H x H posterior precisions of src/vbmf.jl:96-97,110-111 and the
// per-column blocks of src/vbmf_sparse.jl:178-202) by a BLOCKED SYMMETRIC SWEEP staged in LDS:
//
//   * the matrix is cut into 16 x 16 blocks; block step K sweeps the diagonal block inside ONE wavefront -- no s_barrier, lane
//     traffic by DPP row broadcasts and v_permlane swaps only -- and every rank-16 update of the rest is four
//     v_mfma_f64_16x16x4_f64 per block;
//   * a whole inverse can therefore run on a single wavefront with no barrier at all (one matrix per wave: the per-column
//     blocks of full_cov), or on the waves of a workgroup with two barriers per block step (the control chain) instead of
//     one barrier + two LDS round trips per PIVOT (round 2's register-tiled Gauss-Jordan: 0.5 us per pivot inside a pass).
//
// Sweep operator on a block K of a symmetric matrix W (S = inv(W_KK)):
//     W_IJ <- W_IJ - W_IK S W_KJ      W_KJ <- S W_KJ      W_IK <- W_IK S      W_KK <- -S          (I, J != K)
// keeps W symmetric at every stage and leaves -inv(W) after all blocks; the scalar pivots met inside the diagonal blocks are
// the pivots of the unblocked elimination, so their logs sum to log det W.  Only blocks (I, J) with I <= J are stored
// authoritatively ("upper blocks"); a lower block is read as the transpose of its mirror.
//
// Register layouts (v_mfma_f64_16x16x4_f64; lane l = 16 q + c):
//     C/D:  4 doubles per lane, register r holds element [row q + 4 r][column c]
//     A:    one double per lane and k-chunk kk (4 chunks per 16-deep product):  A[row c][k = 4 kk + q]
//     B:    B[k = 4 kk + q][column c]
// so a block X loaded in the C/D layout IS the B operand of X (chunk kk = register kk) and the A operand of X' -- every
// operand of the sweep is a row-contiguous LDS read of an upper block or a column-strided one; with a leading dimension
// ld = 2 mod 32 doubles both are bank-conflict-free.
#pragma once
#include "common.hpp"

namespace vbmf {

typedef __attribute__((ext_vector_type(4))) double f64x4;

__device__ __forceinline__ double blk_rcp(double x) {          // 1/x to fp64 accuracy: v_rcp_f64 + two Newton steps (x > 0, normal)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// value of lane position K (0..15) of each 16-lane row, to every lane of that row
template <int K>
__device__ __forceinline__ double row_bcast16(double v) {
    // (mov_dpp: no `old` operand to initialise -- every lane is written: all rows and banks enabled, the source lane always exists)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + K, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// value of 16-lane row Q0 (0..3), same position in the row, to all four rows:
// x = [r0 r1 r2 r3]: v_permlane16_swap(x, x) -> [r0 r0 r2 r2], [r1 r1 r3 r3]; v_permlane32_swap of one of them with itself ->
// [ra ra ra ra], [rb rb rb rb]
template <int Q0>
__device__ __forceinline__ unsigned rows_bcast_u32(unsigned x) {
    const auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const unsigned y = (Q0 & 1) ? p[1] : p[0];
    const auto s = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return (Q0 & 2) ? s[1] : s[0];
}
template <int Q0>
__device__ __forceinline__ double rows_bcast(double v) {
    const unsigned lo = rows_bcast_u32<Q0>((unsigned)__double2loint(v));
    const unsigned hi = rows_bcast_u32<Q0>((unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}

// Pivots of the sweeps a wave has run: their product per diagonal block (16 pivots of an ARD-conditioned precision stay far
// inside the fp64 range), folded into mantissa + exponent between blocks; log det = log(m) + e ln 2.  A pivot that is not
// positive is flagged when it is met (two negative ones would hide in the product); a non-finite one poisons the product.
struct PivAcc {
    double m = 1.0; int e = 0; int bad = 0;
    double blk = 1.0;
    __device__ __forceinline__ void push(double d) {
        bad |= !(d > 0.0);
        blk *= d;
    }
    __device__ __forceinline__ void fold() {
        if (!isfinite(blk) || !(blk > 0.0)) bad = 1;
        int ex;
        m *= frexp(blk, &ex);
        e += ex;
        m = frexp(m, &ex);
        e += ex;
        blk = 1.0;
    }
    __device__ __forceinline__ double logdet() const { return log(m) + (double)e * 0.6931471805599453094; }
};

// In-wave symmetric sweep of the 16 x 16 block held in the C/D layout: a <- -inv(a).  One step of the sweep operator per
// pivot k, fully unrolled.  Pivot k needs column k at the lane's rows (lane position k of each 16-lane row: DPP row_newbcast) and
// row k at the lane's column, which lives in 16-lane row k & 3, register k >> 2 -- a cross-row broadcast (ds_bpermute).
// The step is ONE fma per element, a_ij <- a_ij - v_i w_j with v = column k, w = row k / d, the sweep's special cases (row k,
// column k, the corner) obtained EXACTLY by patching the operands instead of selecting among three results per element:
//     column-k lanes:  w := -1/d, a := 0     =>  -v_i (-1/d) = v_i / d
//     row-k lanes:     v := -1,   a := 0     =>  w_j = u_j / d          (corner: both patches => -1/d)
// (A variant that kept a and patched v := d - 1, w := 1 - 1/d is algebraically the same and one select cheaper per register --
//  and cancels catastrophically once an ARD precision reaches 1e10: found by the two-group model's H0 = H test.)
// "a := 0" clears the HIGH dword only: what is left is a denormal below 2^-1042, which the fma's rounding absorbs (one
// v_cndmask instead of two).
// The cross-row broadcast is taken off the critical path: row k+1 is fetched from the block BEFORE step k updates it (issued
// first, waited for last) and brought up to date with the step's own w:  u_{k+1} <- u_{k+1} - A[k+1][k] w  (column k: the
// patched form again).  What remains between two pivots is DPP(d) -> rcp -> w -> fma.
__device__ __forceinline__ double zero_if(bool z, double x) {          // z ? (effectively) 0 : x
    return __hiloint2double(z ? 0 : __double2hiint(x), __double2loint(x));
}
__device__ __forceinline__ double bperm_f64(int addr, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int KK>
__device__ __forceinline__ void sweep16_step(f64x4& a, double& u, int c, int q, const int (&rowaddr)[4], PivAcc& pv, double* pivout) {
    constexpr int Q0 = KK & 3, R0 = KK >> 2;
    constexpr int Q1 = (KK + 1) & 3, R1 = ((KK + 1) >> 2) & 3;
    double un = 0.0;
    if constexpr (KK < 15) un = bperm_f64(rowaddr[Q1], a[R1]);   // row k+1 as it stands BEFORE this step: A[k+1][c]
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = row_bcast16<KK>(a[r]);    // A[q + 4r][k]
    const double d = row_bcast16<KK>(u);                         // A[k][k]
    pv.push(d);
    if (pivout != nullptr && c == 0 && q == 0) pivout[KK] = d;   // (optional: the pivots one by one, for callers that fold them themselves)
    const double dinv = blk_rcp(d);
    const bool is_col = (c == KK), is_row = (q == Q0);
    const double w = is_col ? -dinv : u * dinv;
    v[R0] = is_row ? -1.0 : v[R0];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = fma(-v[r], w, zero_if(is_col || (r == R0 && is_row), a[r]));
    if constexpr (KK < 15) {
        const double vk1 = row_bcast16<KK>(un);                  // A[k+1][k] before the step
        u = fma(-vk1, w, zero_if(is_col, un));                   // row k+1 after it
    }
}
__device__ __forceinline__ void sweep16(f64x4& a, int lane, PivAcc& pv, double* pivout = nullptr) {
    const int c = lane & 15, q = lane >> 4;
    const int rowaddr[4] = {4 * c, 4 * (16 + c), 4 * (32 + c), 4 * (48 + c)};      // ds_bpermute byte addresses of lane (c, row Q)
    double u = bperm_f64(rowaddr[0], a[0]);                      // row 0
    sweep16_step<0>(a, u, c, q, rowaddr, pv, pivout);  sweep16_step<1>(a, u, c, q, rowaddr, pv, pivout);  sweep16_step<2>(a, u, c, q, rowaddr, pv, pivout);
    sweep16_step<3>(a, u, c, q, rowaddr, pv, pivout);  sweep16_step<4>(a, u, c, q, rowaddr, pv, pivout);  sweep16_step<5>(a, u, c, q, rowaddr, pv, pivout);
    sweep16_step<6>(a, u, c, q, rowaddr, pv, pivout);  sweep16_step<7>(a, u, c, q, rowaddr, pv, pivout);  sweep16_step<8>(a, u, c, q, rowaddr, pv, pivout);
    sweep16_step<9>(a, u, c, q, rowaddr, pv, pivout);  sweep16_step<10>(a, u, c, q, rowaddr, pv, pivout); sweep16_step<11>(a, u, c, q, rowaddr, pv, pivout);
    sweep16_step<12>(a, u, c, q, rowaddr, pv, pivout); sweep16_step<13>(a, u, c, q, rowaddr, pv, pivout); sweep16_step<14>(a, u, c, q, rowaddr, pv, pivout);
    sweep16_step<15>(a, u, c, q, rowaddr, pv, pivout);
    pv.fold();
}

// ---- block loads / stores (W: LDS or global, row-major, leading dimension ld doubles) ----------------------------------------
// C/D layout of block (bi, bj): register r = W[16 bi + q + 4 r][16 bj + c]   (= B operand chunks of the block, A chunks of its transpose)
__device__ __forceinline__ f64x4 blk_ld_rows(const double* W, int ld, int bi, int bj, int lane) {
    const double* p = W + (16 * bi + (lane >> 4)) * ld + 16 * bj + (lane & 15);
    return f64x4{p[0], p[4 * ld], p[8 * ld], p[12 * ld]};
}
// transposed read: register r = W[16 bi + c][16 bj + q + 4 r]   (= A operand chunks of the block, B chunks of its transpose)
__device__ __forceinline__ f64x4 blk_ld_cols(const double* W, int ld, int bi, int bj, int lane) {
    const double* p = W + (16 * bi + (lane & 15)) * ld + 16 * bj + (lane >> 4);
    return f64x4{p[0], p[4], p[8], p[12]};
}
__device__ __forceinline__ void blk_st_rows(double* W, int ld, int bi, int bj, int lane, const f64x4& x) {
    double* p = W + (16 * bi + (lane >> 4)) * ld + 16 * bj + (lane & 15);
    p[0] = x[0]; p[4 * ld] = x[1]; p[8 * ld] = x[2]; p[12 * ld] = x[3];
}
__device__ __forceinline__ void blk_st_cols(double* W, int ld, int bi, int bj, int lane, const f64x4& x) {   // stores x' into block (bi, bj)
    double* p = W + (16 * bi + (lane & 15)) * ld + 16 * bj + (lane >> 4);
    p[0] = x[0]; p[4] = x[1]; p[8] = x[2]; p[12] = x[3];
}
// acc += A * B with A, B given as operand chunks
__device__ __forceinline__ f64x4 blk_mma(const f64x4& a, const f64x4& b, f64x4 acc) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
    return acc;
}

// Blocked sweep of the NB x NB blocks of W (upper blocks authoritative) by the NW waves of the workgroup that call it
// (wave index w, all with the same arguments).  On return the upper blocks hold -inv(W); pv carries the pivots of every
// diagonal block (identical on every wave).  NW = 1: a single wave, no barrier anywhere.  NW > 1: two workgroup barriers per
// block step; every wave sweeps the diagonal block and forms the row panel R_J = inv(W_KK) W_KJ itself (redundant MFMAs, no
// exchange), the block pairs (I <= J) of the rank-16 update are dealt round-robin, wave 0 stores the panel.
// nb_used: blocks actually present (nb_used <= NB); n: matrix order (pivots beyond n are identity padding: their sweep is exact).
template <int NB, int NW>
__device__ __forceinline__ void blk_sweep(double* W, int ld, int nb_used, int w, int lane, PivAcc& pv) {
    for (int K = 0; K < nb_used; ++K) {
        f64x4 S = blk_ld_rows(W, ld, K, K, lane);
        sweep16(S, lane, pv);                                     // S = -inv(W_KK), symmetric
        const f64x4 Sn = -S;                                      // inv(W_KK): its own A operand (symmetric)
        // row panel R_J = inv(W_KK) * W_KJ, J != K  (W_KJ: block (K, J) for J > K, the transpose of block (J, K) for J < K)
        f64x4 R[NB];
#pragma unroll
        for (int J = 0; J < NB; ++J) {
            if (J >= nb_used || J == K) continue;                 // uniform
            const f64x4 b = J > K ? blk_ld_rows(W, ld, K, J, lane) : blk_ld_cols(W, ld, J, K, lane);
            R[J] = blk_mma(Sn, b, f64x4{0.0, 0.0, 0.0, 0.0});
        }
        // rank-16 update of the upper blocks: W_IJ -= T_I R_J, T_I = old W_IK (block (I, K) for I < K, transpose of (K, I) for I > K)
        int pairno = 0;
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            if (I >= nb_used || I == K) continue;
            f64x4 T = I < K ? blk_ld_cols(W, ld, I, K, lane) : blk_ld_rows(W, ld, K, I, lane);
            T = -T;
#pragma unroll
            for (int J = I; J < NB; ++J) {
                if (J >= nb_used || J == K) continue;
                if (NW == 1 || (pairno % NW) == w) {
                    f64x4 C = blk_ld_rows(W, ld, I, J, lane);
                    C = blk_mma(T, R[J], C);
                    blk_st_rows(W, ld, I, J, lane, C);
                }
                ++pairno;
            }
        }
        if constexpr (NW > 1) __syncthreads();                    // every reader of the old panel is done
        if (NW == 1 || w == 0) {
#pragma unroll
            for (int J = 0; J < NB; ++J) {
                if (J >= nb_used || J == K) continue;
                if (J > K) blk_st_rows(W, ld, K, J, lane, R[J]);
                else blk_st_cols(W, ld, J, K, lane, R[J]);        // block (J, K) = R_J'
            }
            blk_st_rows(W, ld, K, K, lane, S);
        }
        if constexpr (NW > 1) __syncthreads();
    }
}

// ---- 256 x 256: the blocked sweep with the matrix in REGISTERS (1024 threads = 16 waves) ----------------------------------------------
// A 256 x 256 fp64 matrix is 512 KiB: no LDS image (blk_sweep's form), but exactly the register file of one CU's worth of waves at 8 VGPRs
// per 16 x 16 block.  Only the 136 upper blocks are kept, dealt round-robin (block p = I 16 - I (I - 1) / 2 + J - I to wave p % 16: 8 or 9
// each, C/D layout).  Step K of the sweep:
//   (a) the owner of (K, K) sweeps it in-wave (S = -inv(W_KK)) and leaves inv(W_KK) in LDS; the owners of the panel blocks -- (K, J) for
//       J > K, (I, K) for I < K, i.e. the transposes of W_KI -- leave the OLD row panel T_J = W_KJ in an LDS strip (16 x 256, ld 258)
//   (b) the panel owners form R_J = inv(W_KK) T_J (one 16 x 16 x 16 product each) into a second strip; (K, J) keeps R_J
//   (c) every other block: W_IJ -= T_I' R_J (four fp64 MFMAs, both operands read from the strips); (I, K) takes R_I' back from the strip
// -- three workgroup barriers per step, no global traffic between load and store, no product formed twice (blk_sweep's waves each form
// the whole row panel themselves).  Replaces the Schur-complement inverse of rounds 1-2 (two register-tiled 128 x 128 Gauss-Jordans with a
// barrier per pivot + four VALU GEMMs through global scratch: 0.38 ms alone, 0.46 beside a pass).
// Kg: the SPD matrix (row-major, ld 256, identity-padded beyond its order); Out: its inverse, both triangles; pivs[256]: the pivots in
// order (their logs sum to log det; a non-positive one = not positive definite).  lds: INV256_LDS_DOUBLES doubles.
constexpr int INV256_LD = 258;
constexpr int INV256_LDS_DOUBLES = 2 * 16 * INV256_LD + 16 * 18;
// nb: 16 x 16 blocks actually present (the order rounded up; rows and columns beyond 16 nb are identity padding and are neither swept nor stored)
__device__ __forceinline__ void inv256_blk(const double* __restrict__ Kg, double* __restrict__ Out, double* lds, double* pivs, int nb = 16) {
    constexpr int LD = INV256_LD, NBLK = 136, NOWN = 9;
    double* Tst = lds;
    double* Rst = lds + 16 * LD;
    double* Sb = lds + 2 * 16 * LD;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bi[NOWN], bj[NOWN];
    f64x4 blk[NOWN];
#pragma unroll
    for (int t = 0; t < NOWN; ++t) {
        const int p = w + 16 * t;
        int I = 0, start = 0;
        while (I < 15 && p >= start + (16 - I)) { start += 16 - I; ++I; }    // (wave-uniform)
        const bool have = p < NBLK && I + (p - start) < nb;       // (bi <= bj < nb)
        bi[t] = have ? I : -1;
        bj[t] = have ? I + (p - start) : -1;
        blk[t] = f64x4{0.0, 0.0, 0.0, 0.0};
        if (have) blk[t] = blk_ld_rows(Kg, 256, bi[t], bj[t], lane);
    }
    for (int K = 0; K < nb; ++K) {
#pragma unroll
        for (int t = 0; t < NOWN; ++t) {
            if (bi[t] < 0) continue;
            if (bi[t] == K && bj[t] == K) {
                PivAcc pv;
                sweep16(blk[t], lane, pv, pivs + 16 * K);                    // blk = -inv(W_KK): the block's final value
                blk_st_rows(Sb, 18, 0, 0, lane, -blk[t]);
            } else if (bi[t] == K) {
                blk_st_rows(Tst, LD, 0, bj[t], lane, blk[t]);                // T_J = W_KJ
            } else if (bj[t] == K) {
                blk_st_cols(Tst, LD, 0, bi[t], lane, blk[t]);                // T_I = W_KI = (block (I, K))'
            }
        }
        __syncthreads();
        const f64x4 Sn = blk_ld_rows(Sb, 18, 0, 0, lane);                    // inv(W_KK), symmetric: its own A operand
#pragma unroll
        for (int t = 0; t < NOWN; ++t) {
            if (bi[t] < 0 || (bi[t] == K) == (bj[t] == K)) continue;         // panel blocks only (exactly one index = K)
            const int J = bi[t] == K ? bj[t] : bi[t];
            const f64x4 b = blk_ld_rows(Tst, LD, 0, J, lane);
            const f64x4 R = blk_mma(Sn, b, f64x4{0.0, 0.0, 0.0, 0.0});
            blk_st_rows(Rst, LD, 0, J, lane, R);
            if (bi[t] == K) blk[t] = R;                                      // (K, J) := R_J
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NOWN; ++t) {
            if (bi[t] < 0) continue;
            if (bi[t] != K && bj[t] != K) {
                const f64x4 T = -blk_ld_rows(Tst, LD, 0, bi[t], lane);       // A operand chunks of W_IK = (T_I)'
                const f64x4 Rj = blk_ld_rows(Rst, LD, 0, bj[t], lane);
                blk[t] = blk_mma(T, Rj, blk[t]);
            } else if (bj[t] == K && bi[t] != K) {
                blk[t] = blk_ld_cols(Rst, LD, 0, bi[t], lane);               // (I, K) := R_I'
            }
        }
        __syncthreads();
    }
    // the upper blocks hold -inv(W): both triangles out
#pragma unroll
    for (int t = 0; t < NOWN; ++t) {
        if (bi[t] < 0) continue;
        const f64x4 v = -blk[t];
        blk_st_rows(Out, 256, bi[t], bj[t], lane, v);
        if (bi[t] != bj[t]) blk_st_cols(Out, 256, bj[t], bi[t], lane, v);
    }
    __syncthreads();
}

}  // namespace vbmf
