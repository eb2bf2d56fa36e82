// post_kernels.hpp -- everything between the two streaming passes that touches M x H or L x H data.
//
//   post_kernel     Fac[x][h'] = sum_h (sum_s In[s][h][x]) * S[h][h']   (AHat = (Y'B) SigmaA/sigma2,
//                   src/vbmf.jl:98; BHat = (Y A) SigmaB/sigma2, src/vbmf.jl:112), label mask
//                   (src/vbmf.jl:101), fp32 row-major store, and the bf16(-hi/lo)/f32 MFMA operand
//                   tiles the NEXT streaming pass consumes.  Exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//   retile_kernel   operand tiles from an fp32 row-major factor (vbmf_set_state).
//   gram_kernel     G = Fac'Fac and, given the previous factor, D = (Prev-Fac)'(Prev-Fac) -- the
//                   difference is formed element-wise BEFORE squaring (src/util.jl:27-29 needs
//                   ||B_old - B_new|| near convergence where Gram differences would cancel).
//                   Accumulator tiles are fed back as both MFMA operands (sum over the row index
//                   that lives in registers), so no transpose is needed.
//   reduce kernels  fp64 reduction of per-chunk Gram slabs; fp32 sum of split-K slabs.
//   dot_kernel      per-workgroup shares of sum_{x,h} In[h][x]*Fac[x][h]  (tr(Y'BA') outside a B update), folded in fixed order.
#pragma once
#include "common.hpp"

namespace vbmf {

// ---- emit the operand tiles of one 32-row accumulator tile (rows = k of the next pass) ----------
// In the bf16 modes v is REPLACED by the value the tiles encode (hi, or hi+lo -- exactly representable
// in fp32), so the fp32 factor kept for the Grams is bit-for-bit what the next MFMA pass multiplies:
// the residual ||Y||^2 - 2tr(Y'BA') + tr(A'A B'B) then cancels consistently (src/vbmf.jl:154-156).
// frags (bf16 modes, optional): the packed fragments as written, frags[2*s + part] for k-step s of the tile
template <int MODE, int NH>
__device__ __forceinline__ void write_factor_tiles(uint4* __restrict__ Ft, f32x16& v, int xt, int nh, int lane,
                                                   u32x4v* frags = nullptr) {
    constexpr int NPART = ModeTraits<MODE>::NPART;
    if constexpr (MODE == MODE_F32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {                      // k-step (8 rows) 4*xt+g, element e = r&3
            uint4 o;
            const float e0 = v[4 * g + 0], e1 = v[4 * g + 1], e2 = v[4 * g + 2], e3 = v[4 * g + 3];
            o.x = fbits(e0); o.y = fbits(e1); o.z = fbits(e2); o.w = fbits(e3);
            Ft[((long long)(4 * xt + g) * NH + nh) * 64 + lane] = o;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {                      // k-step (16 rows) 2*xt+s, element e = r&7
            unsigned short hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = v[8 * s + e];
                hi[e] = f2bf(f);
                const float fh = bf2f(hi[e]);
                if constexpr (NPART == 2) {
                    lo[e] = f2bf(f - fh);
                    v[8 * s + e] = fh + bf2f(lo[e]);
                } else {
                    lo[e] = 0;
                    v[8 * s + e] = fh;
                }
            }
            uint4 o;
            o.x = hi[0] | ((unsigned)hi[1] << 16); o.y = hi[2] | ((unsigned)hi[3] << 16);
            o.z = hi[4] | ((unsigned)hi[5] << 16); o.w = hi[6] | ((unsigned)hi[7] << 16);
            const long long base = (long long)(2 * xt + s) * NPART;
            Ft[((base + 0) * NH + nh) * 64 + lane] = o;
            if (frags) frags[2 * s] = u32x4v{o.x, o.y, o.z, o.w};
            if constexpr (NPART == 2) {
                o.x = lo[0] | ((unsigned)lo[1] << 16); o.y = lo[2] | ((unsigned)lo[3] << 16);
                o.z = lo[4] | ((unsigned)lo[5] << 16); o.w = lo[6] | ((unsigned)lo[7] << 16);
                Ft[((base + 1) * NH + nh) * 64 + lane] = o;
                if (frags) frags[2 * s + 1] = u32x4v{o.x, o.y, o.z, o.w};
            }
        }
    }
}

// ---- tr(B'YA) = sum_{l,h} (Y A)[l,h] * BHat[l,h], formed where BHat is produced (src/vbmf.jl:154: trace(2*Y'*BHat*AHat')) ----
// One 32 x 32 block of it.  The product block is held "lane = row" (q[t] on lane (half, c) = Q[x0 + c][h0 + k(t, half)],
// k = rho for fragment-major products, 2t + half for row-major ones) and the new factor block "lane = column"
// (b[r] = B[x0 + rho(r, half)][h0 + c], AFTER write_factor_tiles made it the value the tiles encode), so B goes through a
// 32 x 33 LDS tile of this wave (conflict-free both ways) and comes back transposed.  Returns this lane's partial sum.
// (The Gram identity tr(KB * B'B) used before is exact only for an un-rounded B = Q * inv(KB): with B stored as bf16 hi + lo
//  and Sigma/sigma2 as an fp32 table its error, amplified ~400x by sigma2's cancellation, reached 1.5e-3 at 1200 x 900, H = 128.)
// 8 fp32 values -> their bf16 hi and bf16 lo parts as packed MFMA fragments (hi + lo carries ~16 significant bits)
__device__ __forceinline__ void split8_bf16(const float* v, u32x4v& hi, u32x4v& lo) {
    unsigned short h8[8], l8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { h8[e] = f2bf(v[e]); l8[e] = f2bf(v[e] - bf2f(h8[e])); }
    hi = u32x4v{h8[0] | ((unsigned)h8[1] << 16), h8[2] | ((unsigned)h8[3] << 16), h8[4] | ((unsigned)h8[5] << 16), h8[6] | ((unsigned)h8[7] << 16)};
    lo = u32x4v{l8[0] | ((unsigned)l8[1] << 16), l8[2] | ((unsigned)l8[3] << 16), l8[4] | ((unsigned)l8[5] << 16), l8[6] | ((unsigned)l8[7] << 16)};
}

// 8 fp32 values -> NT bf16 parts as packed MFMA fragments: part 0 = bf16(v), part p = bf16(v - part 0 - ... - part p-1).  NT = 2
// is split8_bf16; NT = 3 carries all 24 significant bits of an fp32 value (8 per part, the signs of the residuals give the rest).
template <int NT>
__device__ __forceinline__ void split8_parts(const float* v, u32x4v (&part)[NT]) {
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = v[e];
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        unsigned short b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { b[e] = f2bf(r[e]); r[e] -= bf2f(b[e]); }
        part[p] = u32x4v{b[0] | ((unsigned)b[1] << 16), b[2] | ((unsigned)b[3] << 16), b[4] | ((unsigned)b[5] << 16), b[6] | ((unsigned)b[7] << 16)};
    }
}

constexpr int SIGMA_NT = 3;      // bf16 parts per operand of the H <= 64 tile body's product (post_gram_tile_regs, load_sigma_table)
constexpr int TB_LD = 33;
template <bool RHO>
__device__ __forceinline__ float tile_dot_qb(const float (&q)[16], const f32x16& b, float* tb, int lane) {
    const int c = lane & 31, half = lane >> 5;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) tb[rho(r, half) * TB_LD + c] = b[r];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");      // LDS operations of one wave execute in issue order;
                                                                         // LDS-only fences: global loads may move across them
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s = fmaf(q[t], tb[c * TB_LD + (RHO ? rho(t, half) : 2 * t + half)], s);
    return s;
}
// this wave's share of tr(B'YA): lanes folded in fixed order, one double per wave (summed later in fixed order too)
__device__ __forceinline__ void store_wave_dot(double v, double* __restrict__ slot, int lane) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) *slot = v;
}

template <int MODE, int NH>
__device__ __forceinline__ void read_factor_tiles(const uint4* __restrict__ Ft, f32x16& v, int xt, int nh, int lane);

// ---- delta tiles (H >= 128, bf16 factor modes) ---------------------------------------------------------------------------
// d = old - new of one 32 x 32 block, formed element-wise where the new block is produced (the old one read from the factor's
// operand tiles BEFORE they are overwritten) and stored as bf16 hi + lo operand fragments in the layout of a two-part tile
// buffer: the delta-Gram (src/util.jl:27-29 needs ||B_old - B_new||) is then a plain tile Gram of that buffer -- 1 KiB wave
// loads, no 4-byte row gathers of the previous fp32 factor (they were what the delta-Gram kernel spent its time on).
template <int NH>
__device__ __forceinline__ void write_delta_tiles(uint4* __restrict__ Fd, const f32x16& oldv, const f32x16& newv, int xt, int nh, int lane) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = oldv[8 * s2 + e] - newv[8 * s2 + e];
        u32x4v hi, lo;
        split8_bf16(d, hi, lo);
        const long long base = (long long)(2 * xt + s2) * 2;
        Fd[((base + 0) * NH + nh) * 64 + lane] = uint4{hi[0], hi[1], hi[2], hi[3]};
        Fd[((base + 1) * NH + nh) * 64 + lane] = uint4{lo[0], lo[1], lo[2], lo[3]};
    }
}

// One wave per NXT consecutive 32-row tiles of the factor (NXT = 1 up to H = 64; from H = 128 on several tiles share
// every fetch of the H x H table, which no longer fits a wave's registers: 16 accumulator tiles per wave).
// In: [nslab][Hp][ldIn] fp32 (x fastest), S: [Hp][Hp] fp32 row-major, Fac: [XT*32][Hp] fp32 row-major.
#ifndef VBMF_POST_NXT8
#define VBMF_POST_NXT8 2          // tuning switch (A/B on the GPU): 32-row tiles per wave of the H = 256 post kernels
#endif
template <int NH> struct PostCfg { static constexpr int NXT = NH >= 8 ? VBMF_POST_NXT8 : 1; };    // (more tiles per wave at H = 128 are slower: 591 -> 830 us per 1M rows)

template <int MODE, int NH>
__global__ __launch_bounds__(256) void post_kernel(const float* __restrict__ In, long long ldIn, int nslab,
                                                   long long slabStride, const float* __restrict__ S,
                                                   float* __restrict__ Fac, uint4* __restrict__ Ft,
                                                   const unsigned char* __restrict__ mask, int hmask_start, int XT,
                                                   const int* __restrict__ stop, double* __restrict__ trpart = nullptr,
                                                   uint4* __restrict__ Fd = nullptr, int store_fac = 1) {
    constexpr int Hp = NH * 32;
    constexpr int NXT = PostCfg<NH>::NXT;
    __shared__ float tbuf[4][32 * TB_LD];
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int xt0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NXT;
    if (xt0 >= XT) {
        if (trpart && lane == 0) trpart[blockIdx.x * 4 + (threadIdx.x >> 6)] = 0.0;
        return;
    }
    const int c = lane & 31, half = lane >> 5;

    f32x16 acc[NXT][NH];
#pragma unroll
    for (int i = 0; i < NXT; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;

    for (int hin = 0; hin < NH; ++hin) {
        // (unrolling all 16 steps to put more of the 4-byte product loads in flight spills and is slower: 419 vs 382 us)
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
            const int hk = hin * 32 + 2 * t + half;               // contraction index of this lane-half
            float a[NXT];
#pragma unroll
            for (int i = 0; i < NXT; ++i) {
                // tiles past XT (ragged last wave) read tile XT-1 again and are dropped at the store
                const int xt = xt0 + i < XT ? xt0 + i : XT - 1;
                const float* ip = In + (long long)hk * ldIn + (long long)xt * 32 + c;
                float v = 0.f;
                for (int s = 0; s < nslab; ++s) v += ip[(long long)s * slabStride];
                a[i] = v;
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float b = S[(long long)hk * Hp + h * 32 + c];
#pragma unroll
                for (int i = 0; i < NXT; ++i)
                    acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b, acc[i][h], 0, 0, 0);
            }
        }
    }

    // accumulator layout: lane (c = h' in tile, half), register r -> row x0 + rho(r, half)
#pragma unroll
    for (int i = 0; i < NXT; ++i) {
        const int xt = xt0 + i;
        if (xt >= XT) break;
        const long long x0 = (long long)xt * 32;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int hcol = h * 32 + c;
            if (mask != nullptr && hcol >= hmask_start) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (mask[x0 + rho(r, half)]) acc[i][h][r] = 0.f;
            }
            // Fd (B side, bf16 factor modes): old - new of this block as delta tiles; the old block comes from the operand
            // tiles this wave is about to overwrite.  store_fac = 0 (inside the run loops): no fp32 copy per sweep.
            f32x16 oldv;
            if (Fd != nullptr) read_factor_tiles<MODE, NH>(Ft, oldv, xt, h, lane);
            write_factor_tiles<MODE, NH>(Ft, acc[i][h], xt, h, lane);
            if (Fd != nullptr) write_delta_tiles<NH>(Fd, oldv, acc[i][h], xt, h, lane);
            if (store_fac) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Fac[(x0 + rho(r, half)) * Hp + hcol] = acc[i][h][r];
            }
        }
    }
    if (trpart) {                                            // tr(B'YA): the product rows again (L2-hot), block by block
        double tr = 0.0;
        float* tb = tbuf[threadIdx.x >> 6];
#pragma unroll
        for (int i = 0; i < NXT; ++i) {
            const int xt = xt0 + i;
            if (xt >= XT) break;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float q[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float* ip = In + (long long)(h * 32 + 2 * t + half) * ldIn + (long long)xt * 32 + c;
                    float v = 0.f;
                    for (int sl = 0; sl < nslab; ++sl) v += ip[(long long)sl * slabStride];
                    q[t] = v;
                }
                tr += (double)tile_dot_qb<false>(q, acc[i][h], tb, lane);
            }
        }
        store_wave_dot(tr, trpart + blockIdx.x * 4 + (threadIdx.x >> 6), lane);
    }
}

// post_kernel for a FRAGMENT-MAJOR product (stream_gemm.hpp, frag_out; H >= 128, un-split pass): the product tile of
// (x tile, h tile) arrives as this lane's 16 accumulator registers in four 16-byte loads; register t is the A operand
// of MFMA step t, whose two k values are rows rho(t, half) of that h tile, so the table is read in that row order.
template <int MODE, int NH>
__global__ __launch_bounds__(256) void post_frag_kernel(const float4* __restrict__ In4, const float* __restrict__ S,
                                                        float* __restrict__ Fac, uint4* __restrict__ Ft,
                                                        const unsigned char* __restrict__ mask, int hmask_start, int XT,
                                                        const int* __restrict__ stop, double* __restrict__ trpart = nullptr,
                                                        uint4* __restrict__ Fd = nullptr, int store_fac = 1,
                                                        const uint4* __restrict__ Sf = nullptr) {
    constexpr int Hp = NH * 32;
    constexpr int NXT = PostCfg<NH>::NXT;
    __shared__ float tbuf[4][32 * TB_LD];
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int xt0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NXT;
    if (xt0 >= XT) {
        if (trpart && lane == 0) trpart[blockIdx.x * 4 + (threadIdx.x >> 6)] = 0.0;
        return;
    }
    const int c = lane & 31, half = lane >> 5;

    f32x16 acc[NXT][NH];
#pragma unroll
    for (int i = 0; i < NXT; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;

    for (int hin = 0; hin < NH; ++hin) {
        float a[NXT][16];
#pragma unroll
        for (int i = 0; i < NXT; ++i) {
            const int xt = xt0 + i < XT ? xt0 + i : XT - 1;
            const float4* t = In4 + (((long long)xt * NH + hin) * 64 + lane) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = t[q];
                a[i][4 * q] = v.x; a[i][4 * q + 1] = v.y; a[i][4 * q + 2] = v.z; a[i][4 * q + 3] = v.w;
            }
        }
        if constexpr (MODE == MODE_F32) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {                        // fully unrolled: a[i][t] stays in registers
                const int hk = hin * 32 + rho(t, half);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float b = S[(long long)hk * Hp + h * 32 + c];
#pragma unroll
                    for (int i = 0; i < NXT; ++i)
                        acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b, acc[i][h], 0, 0, 0);
                }
            }
        } else {
            // bf16 factor modes: the product is about to be rounded to bf16 hi + lo (2^-17) anyway, so product and table are
            // split the same way and multiplied as hi*hi + hi*lo + lo*hi (the dropped lo*lo is 2^-18 relative): three bf16
            // MFMAs per 16 contraction steps instead of eight exact-f32 ones at half the rate -- 5x less MFMA time.  The 8
            // values of a lane's fragment are its registers 8s..8s+7 (contraction order rho), on both operands.
            auto split8 = [](const float* v, u32x4v& hi, u32x4v& lo) __attribute__((always_inline)) { split8_bf16(v, hi, lo); };
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4v ah[NXT], al[NXT];
#pragma unroll
                for (int i = 0; i < NXT; ++i) split8(&a[i][8 * s2], ah[i], al[i]);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    u32x4v bh, bl;
                    if (Sf != nullptr) {                     // the table's fragments, split once per sweep (split_table_kernel)
                        const int combo = (hin * 2 + s2) * NH + h;
                        const uint4 xh = Sf[(combo * 2 + 0) * 64 + lane], xl = Sf[(combo * 2 + 1) * 64 + lane];
                        bh = u32x4v{xh.x, xh.y, xh.z, xh.w};
                        bl = u32x4v{xl.x, xl.y, xl.z, xl.w};
                    } else {
                        float b[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) b[e] = S[(long long)(hin * 32 + rho(8 * s2 + e, half)) * Hp + h * 32 + c];
                        split8(b, bh, bl);
                    }
#pragma unroll
                    for (int i = 0; i < NXT; ++i) {
                        acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh), acc[i][h], 0, 0, 0);
                        acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl), acc[i][h], 0, 0, 0);
                        acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh), acc[i][h], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NXT; ++i) {
        const int xt = xt0 + i;
        if (xt >= XT) break;
        const long long x0 = (long long)xt * 32;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int hcol = h * 32 + c;
            if (mask != nullptr && hcol >= hmask_start) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (mask[x0 + rho(r, half)]) acc[i][h][r] = 0.f;
            }
            // Fd (B side, bf16 factor modes): old - new of this block as delta tiles; the old block comes from the operand
            // tiles this wave is about to overwrite.  store_fac = 0 (inside the run loops): no fp32 copy per sweep.
            f32x16 oldv;
            if (Fd != nullptr) read_factor_tiles<MODE, NH>(Ft, oldv, xt, h, lane);
            write_factor_tiles<MODE, NH>(Ft, acc[i][h], xt, h, lane);
            if (Fd != nullptr) write_delta_tiles<NH>(Fd, oldv, acc[i][h], xt, h, lane);
            if (store_fac) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Fac[(x0 + rho(r, half)) * Hp + hcol] = acc[i][h][r];
            }
        }
    }
    if (trpart) {                                            // tr(B'YA): the product fragments again (L2-hot), block by block
        double tr = 0.0;
        float* tb = tbuf[threadIdx.x >> 6];
#pragma unroll
        for (int i = 0; i < NXT; ++i) {
            const int xt = xt0 + i;
            if (xt >= XT) break;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float4* t4 = In4 + (((long long)xt * NH + h) * 64 + lane) * 4;
                float q[16];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 v = t4[qd];
                    q[4 * qd] = v.x; q[4 * qd + 1] = v.y; q[4 * qd + 2] = v.z; q[4 * qd + 3] = v.w;
                }
                tr += (double)tile_dot_qb<true>(q, acc[i][h], tb, lane);
            }
        }
        store_wave_dot(tr, trpart + blockIdx.x * 4 + (threadIdx.x >> 6), lane);
    }
}

// ---- the inverse of write_factor_tiles: the fp32 factor values a tile's operand fragments encode (hi + lo) ------
template <int MODE, int NH>
__device__ __forceinline__ void read_factor_tiles(const uint4* __restrict__ Ft, f32x16& v, int xt, int nh, int lane) {
    constexpr int NPART = ModeTraits<MODE>::NPART;
    if constexpr (MODE == MODE_F32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint4 o = Ft[((long long)(4 * xt + g) * NH + nh) * 64 + lane];
            v[4 * g + 0] = bitsf(o.x); v[4 * g + 1] = bitsf(o.y); v[4 * g + 2] = bitsf(o.z); v[4 * g + 3] = bitsf(o.w);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const long long base = (long long)(2 * xt + s) * NPART;
            const uint4 h = Ft[((base + 0) * NH + nh) * 64 + lane];
            const unsigned hw[4] = {h.x, h.y, h.z, h.w};
            unsigned lw[4] = {0u, 0u, 0u, 0u};
            if constexpr (NPART == 2) {
                const uint4 l = Ft[((base + 1) * NH + nh) * 64 + lane];
                lw[0] = l.x; lw[1] = l.y; lw[2] = l.z; lw[3] = l.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned short hb = (unsigned short)(hw[e >> 1] >> (16 * (e & 1)));
                float f = bf2f(hb);
                if constexpr (NPART == 2) f += bf2f((unsigned short)(lw[e >> 1] >> (16 * (e & 1))));
                v[8 * s + e] = f;
            }
        }
    }
}

// fp32 row-major factor from its operand tiles (the lazy counterpart of the register epilogue's fp32 store)
template <int MODE, int NH>
__global__ __launch_bounds__(256) void untile_factor_kernel(const uint4* __restrict__ Ft, float* __restrict__ Fac, int XT) {
    constexpr int Hp = NH * 32;
    const int lane = threadIdx.x & 63;
    const int xt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (xt >= XT) return;
    const int c = lane & 31, half = lane >> 5;
    const long long x0 = (long long)xt * 32;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        f32x16 v;
        read_factor_tiles<MODE, NH>(Ft, v, xt, h, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) Fac[(x0 + rho(r, half)) * Hp + h * 32 + c] = v[r];
    }
}

// ---- one 32-row tile of "post + Gram" from a product tile that is still in accumulator registers --------------
// q[hin][r] on lane (half, c) holds (Y A)[x0 + c][hin*32 + rho(r, half)] -- the streaming kernel's accumulator as it
// stands.  An exact-f32 MFMA of k = 2 takes its two k values from the two lane halves, and the contraction may run in
// any k order as long as both operands agree, so register r IS the A operand of instruction r when the table operand
// is read in the same permuted order: S[hin*32 + rho(r, half)][h*32 + c] (from the workgroup's LDS copy).  No store, no reload,
// no shuffle between the product and the update B = (Y A) SigmaB / sigma2 (src/vbmf.jl:112).
// BSIDE = true: the B update (delta-Gram against the previous factor rows pv, tr(B'YA)); false: the A update (label mask
// of src/vbmf.jl:101, no delta-Gram, no trace).
template <int MODE, int NH, bool BSIDE = true>
__device__ __forceinline__ void post_gram_tile_regs(const f32x16 (&q)[NH], const float* stab /* LDS: load_sigma_table's image */, int xt,
                                                    float* __restrict__ Fac, const float* __restrict__ Prev,
                                                    uint4* __restrict__ Ft, int lane,
                                                    f32x16 (&G)[NH * (NH + 1) / 2], f32x16 (&D)[NH * (NH + 1) / 2],
                                                    const f32x16 (&pv)[NH], int store_fac, float* tb, double& trd,
                                                    const unsigned char* __restrict__ mask = nullptr, int hmask_start = 0) {
    constexpr int Hp = NH * 32;
    const int c = lane & 31, half = lane >> 5;
    const long long x0 = (long long)xt * 32;
    (void)Prev;
    f32x16 acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    // f32 mode: exact-f32 MFMA, the table as plain fp32 in LDS.  bf16 factor modes: the SIX-TERM bf16 product of post_frag2_kernel --
    // product and table split into three bf16 parts each (together all 24 significant bits of the fp32 values), the products with
    // part indices i + j <= 2 kept: what is dropped is 2^-24 relative, an fp32 rounding.  48 MFMAs of 32 cycles per 32-row tile at
    // H = 64 instead of 64 of 64.  (The THREE-term product hi*hi + hi*lo + lo*hi -- 2^-17 per term -- was built first and
    // rejected: B = Q * inv(K_B) cancels by the condition number of K_B, BHat off by 5e-4 .. 2.6e-3 on rank-deficient data,
    // profiles/r02_e_three_term_product.txt.)  The table's fragments come pre-split from LDS (load_sigma_frags), in order of use.
    if constexpr (MODE == MODE_F32) {
#pragma unroll
        for (int hin = 0; hin < NH; ++hin)
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float a = q[hin][t];
                    acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, stab[(hin * 32 + rho(t, half)) * Hp + h * 32 + c], acc[h], 0, 0, 0);
                }
    } else {
        constexpr int NT = SIGMA_NT;
        const u32x4v* sf = reinterpret_cast<const u32x4v*>(stab);
#pragma unroll
        for (int hin = 0; hin < NH; ++hin)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = q[hin][8 * s2 + e];
                u32x4v ap[NT];
                split8_parts<NT>(v, ap);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const int combo = (hin * 2 + s2) * NH + h;
                    u32x4v bp[NT];
#pragma unroll
                    for (int p = 0; p < NT; ++p) bp[p] = sf[(combo * NT + p) * 64 + lane];
#pragma unroll
                    for (int pa = 0; pa < NT; ++pa)
#pragma unroll
                        for (int pb = 0; pa + pb < NT; ++pb)
                            acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[pa]), __builtin_bit_cast(bf16x8, bp[pb]), acc[h], 0, 0, 0);
                }
            }
    }
    // In the bf16 modes the new tile IS hi + lo, and its operand fragments (lane = column, 8 consecutive k per lane) are
    // both operands of the bf16 MFMA with the row index as k: F'F = hi'hi + hi'lo + lo'hi + lo'lo, every product exact,
    // at 1/8 of the exact-f32 MFMA time.  The delta d = old - new is re-split into hi + lo the same way (2^-17 on d).
    u32x4v fr[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        if constexpr (!BSIDE) {
            if (mask != nullptr && h * 32 + c >= hmask_start) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (mask[x0 + rho(r, half)]) acc[h][r] = 0.f;
            }
        }
        write_factor_tiles<MODE, NH>(Ft, acc[h], xt, h, lane, MODE == MODE_F32 ? nullptr : fr[h]);
        if (store_fac) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Fac[(x0 + rho(r, half)) * Hp + h * 32 + c] = acc[h][r];
        }
    }
    if constexpr (BSIDE) {   // tr(B'YA) of this tile: the product is still in registers, the new factor is now exactly what the tiles encode
        float tsum = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float qf[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) qf[t] = q[h][t];
            tsum += tile_dot_qb<true>(qf, acc[h], tb, lane);
        }
        trd += (double)tsum;
    }
    if constexpr (MODE == MODE_F32) {
        int p = 0;
#pragma unroll
        for (int h1 = 0; h1 < NH; ++h1)
#pragma unroll
            for (int h2 = h1; h2 < NH; ++h2, ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float u = acc[h1][r], v = acc[h2][r];
                    G[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(u, v, G[p], 0, 0, 0);
                    if constexpr (BSIDE) {
                        const float du = pv[h1][r] - u, dv = pv[h2][r] - v;
                        D[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(du, dv, D[p], 0, 0, 0);
                    }
                }
    } else {
        constexpr int NPART = ModeTraits<MODE>::NPART;
        u32x4v dr[NH][4];
        if constexpr (BSIDE) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                unsigned short hi[8], lo[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = pv[h][8 * s2 + e] - acc[h][8 * s2 + e];
                    hi[e] = f2bf(d);
                    lo[e] = f2bf(d - bf2f(hi[e]));
                }
                dr[h][2 * s2] = u32x4v{hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16),
                                       hi[4] | ((unsigned)hi[5] << 16), hi[6] | ((unsigned)hi[7] << 16)};
                dr[h][2 * s2 + 1] = u32x4v{lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16),
                                           lo[4] | ((unsigned)lo[5] << 16), lo[6] | ((unsigned)lo[7] << 16)};
            }
        }
        int p = 0;
#pragma unroll
        for (int h1 = 0; h1 < NH; ++h1)
#pragma unroll
            for (int h2 = h1; h2 < NH; ++h2, ++p)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                    for (int pa = 0; pa < NPART; ++pa)
#pragma unroll
                        for (int pb = 0; pb < NPART; ++pb)
                            G[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[h1][2 * s2 + pa]),
                                                                           __builtin_bit_cast(bf16x8, fr[h2][2 * s2 + pb]), G[p], 0, 0, 0);
                    if constexpr (BSIDE) {
#pragma unroll
                        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                            for (int pb = 0; pb < 2; ++pb)
                                D[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dr[h1][2 * s2 + pa]),
                                                                               __builtin_bit_cast(bf16x8, dr[h2][2 * s2 + pb]), D[p], 0, 0, 0);
                    }
                }
    }
}

// The Sigma / sigma2 table of post_gram_tile_regs into the workgroup's LDS, by all 256 threads.  The caller barriers.
//   f32 mode:          the plain fp32 table (4 Hp^2 bytes, 16-byte loads)
//   bf16 factor modes: pre-split into SIGMA_NT bf16 parts as MFMA B-operand fragments in order of use,
//                      [(hin, s2, h)][part][lane] (1 KiB each: 24 KiB at H = 64), read back conflict-free with ds_read_b128
template <int MODE, int NH> constexpr int sigma_lds_floats() {
    return MODE == MODE_F32 ? NH * 32 * NH * 32 : NH * 2 * NH * SIGMA_NT * 256;
}
template <int MODE, int NH>
__device__ __forceinline__ void load_sigma_table(float* stab, const float* __restrict__ S) {
    constexpr int Hp = NH * 32;
    if constexpr (MODE == MODE_F32) {
        for (int i = threadIdx.x; i < Hp * Hp / 4; i += 256)
            reinterpret_cast<float4*>(stab)[i] = reinterpret_cast<const float4*>(S)[i];
    } else {
        u32x4v* sf = reinterpret_cast<u32x4v*>(stab);
        for (int w = threadIdx.x; w < NH * 2 * NH * 64; w += 256) {
            const int ln = w & 63, combo = w >> 6;
            const int h = combo % NH, s2 = (combo / NH) & 1, hin = combo / (2 * NH);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = S[(long long)(hin * 32 + rho(8 * s2 + e, ln >> 5)) * Hp + h * 32 + (ln & 31)];
            u32x4v part[SIGMA_NT];
            split8_parts<SIGMA_NT>(v, part);
#pragma unroll
            for (int p = 0; p < SIGMA_NT; ++p) sf[(combo * SIGMA_NT + p) * 64 + ln] = part[p];
        }
    }
}

// The Sigma / sigma2 table as pre-split bf16 hi / lo MFMA fragments [(hin, s2, h)][hi | lo][lane] for post_frag_kernel's
// three-term product: split ONCE per sweep here instead of by every wave for every tile pair (8 four-byte loads and two
// splits per fragment: 1024 scattered loads per wave at H = 256, what the kernel spent most of its time on).
template <int NH>
__global__ __launch_bounds__(256) void split_table_kernel(const float* __restrict__ S, uint4* __restrict__ Sf,
                                                          const int* __restrict__ stop) {
    constexpr int Hp = NH * 32;
    if (stop && *stop) return;
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= NH * 2 * NH * 64) return;
    const int ln = w & 63, combo = w >> 6;
    const int h = combo % NH, s2 = (combo / NH) & 1, hin = combo / (2 * NH);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = S[(long long)(hin * 32 + rho(8 * s2 + e, ln >> 5)) * Hp + h * 32 + (ln & 31)];
    u32x4v hi, lo;
    split8_bf16(v, hi, lo);
    Sf[(combo * 2 + 0) * 64 + ln] = uint4{hi[0], hi[1], hi[2], hi[3]};
    Sf[(combo * 2 + 1) * 64 + ln] = uint4{lo[0], lo[1], lo[2], lo[3]};
}

// ---- post_frag2: the H >= 128 factor update from a fragment-major product, software-pipelined (bf16 factor modes) ----------
// Same contract and epilogue as post_frag_kernel's bf16 branch.  What changes is how the operands arrive.  post_frag_kernel
// loaded each pair of table fragments right before the MFMAs that consume them -- 2 * NH * NH dependent L2 round trips per
// wave (profiles/r02_pmc_mfma_cfg5.json: 7 % MFMA busy, 75 % of the wave cycles waiting; 215 us at 100k x 256 against ~75 us
// of HBM time).  Here
//   * the table is a LINEAR stream: split_table_kernel<NH, NT> lays the fragments out in exactly the order of use,
//     [iteration = (hin, s2, h)][part][lane], and a D-deep register ring of iterations is refilled in place right after use
//     (SGPR-offset buffer loads + sched_group_barrier, the streaming kernel's recipe, so the loads stay in flight across the
//     loop); the run-ahead past the table is clamped to its last iteration;
//   * the next h tile's 16 product registers per row tile are requested one whole hin step ahead;
//   * NT = 3 parts per operand (bf16 hi + mid + lo carry all 24 significant bits of an fp32 value) and the six products with
//     part indices i + j <= 2: what is dropped is 2^-24 relative, an fp32 rounding -- the EXACT product again, at 12 bf16 MFMAs
//     of 32 cycles per 32 x 32 x 32 block where the exact-f32 MFMA needs 16 of 64.  (NT = 2, three products, is the round-1
//     form: each product term then carries 2^-17, which B = Q inv(K_B) amplifies by cond(K_B).)
// BSIDE = false: the A update (label mask, no delta tiles, no trace).
// The Sigma / sigma2 table as pre-split bf16 MFMA fragments in order of use: [(hin, s2, h)][part < NT][lane] (1 KiB each)
template <int NH, int NT>
__global__ __launch_bounds__(256) void split_table_parts_kernel(const float* __restrict__ S, uint4* __restrict__ Sf,
                                                                const int* __restrict__ stop) {
    constexpr int Hp = NH * 32;
    if (stop && *stop) return;
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= NH * 2 * NH * 64) return;
    const int ln = w & 63, combo = w >> 6;
    const int h = combo % NH, s2 = (combo / NH) & 1, hin = combo / (2 * NH);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = S[(long long)(hin * 32 + rho(8 * s2 + e, ln >> 5)) * Hp + h * 32 + (ln & 31)];
    u32x4v part[NT];
    split8_parts<NT>(v, part);
#pragma unroll
    for (int p = 0; p < NT; ++p) Sf[((long long)combo * NT + p) * 64 + ln] = uint4{part[p][0], part[p][1], part[p][2], part[p][3]};
}

// row tiles per wave on a LONG side (tuning switches, A/B on the GPU): 256 accumulator registers' worth (one wave per SIMD) or half
// of it (two waves per SIMD: one wave's epilogue overlaps the other's MFMA loop)
#ifndef VBMF_POST2_NXT8
#define VBMF_POST2_NXT8 2
#endif
#ifndef VBMF_POST2_NXT4
#define VBMF_POST2_NXT4 4
#endif
template <int NH> struct PostFrag2Cfg { static constexpr int NXT = NH >= 8 ? VBMF_POST2_NXT8 : VBMF_POST2_NXT4; };

// The tail of the H >= 128 factor update, shared by post_frag2_kernel and post_frag3_kernel: the wave's NXT x NH accumulator blocks
// -> mask, operand tiles, delta tiles, fp32 factor, tr(B'YA) share.  `wslot` = this wave's slot in trpart ([4 * workgroups]).
template <int MODE, int NH, int NXT, bool BSIDE, int AHEAD_MIN = 16>
__device__ __forceinline__ void post_frag_tail(f32x16 (&acc)[NXT][NH], const float4* __restrict__ In4, float* __restrict__ Fac,
                                               uint4* __restrict__ Ft, const unsigned char* __restrict__ mask, int hmask_start,
                                               int XT, int xt0, double* __restrict__ trpart, uint4* __restrict__ Fd, int store_fac,
                                               float* tbw, int wslot, int lane) {
    constexpr int Hp = NH * 32;
    // accumulator layout: lane (c = h' in tile, half), register r -> row x0 + rho(r, half)
    // (the store addresses are derived from an opaque copy of the lane id: computed up front, as the optimiser would, the 16
    //  blocks' row pointers live across the main loop and push its operand rings into scratch)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int c = lane_e & 31, half = lane_e >> 5;
    double tr = 0.0;
    const bool want_tr = BSIDE && trpart != nullptr;
    const bool want_old = BSIDE && Fd != nullptr;
    constexpr int NPART = ModeTraits<MODE>::NPART;
    constexpr int NB = NXT * NH;                             // 32 x 32 blocks of this wave, b = i * NH + h
    // ONE block at a time, everything that needs its accumulator tile inside (mask, operand tiles, delta tiles, fp32 store,
    // tr(B'YA) share): a tile that had to survive until a later loop is 16 more live registers per block.  What a block READS
    // (its previous operand tiles for the delta, its product fragments again for the trace: L2-hot) is requested one block
    // ahead, so the round trip overlaps the block before.
    uint4 on[2 * NPART], oc[2 * NPART];
    float4 qn[4], qc4[4];
    auto request = [&](int b) __attribute__((always_inline)) {
        const int i = b / NH, h = b % NH;
        const int xt = xt0 + i < XT ? xt0 + i : XT - 1;
        if (want_old) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pa = 0; pa < NPART; ++pa) on[s * NPART + pa] = Ft[(((long long)(2 * xt + s) * NPART + pa) * NH + h) * 64 + lane_e];
        }
        if (want_tr) {
            const float4* t4 = In4 + (((long long)xt * NH + h) * 64 + lane_e) * 4;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) qn[qd] = t4[qd];
        }
    };
    // (with half the accumulator registers or fewer, two or more waves run per SIMD and overlap each other's round trips: no
    //  read-ahead there, it would cost the 32 registers that keep the wave under 128 VGPRs)
    constexpr bool AHEAD = NXT * NH >= AHEAD_MIN;
    if constexpr (AHEAD) request(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = b / NH, h = b % NH;
        const int xt = xt0 + i;
        if (xt >= XT) break;
        const long long x0 = (long long)xt * 32;
        const int hcol = h * 32 + c;
        if constexpr (!AHEAD) request(b);
#pragma unroll
        for (int u = 0; u < 2 * NPART; ++u) oc[u] = on[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) qc4[u] = qn[u];
        if constexpr (AHEAD) {
            if (b + 1 < NB) request(b + 1);
        }
        // (the tile is taken out of the accumulation registers HERE: common.hpp, acc_read_tile)
        f32x16 a;
        acc_read_tile(acc[i][h], a);
        if constexpr (!BSIDE) {
            if (mask != nullptr && hcol >= hmask_start) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (mask[x0 + rho(r, half)]) a[r] = 0.f;
            }
        }
        // Fd (B side): old - new of this block as delta tiles; the old block is what the operand tiles held (read above, before
        // they are overwritten here).  store_fac = 0 (inside the run loops): no fp32 copy per sweep.
        write_factor_tiles<MODE, NH>(Ft, a, xt, h, lane_e);
        if (want_old) {
            f32x16 oldv;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned hw[4] = {oc[s * NPART].x, oc[s * NPART].y, oc[s * NPART].z, oc[s * NPART].w};
                unsigned lw[4] = {0u, 0u, 0u, 0u};
                if constexpr (NPART == 2) { lw[0] = oc[s * NPART + 1].x; lw[1] = oc[s * NPART + 1].y; lw[2] = oc[s * NPART + 1].z; lw[3] = oc[s * NPART + 1].w; }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = bf2f((unsigned short)(hw[e >> 1] >> (16 * (e & 1))));
                    if constexpr (NPART == 2) f += bf2f((unsigned short)(lw[e >> 1] >> (16 * (e & 1))));
                    oldv[8 * s + e] = f;
                }
            }
            write_delta_tiles<NH>(Fd, oldv, a, xt, h, lane_e);
        }
        if (store_fac) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Fac[(x0 + rho(r, half)) * Hp + hcol] = a[r];
        }
        if (want_tr) {
            float q[16];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) { q[4 * qd] = qc4[qd].x; q[4 * qd + 1] = qc4[qd].y; q[4 * qd + 2] = qc4[qd].z; q[4 * qd + 3] = qc4[qd].w; }
            tr += (double)tile_dot_qb<true>(q, a, tbw, lane_e);
        }
    }
    if (want_tr) store_wave_dot(tr, trpart + wslot, lane_e);
}

// NXT: 32-row tiles per wave (PostFrag2Cfg<NH>::NXT on a long side, 1 on a short one: more workgroups)
template <int MODE, int NH, int NXT, int NT, bool BSIDE>
__global__ __launch_bounds__(256) void post_frag2_kernel(const float4* __restrict__ In4, const uint4* __restrict__ Sf,
                                                         float* __restrict__ Fac, uint4* __restrict__ Ft,
                                                         const unsigned char* __restrict__ mask, int hmask_start, int XT,
                                                         const int* __restrict__ stop, double* __restrict__ trpart,
                                                         uint4* __restrict__ Fd, int store_fac) {
    static_assert(MODE != MODE_F32, "bf16 factor modes only (the fp32 mode keeps post_frag_kernel's exact-f32 MFMA)");
    constexpr int Hp = NH * 32;
    constexpr int D = (NH >= 8 && NXT >= 2) ? 8 : 4;          // table ring depth (iterations); 4 keeps NXT = 1 under 256 registers
    constexpr int NIT = 2 * NH;                               // table iterations (s2, h) per hin
    static_assert(NIT % D == 0, "ring slot = iteration mod D must not depend on hin");
    __shared__ float tbuf[4][32 * TB_LD];
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xt0 = (blockIdx.x * 4 + wib) * NXT;
    if (xt0 >= XT) {
        if (trpart && lane == 0) trpart[blockIdx.x * 4 + wib] = 0.0;
        return;
    }
    // descriptors: the table stream, and each row tile's NH product blocks (4 KiB each, contiguous); tiles past XT (ragged last
    // wave) read tile XT-1 again and are dropped at the store
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)Sf, 0, (unsigned)(NH * NIT * NT) * 1024u, 0x00020000);
    __amdgpu_buffer_rsrc_t qrs[NXT];
#pragma unroll
    for (int i = 0; i < NXT; ++i) {
        const int xt = xt0 + i < XT ? xt0 + i : XT - 1;
        qrs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(In4 + (long long)xt * NH * 256), 0, (unsigned)NH * 4096u, 0x00020000);
    }
    const int tvo = lane * 16, qvo = lane * 64;

    f32x16 acc[NXT][NH];
#pragma unroll
    for (int i = 0; i < NXT; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;

    u32x4v tb[D][NT];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int p = 0; p < NT; ++p) tb[d][p] = __builtin_amdgcn_raw_buffer_load_b128(trs, tvo, (d * NT + p) * 1024, 0);
    u32x4v qc[NXT][4];
#pragma unroll
    for (int i = 0; i < NXT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) qc[i][q] = __builtin_amdgcn_raw_buffer_load_b128(qrs[i], qvo + q * 16, 0, 0);

    for (int hin = 0; hin < NH; ++hin) {
        // (the SGPR offset is not part of the descriptor's bounds check, so run-ahead is clamped, not left to it)
        const int hnext = hin + 1 < NH ? hin + 1 : hin;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4v ap[NXT][NT];
#pragma unroll
            for (int i = 0; i < NXT; ++i) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned w = qc[i][2 * s2 + (e >> 2)][e & 3];
                    v[e] = bitsf(w);
                }
                split8_parts<NT>(v, ap[i]);
            }
            if (s2 == 1) {
                // the product registers are dead once split: the next h tile's 16 per row tile are requested into them now, half a
                // hin step (8 NH MFMA groups) ahead of their use (the last step re-reads its own block: never used)
#pragma unroll
                for (int i = 0; i < NXT; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) qc[i][q] = __builtin_amdgcn_raw_buffer_load_b128(qrs[i], qvo + q * 16, hnext * 4096, 0);
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int it = s2 * NH + h, slot = it % D;
                const int itn = min(hin * NIT + it + D, NH * NIT - 1);       // run-ahead past the table re-reads its last iteration
#pragma unroll
                for (int pa = 0; pa < NT; ++pa)
#pragma unroll
                    for (int pb = 0; pa + pb < NT; ++pb)
#pragma unroll
                        for (int i = 0; i < NXT; ++i)
                            acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[i][pa]),
                                                                               __builtin_bit_cast(bf16x8, tb[slot][pb]), acc[i][h], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < NT; ++p)
                    tb[slot][p] = __builtin_amdgcn_raw_buffer_load_b128(trs, tvo, (itn * NT + p) * 1024, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NXT * (NT * (NT + 1) / 2), 0);
                __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);
            }
        }
    }

    post_frag_tail<MODE, NH, NXT, BSIDE>(acc, In4, Fac, Ft, mask, hmask_start, XT, xt0, trpart, Fd, store_fac, tbuf[wib],
                                         blockIdx.x * 4 + wib, lane);
}

// ---- post_frag3: post_frag2 with the table shared through LDS (two waves per SIMD) ---------------------------------------
// post_frag2 at 100k x 256 (profiles/r03_cfg5_timeline.txt: 140 us) is neither at its matrix-pipe time (768 MFMAs per row tile:
// ~36 us on 1024 SIMDs) nor at its HBM time (~70 us: product in, operand + delta tiles out, previous tiles in): every wave pulls
// the whole 384 KiB table through its own registers -- 3 buffer loads per 12 MFMAs, each ~50 cycles of issue during which the one
// wave of the SIMD feeds no MFMA (the streaming kernel's finding, stream_gemm.hpp) -- and 256 accumulator registers leave one wave
// per SIMD, so the tail (stores, delta tiles, trace) of one wave overlaps nobody's MFMAs.  Here a (hin, s2) STAGE of the table
// (NH * NT KiB) is brought into LDS once per workgroup by LDS-DMA (each wave a quarter), double-buffered one stage ahead behind ONE
// raw barrier per stage; the waves read their B operands with ds_read_b128 (lane-linear, conflict-free).  128 accumulator registers
// per wave (NXT = 1 at NH = 8, 2 at NH = 4) -> two workgroups per CU.  Same products in the same order as post_frag2: bit-identical.
#ifndef VBMF_POST3_AHEAD_MIN
#define VBMF_POST3_AHEAD_MIN 16     // blocks per wave from which the tail requests a block's reads one block ahead.  8 (every long-side launch; 256 VGPRs) measured no faster: 130 vs 122-125 us at 100k x 256, 64.6 vs 64.2 us at 125k x 128
#endif
template <int MODE, int NH, int NXT, int NT, bool BSIDE>
__global__ __launch_bounds__(256, 2) void post_frag3_kernel(const float4* __restrict__ In4, const uint4* __restrict__ Sf,
                                                            float* __restrict__ Fac, uint4* __restrict__ Ft,
                                                            const unsigned char* __restrict__ mask, int hmask_start, int XT,
                                                            const int* __restrict__ stop, double* __restrict__ trpart,
                                                            uint4* __restrict__ Fd, int store_fac) {
    static_assert(MODE != MODE_F32, "bf16 factor modes only");
    constexpr int STAGE = NH * NT;                            // 1 KiB pieces of one (hin, s2) stage: [h][part]
    constexpr int NSTG = 2 * NH;
    constexpr int PPW = STAGE / 4;                            // pieces each wave fetches per stage
    static_assert(STAGE % 4 == 0, "four waves share the fetch");
    static_assert(2 * STAGE * 1024 >= 4 * 32 * TB_LD * 4, "the tail's per-wave transposition buffers alias the ring");
    __shared__ __attribute__((aligned(16))) unsigned char ring[2 * STAGE * 1024];
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xt0r = (blockIdx.x * 4 + wib) * NXT;
    const bool active = xt0r < XT;                           // a wave without tiles still fetches its quarter and meets the barriers
    const int xt0 = active ? xt0r : XT - 1;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)Sf, 0, (unsigned)(NSTG * STAGE) * 1024u, 0x00020000);
    __amdgpu_buffer_rsrc_t qrs[NXT];
#pragma unroll
    for (int i = 0; i < NXT; ++i) {
        const int xt = xt0 + i < XT ? xt0 + i : XT - 1;
        qrs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(In4 + (long long)xt * NH * 256), 0, (unsigned)NH * 4096u, 0x00020000);
    }
    const int tvo = lane * 16, qvo = lane * 64;
    auto dma_stage = [&](int st, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PPW; ++q)
            lds_dma_piece(trs, ring + (slot * STAGE + wib * PPW + q) * 1024, tvo, (st * STAGE + wib * PPW + q) * 1024);
    };

    f32x16 acc[NXT][NH];
#pragma unroll
    for (int i = 0; i < NXT; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;

    dma_stage(0, 0);
    u32x4v qc[NXT][4];
#pragma unroll
    for (int i = 0; i < NXT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) qc[i][q] = __builtin_amdgcn_raw_buffer_load_b128(qrs[i], qvo + q * 16, 0, 0);

    for (int hin = 0; hin < NH; ++hin) {
        const int hnext = hin + 1 < NH ? hin + 1 : hin;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int st = 2 * hin + s2;                      // stage st sits in slot st & 1 = s2
            // my quarter of stage st (and, s2 = 0, this hin's product registers) has landed; after the barrier everybody's has, and
            // every wave is done reading stage st - 1, whose slot the next stage is fetched into now
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st + 1 < NSTG) dma_stage(st + 1, s2 ^ 1);
            u32x4v ap[NXT][NT];
#pragma unroll
            for (int i = 0; i < NXT; ++i) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned w = qc[i][2 * s2 + (e >> 2)][e & 3];
                    v[e] = bitsf(w);
                }
                split8_parts<NT>(v, ap[i]);
            }
            if (s2 == 1) {
#pragma unroll
                for (int i = 0; i < NXT; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) qc[i][q] = __builtin_amdgcn_raw_buffer_load_b128(qrs[i], qvo + q * 16, hnext * 4096, 0);
            }
            const u32x4v* tl = reinterpret_cast<const u32x4v*>(ring + s2 * STAGE * 1024) + lane;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                u32x4v tb[NT];
#pragma unroll
                for (int p = 0; p < NT; ++p) tb[p] = tl[(h * NT + p) * 64];
#pragma unroll
                for (int pa = 0; pa < NT; ++pa)
#pragma unroll
                    for (int pb = 0; pa + pb < NT; ++pb)
#pragma unroll
                        for (int i = 0; i < NXT; ++i)
                            acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[i][pa]),
                                                                               __builtin_bit_cast(bf16x8, tb[pb]), acc[i][h], 0, 0, 0);
            }
        }
    }
    // the ring is dead once every wave has read the last stage: its space becomes the tail's transposition buffers
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) {
        if (trpart && lane == 0) trpart[blockIdx.x * 4 + wib] = 0.0;
        return;
    }
    post_frag_tail<MODE, NH, NXT, BSIDE, VBMF_POST3_AHEAD_MIN>(acc, In4, Fac, Ft, mask, hmask_start, XT, xt0, trpart, Fd, store_fac,
                                         reinterpret_cast<float*>(ring) + wib * 32 * TB_LD, blockIdx.x * 4 + wib, lane);
}

// ---- fused post + Gram (NH <= 2), second form: the tile body of the register epilogue as its own kernel ------------------
// For products that did go through HBM: Y'B always (split-K slabs, summed by slab_sum / all-reduced over the ranks before), Y*A
// when that pass is split (short row shards, narrow problems).  `In` is ONE fragment-major product (stream_gemm.hpp, frag_out):
// the wave's tile arrives as its accumulator registers in four 16-byte loads per column tile and runs through
// post_gram_tile_regs -- the exact-f32 product and, in the bf16 factor modes, the bf16 Gram / delta-Gram of the operand fragments
// (every product exact): 5 600 MFMA cycles per tile where round 1's all-f32 post + Gram kernel spent 10 240.  The previous factor's rows
// (BSIDE: for the delta-Gram) come from its operand tiles (1 KiB wave loads), the table from LDS; the stop flag is requested
// first and looked at before the first store, so its round trip overlaps the loads.  store_fac = 0 (inside the run loops):
// the fp32 row-major factor is not written (rebuilt from the tiles once at the end, like the register epilogue's).
//   slabs: [gridDim.x][2 (Gram, delta)][NPAIR][16 regs][64 lanes] fp32;  trpart: [gridDim.x][4] per-wave shares of tr(B'YA)
template <int MODE, int NH, bool BSIDE>
__global__ __launch_bounds__(256) void post_gram2_kernel(const float* __restrict__ In, const float* __restrict__ S,
                                                         float* __restrict__ Fac, uint4* __restrict__ Ft,
                                                         const unsigned char* __restrict__ mask, int hmask_start, int XT,
                                                         float* __restrict__ slabs, const int* __restrict__ stop,
                                                         double* __restrict__ trpart, int store_fac) {
    static_assert(NH <= 2, "fused Gram keeps NH(NH+1)/2 pair tiles per matrix in registers");
    constexpr int Hp = NH * 32;
    constexpr int NPAIR = NH * (NH + 1) / 2;
    __shared__ float fold[2 * NPAIR * 16 * 64];
    __shared__ float tbuf[4][32 * TB_LD];
    __shared__ __attribute__((aligned(16))) float stab[sigma_lds_floats<MODE, NH>()];
    const int stopped = stop ? __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    load_sigma_table<MODE, NH>(stab, S);
    f32x16 G[NPAIR], D[NPAIR];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) { G[p][r] = 0.f; D[p][r] = 0.f; }
    double trd = 0.0;
    bool first = true;
    for (int xt0 = blockIdx.x * 4; xt0 < XT; xt0 += gridDim.x * 4) {        // workgroup-uniform trip count
        const int xt = xt0 + wib;
        const bool active = xt < XT;
        f32x16 q[NH], pv[NH];
        if (active) {
#pragma unroll
            for (int hin = 0; hin < NH; ++hin) {
                const float4* ip = reinterpret_cast<const float4*>(In) + (((long long)xt * NH + hin) * 64 + lane) * 4;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 v = ip[qd];
                    q[hin][4 * qd] = v.x; q[hin][4 * qd + 1] = v.y; q[hin][4 * qd + 2] = v.z; q[hin][4 * qd + 3] = v.w;
                }
            }
            if constexpr (BSIDE) {
#pragma unroll
                for (int h = 0; h < NH; ++h) read_factor_tiles<MODE, NH>(Ft, pv[h], xt, h, lane);
            }
        }
        if (first) {
            __syncthreads();                                                  // the table is in LDS
            if (stopped) return;                                              // uniform over the grid; nothing stored yet
            first = false;
        }
        if (active)
            post_gram_tile_regs<MODE, NH, BSIDE>(q, stab, xt, Fac, nullptr, Ft, lane, G, D, pv, store_fac, tbuf[wib], trd, mask, hmask_start);
    }
    // fold the four waves' partials (fixed order => deterministic), then one coalesced slab store
    for (int wv = 0; wv < 4; ++wv) {
        if (wib == wv) {
#pragma unroll
            for (int p = 0; p < NPAIR; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ig = (p * 16 + r) * 64 + lane, id = ((NPAIR + p) * 16 + r) * 64 + lane;
                    const float g = G[p][r], d = D[p][r];
                    if (wv == 0) { fold[ig] = g; fold[id] = d; }
                    else { fold[ig] += g; fold[id] += d; }
                }
        }
        __syncthreads();
    }
    float* o = slabs + (long long)blockIdx.x * (2 * NPAIR * 1024);
    for (int i = threadIdx.x; i < 2 * NPAIR * 1024; i += 256) o[i] = fold[i];
    if (BSIDE && trpart != nullptr) store_wave_dot(trd, trpart + blockIdx.x * 4 + wib, lane);
}

// fp64 reduction of the post_gram slabs into dense Hp x Hp matrices (both triangles).
//   one thread per (matrix, pair, reg, lane) x 4 slab groups; LDS fold of the groups.
// fixed-order sum of the per-wave shares of tr(B'YA) (one block; deterministic: fixed strides, fixed tree)
__device__ __forceinline__ void fold_wave_dots(const double* __restrict__ trpart, int ntr, double* __restrict__ outTr) {
    __shared__ double trw[16];
    double v = 0.0;
    for (int k = threadIdx.x; k < ntr; k += blockDim.x) v += trpart[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) trw[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += trw[w];
        *outTr = t;
    }
}

// This rank's device error state as one more number of the packed Gram message of a row-sharded run (summed over the ranks by
// the same all-reduce, read by every rank's loop test: ctrl_kernels.hpp, remote_error_code): 1 per numeric error (a pivot),
// 1000 per in-launch hand-off timeout, 0 otherwise.  `ints` = the context's flag block (I_ERR at [2]).
__device__ __forceinline__ void publish_err_flag(const int* __restrict__ ints, double* __restrict__ outErr) {
    const int e = __hip_atomic_load(ints + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *outErr = e == 0 ? 0.0 : (e == 2 ? 1000.0 : 1.0);
}

template <int NH>
__global__ __launch_bounds__(1024) void pair_slab_reduce_kernel(const float* __restrict__ slabs, int nslab,
                                                                double* __restrict__ outG, double* __restrict__ outD,
                                                                const int* __restrict__ stop,
                                                                const double* __restrict__ trpart = nullptr, int ntr = 0,
                                                                double* __restrict__ outTr = nullptr,
                                                                double* __restrict__ outErr = nullptr) {
    constexpr int Hp = NH * 32;
    constexpr int NPAIR = NH * (NH + 1) / 2;
    constexpr int NOUT = 2 * NPAIR * 1024;
    constexpr int NG = 16;
    __shared__ double part[NG][64];
    if (stop && *stop) return;
    if (outErr != nullptr && stop != nullptr && blockIdx.x == 0 && threadIdx.x == 0) publish_err_flag(stop, outErr);   // (stop = ints + I_STOP = ints)
    if (outTr != nullptr && blockIdx.x == gridDim.x - 1) fold_wave_dots(trpart, ntr, outTr);
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + j;                       // (mat, pair, r, lane) flat, grid = NOUT/64
    if ((idx / (NPAIR * 1024) ? outD : outG) == nullptr) return;   // (block-uniform: 1024 % 64 == 0) a matrix nobody asked for
    double s = 0.0;
#pragma unroll 4
    for (int k = g; k < nslab; k += NG) s += (double)slabs[(long long)k * NOUT + idx];
    part[g][j] = s;
    __syncthreads();
    if (g == 0) {
        s = 0.0;
#pragma unroll
        for (int q = 0; q < NG; ++q) s += part[q][j];
        const int mat = idx / (NPAIR * 1024), rem = idx % (NPAIR * 1024);
        const int p = rem / 1024, r = (rem % 1024) / 64, lane = rem % 64;
        int h1 = 0, h2 = 0, q = 0;
        for (int a = 0; a < NH; ++a) for (int b = a; b < NH; ++b, ++q) if (q == p) { h1 = a; h2 = b; }
        const int row = h1 * 32 + rho(r, lane >> 5), col = h2 * 32 + (lane & 31);
        double* out = mat ? outD : outG;
        if (out != nullptr) {
            out[(long long)row * Hp + col] = s;
            if (h1 != h2) out[(long long)col * Hp + row] = s;
        }
    }
}

// rowscale != nullptr: the value tiled (and, with writeback, stored back) is rowscale[x] * Fac[x][h]
template <int MODE, int NH>
__global__ __launch_bounds__(256) void retile_kernel(float* __restrict__ Fac, uint4* __restrict__ Ft, int XT,
                                                     const float* __restrict__ rowscale, int writeback,
                                                     const int* __restrict__ stop) {
    constexpr int Hp = NH * 32;
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int xt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (xt >= XT) return;
    const int c = lane & 31, half = lane >> 5;
    const long long x0 = (long long)xt * 32;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = Fac[(x0 + rho(r, half)) * Hp + h * 32 + c];
            if (rowscale) v[r] *= rowscale[x0 + rho(r, half)];
        }
        write_factor_tiles<MODE, NH>(Ft, v, xt, h, lane);
        if (writeback && (MODE != MODE_F32 || rowscale)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Fac[(x0 + rho(r, half)) * Hp + h * 32 + c] = v[r];
        }
    }
}

// One wave per (chunk of 32-row tiles, ordered tile pair (h1,h2)).  slabs: [nchunk][2][Hp][Hp] fp32.
template <int NH>
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ Cur, const float* __restrict__ Prev,
                                                   float* __restrict__ slabs, int XT, int tiles_per_chunk,
                                                   int nchunk, const int* __restrict__ stop) {
    constexpr int Hp = NH * 32;
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nchunk * NH * NH) return;
    const int chunk = w / (NH * NH), pair = w % (NH * NH);
    const int h1 = pair / NH, h2 = pair % NH;
    const int c = lane & 31, half = lane >> 5;
    f32x16 g, d;
#pragma unroll
    for (int r = 0; r < 16; ++r) { g[r] = 0.f; d[r] = 0.f; }
    const int t0 = chunk * tiles_per_chunk;
    const int t1 = min(XT, t0 + tiles_per_chunk);
    for (int xt = t0; xt < t1; ++xt) {
        const long long x0 = (long long)xt * 32;
        float a1[16], a2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = (x0 + rho(r, half)) * Hp;
            a1[r] = Cur[row + h1 * 32 + c];
            a2[r] = Cur[row + h2 * 32 + c];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) g = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], a2[r], g, 0, 0, 0);
        if (Prev != nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = (x0 + rho(r, half)) * Hp;
                a1[r] = Prev[row + h1 * 32 + c] - a1[r];
                a2[r] = Prev[row + h2 * 32 + c] - a2[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], a2[r], d, 0, 0, 0);
        }
    }
    float* o = slabs + (long long)chunk * 2 * Hp * Hp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long idx = (long long)(h1 * 32 + rho(r, half)) * Hp + h2 * 32 + c;
        o[idx] = g[r];
        o[(long long)Hp * Hp + idx] = d[r];
    }
}

// ---- Grams from the factor's OPERAND TILES with bf16 MFMAs (H >= 128, bf16 factor modes) ------------------------
// The fp32 factor is exactly hi + lo (write_factor_tiles), and a tile's fragments are laid out "lane = column, 8
// consecutive k per lane" -- which is BOTH the A- and the B-operand layout of v_mfma_f32_32x32x16_bf16 with the row
// index as k.  So  F'F = hi'hi + hi'lo + lo'hi + lo'lo  is four bf16 MFMAs per tile pair and 16-row step straight from
// the tiles (1 KiB wave loads), every product exact, fp32 accumulation as before: the same Gram at 1/16 of the MFMA time
// of the exact-f32 path and without its 4-byte row gathers (measured at 100k x 256: 286 -> see DESIGN.md).
// The delta-Gram needs d = old - new element-wise BEFORE squaring (src/util.jl:27-29 near convergence): old from the fp32
// factor of the previous sweep, new from the tiles, d re-split into hi + lo (2^-17 relative on d).
// One workgroup per chunk of tiles; wave w owns the upper-triangular tile pairs p = w (mod 4); slab format of gram_kernel.
// WHAT = 0: the Gram only; 1: the delta-Gram only (two launches at H = 256: 9 pair tiles per wave and matrix, and both
// matrices' accumulators together with the fragments would spill).
// out_which (WHAT = 0): the slab half the result goes to -- 0 for a factor's own Gram, 1 when `Ft` is a DELTA tile buffer
// (write_delta_tiles) and the result is the delta-Gram.
template <int NH, int NPART, int WHAT>
__global__ __launch_bounds__(256) void gram_tiles_kernel(const uint4* __restrict__ Ft, const float* __restrict__ Prev,
                                                         float* __restrict__ slabs, int XT, int tiles_per_chunk,
                                                         const int* __restrict__ stop, int out_which = WHAT) {
    constexpr int Hp = NH * 32;
    constexpr int NPAIR = NH * (NH + 1) / 2;
    constexpr int PW = (NPAIR + 3) / 4;
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int chunk = blockIdx.x;
    const u32x4v* F = reinterpret_cast<const u32x4v*>(Ft);
    f32x16 G[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) G[q][r] = 0.f;
    const int t0 = chunk * tiles_per_chunk;
    const int t1 = min(XT, t0 + tiles_per_chunk);
    for (int xt = t0; xt < t1; ++xt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4v f[NH][NPART];
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int pa = 0; pa < NPART; ++pa)
                    f[h][pa] = F[(((long long)(2 * xt + s) * NPART + pa) * NH + h) * 64 + lane];
            u32x4v dh[NH][2];
            if constexpr (WHAT == 1) {
                // each wave forms the difference fragments of NH/4 column tiles (the 4-byte row gathers of the old
                // factor are the expensive part) and the workgroup shares them through LDS
                __shared__ u32x4v dsh[NH][2][64];
                __syncthreads();                                 // the previous step's fragments have been consumed
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    if ((h & 3) != wib) continue;                // wave-uniform
                    unsigned short hi[8], lo[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned hw = f[h][0][e >> 1];
                        float v = bf2f((unsigned short)(hw >> (16 * (e & 1))));
                        if constexpr (NPART == 2) v += bf2f((unsigned short)(f[h][1][e >> 1] >> (16 * (e & 1))));
                        const float old = Prev[((long long)xt * 32 + rho(8 * s + e, half)) * Hp + h * 32 + c];
                        const float d = old - v;
                        hi[e] = f2bf(d);
                        lo[e] = f2bf(d - bf2f(hi[e]));
                    }
                    dsh[h][0][lane] = u32x4v{hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16),
                                             hi[4] | ((unsigned)hi[5] << 16), hi[6] | ((unsigned)hi[7] << 16)};
                    dsh[h][1][lane] = u32x4v{lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16),
                                             lo[4] | ((unsigned)lo[5] << 16), lo[6] | ((unsigned)lo[7] << 16)};
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < NH; ++h) { dh[h][0] = dsh[h][0][lane]; dh[h][1] = dsh[h][1][lane]; }
            }
            int p = 0, q = 0;
#pragma unroll
            for (int h1 = 0; h1 < NH; ++h1)
#pragma unroll
                for (int h2 = h1; h2 < NH; ++h2, ++p) {
                    if ((p & 3) != wib) continue;            // wave-uniform; p, q are compile-time per unrolled iteration
                    const int qq = p >> 2;
                    if constexpr (WHAT == 0) {
#pragma unroll
                        for (int pa = 0; pa < NPART; ++pa)
#pragma unroll
                            for (int pb = 0; pb < NPART; ++pb)
                                G[qq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[h1][pa]),
                                                                                __builtin_bit_cast(bf16x8, f[h2][pb]), G[qq], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                            for (int pb = 0; pb < 2; ++pb)
                                G[qq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dh[h1][pa]),
                                                                                __builtin_bit_cast(bf16x8, dh[h2][pb]), G[qq], 0, 0, 0);
                    }
                    (void)q;
                }
        }
    }
    // accumulator (lane (half, c), register r) = element (row rho(r, half) of tile h1, column c of tile h2); both triangles
    float* o = slabs + ((long long)chunk * 2 + out_which) * Hp * Hp;
    int p = 0;
#pragma unroll
    for (int h1 = 0; h1 < NH; ++h1)
#pragma unroll
        for (int h2 = h1; h2 < NH; ++h2, ++p) {
            if ((p & 3) != wib) continue;
            const int qq = p >> 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long i = h1 * 32 + rho(r, half), j = h2 * 32 + c;
                o[i * Hp + j] = G[qq][r];
                if (h1 != h2) o[j * Hp + i] = G[qq][r];
            }
        }
}

// gram_tiles3: the tile Gram of the H >= 128 run loops (bf16 factor modes), Gram of `Ft` and -- in the same launch -- of the
// delta tiles `Fd`.  What it fixes in gram_tiles_kernel<., ., 0>, in the order it was found (profiles/r02_pmc_mfma_cfg5.json: 13 %
// MFMA busy; 66 + 62 us for the 2 x 102 MB of tiles at 100k x 256, 20 us of HBM time each):
//   * one exposed HBM round trip per 16-row step (load, wait, multiply, next load)      -> a DR-deep register ring of steps, refilled
//     in place right after use (buffer loads; the step offset travels in the VGPR offset, so steps past the chunk's end are out
//     of the descriptor's range and arrive as zeros: no remainder loop)                                             128 -> 102 us
//   * 36 wave-uniform branches per step (the wave's pair set as a run-time test)         -> the pair set is a template parameter
//   * every wave loads all NPART * NH fragments of a step itself (its pairs span every column tile): 4 x 16 KiB per step and CU
//     through a 32 KiB L1 with several steps in flight -- the waves evict each other's lines                         -> each wave
//     fetches a QUARTER of the step, publishes it into one of two LDS stages one step ahead, and all four read the whole step
//     from LDS (ds_read_b128, lane-linear: conflict-free) after ONE raw s_barrier per step.  Stage (s + 1) & 1 is written during
//     step s: it last held step s - 1, which every wave finished reading before the barrier that ended step s - 1.   102 -> 92 us
//   * a dense [Hp][Hp] slab with both triangles per chunk, the mirrored half as 4-byte stores 1 KiB apart, 196 chunks on 256 CUs
//     -> pair slabs (the accumulator tiles as they stand, upper pairs only, coalesced; pair_slab_reduce_kernel expands them) and
//     one round of ~250 chunks                                                                                       92 -> 68 us
// Same MFMA order per pair as gram_tiles_kernel.
template <int NH, int NPART, int W>
__device__ __forceinline__ void gram_tiles_lds_body(const uint4* __restrict__ Ft, float* __restrict__ o, int t0, int t1, int lane,
                                                    u32x4v* fl /* LDS: [2][NPART * NH][64] */) {
    constexpr int Hp = NH * 32;
    constexpr int NPAIR = NH * (NH + 1) / 2;
    constexpr int PW = (NPAIR + 3) / 4;
    constexpr int NF = NPART * NH, NFW = NF / 4;
    static_assert(NF % 4 == 0, "four waves share the fetch");
    constexpr int DR = 4;                                        // even: the LDS stage of step s0 + d is d & 1
    constexpr unsigned STEP_BYTES = (unsigned)NF * 1024u;
    const int nsteps = 2 * (t1 - t0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Ft + (long long)t0 * 2 * NF * 64), 0,
                                                                        (unsigned)nsteps * STEP_BYTES, 0x00020000);
    f32x16 G[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) G[q][r] = 0.f;
    u32x4v gl[DR][NFW];
#pragma unroll
    for (int d = 0; d < DR; ++d) {
        const int vo = lane * 16 + d * (int)STEP_BYTES;          // (steps past the chunk: out of the descriptor's range -> zeros)
#pragma unroll
        for (int j = 0; j < NFW; ++j) gl[d][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (W * NFW + j) * 1024, 0);
    }
    // the previous job's last reads of the stages are over (all waves), then step 0 -> stage 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < NFW; ++j) fl[(W * NFW + j) * 64 + lane] = gl[0][j];
    {
        const int vo = lane * 16 + DR * (int)STEP_BYTES;
#pragma unroll
        for (int j = 0; j < NFW; ++j) gl[0][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (W * NFW + j) * 1024, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int s0 = 0; s0 < nsteps; s0 += DR) {
#pragma unroll
        for (int d = 0; d < DR; ++d) {
            const int cur = d & 1, nxt = cur ^ 1, gd = (d + 1) % DR;
            // publish this wave's quarter of step s0 + d + 1 (fetched DR steps ago), refetch the slot for step s0 + d + 1 + DR
#pragma unroll
            for (int j = 0; j < NFW; ++j) fl[(nxt * NF + W * NFW + j) * 64 + lane] = gl[gd][j];
            const int vo = lane * 16 + (s0 + d + 1 + DR) * (int)STEP_BYTES;
#pragma unroll
            for (int j = 0; j < NFW; ++j) gl[gd][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (W * NFW + j) * 1024, 0);
            // the whole step from LDS, then this wave's pairs
            u32x4v f[NH][NPART];
#pragma unroll
            for (int pa = 0; pa < NPART; ++pa)
#pragma unroll
                for (int h = 0; h < NH; ++h) f[h][pa] = fl[(cur * NF + pa * NH + h) * 64 + lane];
            int p = 0;
#pragma unroll
            for (int h1 = 0; h1 < NH; ++h1)
#pragma unroll
                for (int h2 = h1; h2 < NH; ++h2, ++p) {
                    if ((p & 3) != W) continue;              // compile-time after unrolling
                    const int qq = p >> 2;
#pragma unroll
                    for (int pa = 0; pa < NPART; ++pa)
#pragma unroll
                        for (int pb = 0; pb < NPART; ++pb)
                            G[qq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[h1][pa]),
                                                                            __builtin_bit_cast(bf16x8, f[h2][pb]), G[qq], 0, 0, 0);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    // pair-slab format (pair_slab_reduce_kernel): [pair][16 registers][64 lanes], the accumulator tile as it stands -- one
    // coalesced 256-byte row per register, upper-triangular pairs only (the dense [Hp][Hp] slab with both triangles of
    // gram_tiles_kernel is 1.8x the bytes and its mirrored half goes out as 4-byte stores 1 KiB apart)
    int lane_e = lane;                                       // (store addresses from an opaque copy: not hoisted over the loop)
    asm volatile("" : "+v"(lane_e));
    int p = 0;
#pragma unroll
    for (int h1 = 0; h1 < NH; ++h1)
#pragma unroll
        for (int h2 = h1; h2 < NH; ++h2, ++p) {
            if ((p & 3) != W) continue;
            const int qq = p >> 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(p * 16 + r) * 64 + lane_e] = G[qq][r];
        }
}
template <int NH, int NPART, int W>
__device__ __forceinline__ void gram_tiles_lds_job(const uint4* __restrict__ Ft, const uint4* __restrict__ Fd, float* __restrict__ o,
                                                   int t0, int t1, int lane, u32x4v* fl) {
    gram_tiles_lds_body<NH, NPART, W>(Ft, o, t0, t1, lane, fl);
    if (Fd != nullptr) gram_tiles_lds_body<NH, 2, W>(Fd, o + (NH * (NH + 1) / 2) * 1024, t0, t1, lane, fl);   // (workgroup-uniform)
}
template <int NH, int NPART>
__global__ __launch_bounds__(256) void gram_tiles3_kernel(const uint4* __restrict__ Ft, const uint4* __restrict__ Fd,
                                                          float* __restrict__ slabs, int XT, int tiles_per_chunk,
                                                          const int* __restrict__ stop) {
    constexpr int NPAIR = NH * (NH + 1) / 2;
    __shared__ __attribute__((aligned(16))) u32x4v fl[2 * 2 * NH * 64];      // two stages of up to 2 NH fragments
    if (stop && *stop) return;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = blockIdx.x;
    const int t0 = chunk * tiles_per_chunk;
    const int t1 = min(XT, t0 + tiles_per_chunk);
    float* o = slabs + (long long)chunk * 2 * NPAIR * 1024;
    if (wib == 0) gram_tiles_lds_job<NH, NPART, 0>(Ft, Fd, o, t0, t1, lane, fl);
    else if (wib == 1) gram_tiles_lds_job<NH, NPART, 1>(Ft, Fd, o, t0, t1, lane, fl);
    else if (wib == 2) gram_tiles_lds_job<NH, NPART, 2>(Ft, Fd, o, t0, t1, lane, fl);
    else gram_tiles_lds_job<NH, NPART, 3>(Ft, Fd, o, t0, t1, lane, fl);
}

// out[which][i] = sum_chunk slabs[chunk][which][i]  (fp64).  which in {0: Gram, 1: delta-Gram}.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ slabs, int nchunk, int n,
                                                          double* __restrict__ outG, double* __restrict__ outD,
                                                          const int* __restrict__ stop,
                                                          const double* __restrict__ trpart = nullptr, int ntr = 0,
                                                          double* __restrict__ outTr = nullptr,
                                                          double* __restrict__ outErr = nullptr) {
    // 32 consecutive elements per workgroup x 8 chunk groups: thread (e, g) sums chunks g, g + 8, ... (four loads in flight
    // at a time), the groups are then combined in fixed order through LDS.  (One thread walking all the chunks was a chain
    // of dependent-latency loads: 67 us for 245 chunks at 125k x 128.)
    __shared__ double part[8][32];
    if (stop && *stop) return;
    if (outErr != nullptr && stop != nullptr && blockIdx.x == 0 && threadIdx.x == 0) publish_err_flag(stop, outErr);
    if (outTr != nullptr && blockIdx.x == gridDim.x - 1) fold_wave_dots(trpart, ntr, outTr);
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;                         // grid = ceil(2n / 32)
    const bool ok = i < 2 * n;
    const int which = ok ? i / n : 0, j = ok ? i % n : 0;
    double* out = which ? outD : outG;
    double s = 0.0;
    if (ok && out != nullptr) {
        const float* base = slabs + (long long)which * n + j;
        int ch = g;
        for (; ch + 24 < nchunk; ch += 32) {
            const float a = base[(long long)ch * 2 * n], b = base[(long long)(ch + 8) * 2 * n];
            const float c = base[(long long)(ch + 16) * 2 * n], d = base[(long long)(ch + 24) * 2 * n];
            s += (double)a; s += (double)b; s += (double)c; s += (double)d;
        }
        for (; ch < nchunk; ch += 8) s += (double)base[(long long)ch * 2 * n];
    }
    part[g][e] = s;
    __syncthreads();
    if (g == 0 && ok && out != nullptr) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][e];
        out[j] = t;
    }
}

// out[i] = sum_s slabs[s][i]   (fp32, fixed order => deterministic).  16-byte loads, whole-chip grid:
// the split-K partials of the Y'B pass (nsplit x 2.5 MB at 10k x 64) are folded here at HBM/L2 rate
// instead of inside the 80-block post kernel.  n must be a multiple of 4 (it is: Hp * Xp).
// out may be slab 0 itself (element-wise: every thread reads its element of all slabs, then writes it).
// An optional stop-gated side job rides along (saves a launch per sweep): copy cp_n doubles cp_src -> cp_dst and one
// more double sc_src -> sc_dst -- the commit of the speculative SigmaA (ctrl_kernels.hpp, ctrl_chain).
struct SideCopy { const double* cp_src; double* cp_dst; long long cp_n; const double* sc_src; double* sc_dst; };
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* slabs, int nslab,
                                                       long long slabStride, float* out, long long n,
                                                       const int* __restrict__ stop, SideCopy side) {
    if (stop && *stop) return;
    const int nb = gridDim.x < 8 ? (int)gridDim.x : 8;
    if (side.cp_n > 0 && (int)blockIdx.x < nb) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < side.cp_n; i += (long long)nb * blockDim.x)
            side.cp_dst[i] = side.cp_src[i];
        if (blockIdx.x == 0 && threadIdx.x == 0 && side.sc_src) *side.sc_dst = *side.sc_src;
    }
    const long long n4 = n >> 2;
    const float4* in4 = reinterpret_cast<const float4*>(slabs);
    float4* out4 = reinterpret_cast<float4*>(out);
    const long long stride4 = slabStride >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        float4 s = in4[i];
        int k = 1;
        // eight slabs' loads in flight at a time, added in slab order (a plain loop issued them one dependent round trip
        // after the other: 17.5 us for 64 slabs of 128 KB at 10k x 1k)
        for (; k + 7 < nslab; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in4[(long long)(k + u) * stride4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nslab; ++k) {
            const float4 v = in4[(long long)k * stride4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        out4[i] = s;
    }
}

// part[block] = this workgroup's share of  sum_{x<X, h<Hp} (sum_s In[s][h][x]) * Fac[x][h]   (fp64; the caller folds the
// shares in fixed order with sum_partials_kernel: no atomics, so tr(Y'BA') is the same number on every run and rank)
// index of element (h, x) of a fragment-major product (stream_gemm.hpp, frag_out)
__device__ __forceinline__ long long frag_index(int h, long long x, int NH) {
    const int hr = h & 31, half = (hr >> 2) & 1, r = (hr & 3) + 4 * (hr >> 3);          // rho(r, half) == hr
    return ((((x >> 5) * NH + (h >> 5)) * 64 + half * 32 + (int)(x & 31)) << 4) + r;
}
// frag_nh > 0: In is fragment-major with frag_nh = Hp / 32 column tiles
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ In, long long ldIn, int nslab,
                                                  long long slabStride, const float* __restrict__ Fac, int Hp,
                                                  long long X, double* __restrict__ part, int frag_nh) {
    double acc = 0.0;
    for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < X;
         x += (long long)gridDim.x * blockDim.x) {
        for (int h = 0; h < Hp; ++h) {
            float a = 0.f;
            const long long e = frag_nh > 0 ? frag_index(h, x, frag_nh) : (long long)h * ldIn + x;
            for (int s = 0; s < nslab; ++s) a += In[(long long)s * slabStride + e];
            acc += (double)a * (double)Fac[x * Hp + h];
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ double wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// dst[i] = src[i] unless the sweep loop has stopped (keeps the state frozen after `stop`)
__global__ void gated_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int n,
                                  const int* __restrict__ stop) {
    if (stop && *stop) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// fp64 column-major host layout <-> fp32 row-major padded device layout
__global__ void pack_factor_kernel(const double* __restrict__ src, long long ld, long long X, int H, int Hp,
                                   long long Xp, float* __restrict__ dst) {
    const long long total = Xp * Hp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long x = i / Hp; const int h = (int)(i % Hp);
        dst[i] = (x < X && h < H) ? (float)src[x + (long long)h * ld] : 0.f;
    }
}
__global__ void unpack_factor_kernel(const float* __restrict__ src, int Hp, long long X, int H,
                                     double* __restrict__ dst, long long ld) {
    const long long total = X * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long x = i % X; const int h = (int)(i / X);
        dst[x + (long long)h * ld] = (double)src[x * Hp + h];
    }
}

// YHat = BHat * AHat'  (src/vbmf.jl:120-122), fp32 factors, fp64 column-major out.  On demand only.
__global__ __launch_bounds__(256) void yhat_kernel(const float* __restrict__ B, const float* __restrict__ A, int Hp,
                                                   int H, long long L, long long M, double* __restrict__ out,
                                                   long long ld) {
    const long long total = L * M;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long l = i % L, m = i / L;
        double s = 0.0;
        for (int h = 0; h < H; ++h) s += (double)B[l * Hp + h] * (double)A[m * Hp + h];
        out[l + m * ld] = s;
    }
}

}  // namespace vbmf
