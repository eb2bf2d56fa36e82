// stream_gemm_v3.hpp -- EXPERIMENTS of round 3 for the H >= 128 streaming contraction (not built into the library until one wins).
//
//   stream_pair_kernel   the per-wave kernel with the wave tile re-shaped: NXW x tiles  x  NHW (< NH) h tiles, the NH/NHW waves
//                        that share an x group take different h slices.  At H = 256: 4 x 4 tiles instead of 2 x 8, i.e. 12 instead
//                        of 18 one-KiB loads per 32 MFMAs through the CU's L1 port (no LDS, no barriers); the x group's Y tiles are
//                        then fetched by two waves of the same CU (L1/L2 hit for the second).
//
// Same operands, tilings and fragment-major result layout as stream_gemm_kernel (bf16x2 mode only).
#pragma once
#include "common.hpp"

namespace vbmf {

template <int NH, int NHW, int NXW_, int DY, int DF, int YAUX>
__global__ __launch_bounds__(256) void stream_pair_kernel(const uint4* __restrict__ Yt,   // [XT][KS][64]
                                                          const uint4* __restrict__ Ft,   // [KS][2][NH][64]
                                                          float* __restrict__ Out,        // [nsplit][XT][NH][64][16] fragment-major
                                                          int XG, int KS, int steps_per_split, int nsplit, long long ldOut) {
    constexpr int NPART = 2, NF = NPART * NH, NFW = NPART * NHW, HS = NH / NHW, GPW = 4 / HS;
    static_assert(NH % NHW == 0 && 4 % HS == 0, "h slices per x group must divide the workgroup's four waves");
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bps = (XG + GPW - 1) / GPW;
    const int split = blockIdx.x / bps, xb = blockIdx.x % bps;
    const int xg = xb * GPW + wib / HS, hs = wib % HS;
    if (xg >= XG || split >= nsplit) return;

    const long long ks0 = (long long)split * steps_per_split;
    const unsigned ybytes = (unsigned)steps_per_split * 1024u;
    const unsigned fbytes = (unsigned)steps_per_split * (NF * 1024u);
    __amdgpu_buffer_rsrc_t yr[NXW_];
#pragma unroll
    for (int i = 0; i < NXW_; ++i)
        yr[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(Yt + (((long long)(xg * NXW_ + i)) * KS + ks0) * 64), 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Ft + ks0 * (NF * 64) + hs * (NHW * 64)), 0, fbytes, 0x00020000);
    const int voff = lane * 16;

    f32x16 acc[NXW_][NHW];
#pragma unroll
    for (int i = 0; i < NXW_; ++i)
#pragma unroll
        for (int h = 0; h < NHW; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;
    u32x4v yb[DY][NXW_];
    u32x4v fb[DF][NFW];
#pragma unroll
    for (int d = 0; d < DY; ++d)
#pragma unroll
        for (int i = 0; i < NXW_; ++i) yb[d][i] = u32x4v{0u, 0u, 0u, 0u};
#pragma unroll
    for (int d = 0; d < DF; ++d)
#pragma unroll
        for (int j = 0; j < NFW; ++j) fb[d][j] = u32x4v{0u, 0u, 0u, 0u};

    for (int s = -DY; s < steps_per_split; s += DY) {
#pragma unroll
        for (int d = 0; d < DY; ++d) {
            const int fd = d % DF;
#pragma unroll
            for (int j = 0; j < NFW; ++j) {
                const bf16x8 fa = __builtin_bit_cast(bf16x8, fb[fd][j]);
#pragma unroll
                for (int i = 0; i < NXW_; ++i)
                    acc[i][j % NHW] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, yb[d][i]), acc[i][j % NHW], 0, 0, 0);
                fb[fd][j] = __builtin_amdgcn_raw_buffer_load_b128(fr, voff, ((s + d + DF) * NF + (j / NHW) * NH + (j % NHW)) * 1024, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NXW_, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < NXW_; ++i)
                yb[d][i] = __builtin_amdgcn_raw_buffer_load_b128(yr[i], voff, (s + DY + d) * 1024, YAUX);
            __builtin_amdgcn_sched_group_barrier(0x020, NXW_, 0);
        }
    }
    float4* o4 = reinterpret_cast<float4*>(Out + (long long)split * (NH * 32) * ldOut);
#pragma unroll
    for (int i = 0; i < NXW_; ++i)
#pragma unroll
        for (int h = 0; h < NHW; ++h) {
            float4* t = o4 + (((long long)(xg * NXW_ + i) * NH + hs * NHW + h) * 64 + lane) * 4;
            f32x16 a;
            acc_read_tile(acc[i][h], a);
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = float4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
        }
}

}  // namespace vbmf
