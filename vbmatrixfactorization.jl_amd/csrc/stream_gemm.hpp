// stream_gemm.hpp -- the streaming contraction both passes of the sweep run on.
//
//     Out[s][h][x] = sum_{k in split s} F[k][h] * Y[k][x]
//
//   pass 1 (updateA!, src/vbmf.jl:98  Y'*BHat):  x = column m, k = row l,    F = BHat,  split-K over l
//   pass 2 (updateB!, src/vbmf.jl:112 Y*AHat):   x = row l,    k = column m, F = AHat
//
// Design (MI355X): the pass is HBM-bound (bf16 Y: 2H flop per 2 bytes = 64 flop/B at H=64 against a
// machine balance of ~300), so the kernel is built as a pure stream:
//   * Y is read exactly once, from its pre-tiled copy, with 1 KiB fully-coalesced wave loads
//     straight into VGPRs -- each fragment is used by one wave only, so an LDS round trip would be
//     pure overhead.  Waves are independent: no LDS, no barriers, nothing to drain.
//   * The factor operand (H x k-step, L2-resident) is read as pre-tiled MFMA A-fragments; in the
//     bf16x2 mode as a hi and a lo bf16 fragment (two MFMAs) so that only Y's own storage rounding
//     remains -- the MFMA pipe has ~4x headroom at H=64.
//   * A register ring of PIPE_D k-steps keeps >= 8 KiB of Y per wave in flight (64+ KiB per CU at
//     8 waves/CU), refilled immediately after each slot is consumed; the prefetch over-reads up to
//     PIPE_D tiles past the split (buffers carry that slack) so no load sits under a branch.
//   * Output tile: MFMA A operand = factor (rows h), B operand = Y (columns x), so every accumulator
//     register is 32 consecutive x of one h row -> two 128-byte segments per store instruction.
//   * Split-K partials go to per-split slabs with plain stores (deterministic; the consumer sums them
//     while it loads), not atomics.
//   * blockIdx -> (split, x-group) keeps the blocks that share an XCD (b % 8) on the same split, so a
//     split's factor fragments are fetched into that XCD's L2 once.
#pragma once
#include "common.hpp"

namespace vbmf {

// cache policy of the Y stream: 2 = nt (streamed once); 0 = default
constexpr int Y_AUX = 2;

template <int MODE, int NH, int NXW_, int D>
__global__ __launch_bounds__(256) void stream_gemm_kernel(const uint4* __restrict__ Yt,   // [XT][KS][64]
                                                          const uint4* __restrict__ Ft,   // [KS][NPART][NH][64]
                                                          float* __restrict__ Out,        // [nsplit][NH*32][ldOut]
                                                          int XG, int KS, int steps_per_split, int nsplit,
                                                          long long ldOut, const int* __restrict__ stop) {
    constexpr int NPART = ModeTraits<MODE>::NPART;
    constexpr int NF = NPART * NH;
    static_assert(PIPE_D % D == 0, "ring depth must divide the padding quantum");
    if (stop && *stop) return;

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform => SGPR addressing
    const int bps = (XG + 3) >> 2;                       // blocks per split
    int split, xb;
    if ((nsplit & 7) == 0) {                              // XCD-aware: blocks b, b+8, ... share an XCD
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        split = xcd + 8 * (q / bps);
        xb = q % bps;
    } else {
        split = blockIdx.x / bps;
        xb = blockIdx.x % bps;
    }
    const int xg = xb * 4 + wib;
    if (xg >= XG || split >= nsplit) return;              // wave-uniform

    // One buffer descriptor per stream, covering exactly this wave's k-range: loads are addressed
    // as (SGPR descriptor, SGPR step offset, VGPR lane*16), the ring's over-read past the range is
    // clamped to zero by the hardware bounds check, and -- unlike plain loads, which hipcc folds back
    // into a load-at-use (phi-of-loads -> load-of-phi) -- the intrinsic keeps the ring in flight.
    const long long ks0 = (long long)split * steps_per_split;
    const unsigned ybytes = (unsigned)steps_per_split * 1024u;
    const unsigned fbytes = (unsigned)steps_per_split * (NF * 1024u);
    __amdgpu_buffer_rsrc_t yr[NXW_];
#pragma unroll
    for (int i = 0; i < NXW_; ++i)
        yr[i] = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Yt + (((long long)(xg * NXW_ + i)) * KS + ks0) * 64), 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(Ft + ks0 * (NF * 64)), 0, fbytes, 0x00020000);
    const int voff = lane * 16;

    f32x16 acc[NXW_][NH];
#pragma unroll
    for (int i = 0; i < NXW_; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;

    // The ring starts as zeros and the loop runs one extra leading iteration (s = -D) whose MFMAs
    // multiply zeros while its refills fetch steps 0..D-1.  With no separate prologue the only
    // load order the waitcnt pass sees is the loop's own, so every in-loop wait is the counted
    // vmcnt((D-1)*(NXW+NF)) family -- a hand-ordered prologue gets re-ordered by instruction
    // selection and forces a near-drain wait inside the loop.
    u32x4v yb[D][NXW_];
    u32x4v fb[D][NF];
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int i = 0; i < NXW_; ++i) yb[d][i] = u32x4v{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < NF; ++j) fb[d][j] = u32x4v{0u, 0u, 0u, 0u};
    }

    for (int s = -D; s < steps_per_split; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // consume ring slot d ...
            if constexpr (MODE == MODE_F32) {
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const f32x4 fe = __builtin_bit_cast(f32x4, fb[d][h]);
#pragma unroll
                    for (int i = 0; i < NXW_; ++i) {
                        const f32x4 ye = __builtin_bit_cast(f32x4, yb[d][i]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(fe[e], ye[e], acc[i][h], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < NPART; ++p)
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const bf16x8 fa = __builtin_bit_cast(bf16x8, fb[d][p * NH + h]);
#pragma unroll
                        for (int i = 0; i < NXW_; ++i)
                            acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                fa, __builtin_bit_cast(bf16x8, yb[d][i]), acc[i][h], 0, 0, 0);
                    }
            }
            // ... and refill it at once with step s+D+d (past the split the descriptor returns zeros).
            // MFMAs read their operands at issue and the wave issues in order, so the refill may
            // target the very registers the MFMAs above just consumed.
            const int sn = s + D + d;
#pragma unroll
            for (int i = 0; i < NXW_; ++i) yb[d][i] = __builtin_amdgcn_raw_buffer_load_b128(yr[i], voff, sn * 1024, Y_AUX);
#pragma unroll
            for (int j = 0; j < NF; ++j)
                fb[d][j] = __builtin_amdgcn_raw_buffer_load_b128(fr, voff, (sn * NF + j) * 1024, 0);
            // pin that order in the emitted stream (otherwise hipcc sinks every refill to the loop
            // bottom and drains vmcnt(0) each iteration)
            constexpr int NMFMA = NXW_ * NF * (MODE == MODE_F32 ? 4 : 1);
            __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, NXW_ + NF, 0);
        }
    }

    const int c = lane & 31, half = lane >> 5;
    float* o = Out + (long long)split * (NH * 32) * ldOut;
#pragma unroll
    for (int i = 0; i < NXW_; ++i) {
        const long long x = (long long)(xg * NXW_ + i) * 32 + c;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(long long)(h * 32 + rho(r, half)) * ldOut + x] = acc[i][h][r];
    }
}

}  // namespace vbmf
