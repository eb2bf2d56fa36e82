// stream_gemm_lds.hpp -- EXPERIMENT (not built into the library): the streaming contraction for H >= 128 with the
// factor operand shared by a workgroup's four waves through LDS.  Result (scripts/bigh_tune.hip,
// profiles/r01_g_bigh_tune.txt): bit-identical to stream_gemm_kernel and NOT faster (H=256: 0.88-0.93 vs
// 0.83-0.86 ms per 2 GB pass; H=128: 0.58 vs 0.56 ms per 2.5 GB pass).  Both kernels sit at ~1.2 PFLOP/s of
// bf16 MFMA work (hi + lo factor parts) on random data, i.e. the H >= 128 passes are MFMA-issue/power bound, not
// factor-traffic bound; the 0.55 ms "no factor loads" ablation that motivated this ran with constant operands
// (higher clocks).  Kept for the record.
//
// Same contract, operands, tilings and result layout as stream_gemm_kernel (stream_gemm.hpp):
//     Out[s][h][x] = sum_{k in split s} F[k][h] * Y[k][x]
// What changes is where the factor operand comes from.  At H >= 128 a k-step needs NF = 8..16 KiB of
// factor fragments against 2-4 KiB of Y per wave, and with every wave fetching its own copy the CU's
// L1 port (64 B/clk) is the limit (measured, scripts/bigh_tune.hip, H = 256: 0.91 ms per 2 GB pass;
// 0.73 ms with the factor L1-hot; 0.55 ms with no factor loads at all).  Here the four waves of a
// workgroup share ONE copy per k-step through LDS (128 B/clk, and 4x fewer L2->L1 bytes):
//   * each wave fetches a quarter of the step's factor fragments into a small register ring (GF steps
//     of L2 latency), writes it to the LDS stage of step s+1 in the middle of step s, and the workgroup
//     meets at ONE raw s_barrier per k-step;
//   * the quarter is written TWO steps ahead into a ring of four LDS stages: stage (s+2)%4 last held step
//     s-2, which every wave finished before the barrier that ended step s-1, and step s+1 has been
//     complete since that same barrier -- so its first fragments are prefetched before the barrier that
//     ends step s and no LDS latency is exposed at a step boundary;
//   * the fragments are consumed from LDS with ds_read_b128 (lane-linear image: conflict-free);
//   * the Y stream is untouched: 1 KiB wave loads straight into a VGPR ring (one wave uses them);
//   * everything is an ordinary buffer load with counted vmcnt waits -- no LDS-DMA: with a DMA in
//     flight hipcc drains vmcnt(0) at every use of a plain load, which would serialise the Y ring;
//   * raw s_barrier + lgkmcnt(0) instead of __syncthreads(), whose fence would drain the rings too.
// The MFMA order per accumulator tile is that of stream_gemm_kernel, so results are bit-identical.
#pragma once
#include "common.hpp"
#include "ctrl_kernels.hpp"
#include "stream_gemm.hpp"

namespace vbmf {

constexpr int LDS_STAGES = 4;

template <int MODE, int NH, int NXW_, int DY, int GF, int RCTRL>
__global__ __launch_bounds__(256) void stream_gemm_lds_kernel(const uint4* __restrict__ Yt,   // [XT][KS][64]
                                                              const uint4* __restrict__ Ft,   // [KS][NPART][NH][64]
                                                              float* __restrict__ Out,        // [nsplit][NH*32][ldOut]
                                                              int XG, int KS, int steps_per_split, int nsplit,
                                                              long long ldOut, const int* __restrict__ stop,
                                                              CtrlArgs ctrl, int xcd_xb) {
    constexpr int NPART = ModeTraits<MODE>::NPART;
    constexpr int NF = NPART * NH;
    constexpr int NFW = NF / 4;                           // fragments each wave fetches per k-step
    static_assert(NF % 4 == 0, "four waves share the factor fetch");
    static_assert(PIPE_D % DY == 0 && DY % GF == 0 && GF + 2 <= DY, "ring depths vs lead-in");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid = blockIdx.x;
    if (ctrl.mode != 0) {                                 // launch carries a control workgroup (dispatched first)
        if (bid == 0) {
            if constexpr (RCTRL > 0) ctrl_chain<RCTRL>(ctrl, smem);
            return;
        }
        bid -= 1;
    }
    if (stop && *stop) return;

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bps = (XG + 3) >> 2;
    int split, xb;
    if (xcd_xb > 0) {
        const int g = bid & 7, j = bid >> 3;
        xb = g * xcd_xb + j % xcd_xb;
        split = j / xcd_xb;
        if (xb >= bps) return;
    } else {
        split = bid / bps;
        xb = bid % bps;
    }
    if (split >= nsplit) return;                          // workgroup-uniform
    // a wave without an x group of its own (ragged last workgroup) still takes part in the factor
    // fetch and the barriers: it streams the last group again and drops the result
    const bool active = xb * 4 + wib < XG;
    const int xg = active ? xb * 4 + wib : XG - 1;

    const long long ks0 = (long long)split * steps_per_split;
    const unsigned ybytes = (unsigned)steps_per_split * 1024u;
    const unsigned fbytes = (unsigned)steps_per_split * (NF * 1024u);
    __amdgpu_buffer_rsrc_t yr[NXW_];
#pragma unroll
    for (int i = 0; i < NXW_; ++i)
        yr[i] = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Yt + (((long long)(xg * NXW_ + i)) * KS + ks0) * 64), 0, ybytes, 0x00020000);
    // this wave's quarter of every k-step's factor fragments
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(Ft + ks0 * (NF * 64) + wib * (NFW * 64)), 0, fbytes, 0x00020000);
    const int voff = lane * 16;

    u32x4v* fl = reinterpret_cast<u32x4v*>(smem);         // [LDS_STAGES][NF][64]
    for (int t = threadIdx.x; t < LDS_STAGES * NF * 64; t += 256) fl[t] = u32x4v{0u, 0u, 0u, 0u};
    __syncthreads();

    f32x16 acc[NXW_][NH];
#pragma unroll
    for (int i = 0; i < NXW_; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;

    u32x4v yb[DY][NXW_];
    u32x4v fs[GF][NFW];
#pragma unroll
    for (int d = 0; d < DY; ++d)
#pragma unroll
        for (int i = 0; i < NXW_; ++i) yb[d][i] = u32x4v{0u, 0u, 0u, 0u};
#pragma unroll
    for (int d = 0; d < GF; ++d)
#pragma unroll
        for (int q = 0; q < NFW; ++q) fs[d][q] = u32x4v{0u, 0u, 0u, 0u};

    // stage of step st lives in LDS slot st mod LDS_STAGES; the loop starts at st = -DY
    int cs = ((-DY) % LDS_STAGES + LDS_STAGES) % LDS_STAGES;
    u32x4v pf[2] = {u32x4v{0u, 0u, 0u, 0u}, u32x4v{0u, 0u, 0u, 0u}};   // fragments 0, 1 of the coming step

    auto mma = [&](const u32x4v fv, int d, int j) __attribute__((always_inline)) {
        if constexpr (MODE == MODE_F32) {
            const f32x4 fe = __builtin_bit_cast(f32x4, fv);
#pragma unroll
            for (int i = 0; i < NXW_; ++i) {
                const f32x4 ye = __builtin_bit_cast(f32x4, yb[d][i]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fe[e], ye[e], acc[i][j], 0, 0, 0);
            }
        } else {
            const bf16x8 fa = __builtin_bit_cast(bf16x8, fv);
#pragma unroll
            for (int i = 0; i < NXW_; ++i)
                acc[i][j % NH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    fa, __builtin_bit_cast(bf16x8, yb[d][i]), acc[i][j % NH], 0, 0, 0);
        }
    };
    constexpr int JW = NF / 2;                    // fragments consumed before the publish
    constexpr int JS = JW + (NF >= 8 ? 4 : 1);    // ... and before the wait that retires it

    for (int s = -DY; s < steps_per_split; s += DY) {
#pragma unroll
        for (int d = 0; d < DY; ++d) {
            const int n1 = (cs + 1) & (LDS_STAGES - 1), w2 = (cs + 2) & (LDS_STAGES - 1);
            const u32x4v* cslot = fl + cs * (NF * 64) + lane;                   // step st = s+d
            const u32x4v* nslot = fl + n1 * (NF * 64) + lane;                   // step st+1 (complete)
            u32x4v* wslot = fl + w2 * (NF * 64) + wib * (NFW * 64) + lane;      // this wave's part of step st+2
            const int gd = d % GF;
#pragma unroll
            for (int j = 0; j < JW; ++j) mma(j < 2 ? pf[j] : cslot[j * 64], d, j);
            // publish this wave's quarter of step st+2 (fetched GF steps ago) and refetch for step st+2+GF
#pragma unroll
            for (int q = 0; q < NFW; ++q) wslot[q * 64] = fs[gd][q];
#pragma unroll
            for (int q = 0; q < NFW; ++q)
                fs[gd][q] = __builtin_amdgcn_raw_buffer_load_b128(fr, voff, ((s + d + 2 + GF) * NF + q) * 1024, 0);
#pragma unroll
            for (int j = JW; j < JS; ++j) mma(j < 2 ? pf[j] : cslot[j * 64], d, j);
            // the LDS writes have landed by now (LDS operations retire in order; only the last fragment reads
            // can still be pending, and the next MFMAs wait for those anyway)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = JS; j < NF; ++j) mma(cslot[j * 64], d, j);
            // prefetch the head of step st+1, refill Y for step st+DY, and meet
            pf[0] = nslot[0];
            pf[1] = nslot[64];
#pragma unroll
            for (int i = 0; i < NXW_; ++i)
                yb[d][i] = __builtin_amdgcn_raw_buffer_load_b128(yr[i], voff, (s + DY + d) * 1024, Y_AUX);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cs = n1;
        }
    }

    if (!active) return;
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
