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


// ---------------------------------------------------------------------------------------------------------------------------
//   stream_lds8x_kernel  (the experiment that became stream_gemm.hpp's stream_lds8_kernel; kept with BOTH MFMA shapes for the A/B
//                        record of profiles/r03_c_lds8_probe.txt)  512-thread workgroups (two waves per SIMD), EVERY operand through LDS, filled by LDS-DMA
//                        (buffer_load ... lds: no VGPR destination, 1 KiB per wave-instruction).
// Why (profiles/r03_a_mfma_ceiling_probe.txt, r03_a_pmc_mfma_probe.json): a bare 32x32x16 loop on random operands sustains
// 1.82 PFLOP/s (1.80 GHz, MFMA pipe 100 % busy); the per-wave kernels sit at 1.08-1.2 PFLOP/s with the pipe 56-63 % busy and the
// waves issue-stalled 70-75 % of their cycles: every buffer_load_dwordx4 costs its wave ~50 cycles of issue during which that
// wave issues no MFMA (H = 256: 18 loads per 32 MFMAs -> 1817 cycles per k-step instead of 1024; 12 loads -> 1625), and with one
// wave per SIMD nobody else feeds the pipe.  Here a wave issues 3 loads per 16 MFMAs, a ds_read_b128 costs ~1.5 cycles of issue,
// and the partner wave on the SIMD issues MFMAs while this one issues its loads.
// Geometry: wave tile = 2 x tiles x 4 h tiles (8 accumulator tiles, 128 registers); H = 256: the 8 waves are 4 x pairs x 2 h
// halves (workgroup = 8 x tiles), H = 128: 8 x pairs (16 x tiles).  Either way a k-step is 24 one-KiB pieces (NF factor
// fragments + the workgroup's Y tiles), 3 per wave.  LDS: 6 k-step slots = 3 stages of 2 k-steps = 144 KiB; one raw s_barrier
// per stage; the DMAs of stage s+2 are issued while stage s is consumed (two stages = 4 k-steps ~ 2 us in flight); a wave waits
// for its own pieces with a counted vmcnt and the barrier makes everybody's visible (LDS-DMA data is ordered for a ds_read only
// by the issuing wave's vmcnt followed by a barrier the reader has passed).  Fragments are lane-linear in HBM already, so the
// LDS image is a plain copy and every ds_read_b128 is conflict-free.
// SHAPE 0: v_mfma_f32_32x32x16_bf16, accumulation order identical to stream_gemm_kernel (bit-identical results).
// SHAPE 1: v_mfma_f32_16x16x32_bf16 over the stage's two k-steps (the chip holds a higher clock on this shape: 2.06 vs 1.80 GHz in
//          the bare loops); operands are gathered from the same LDS image with per-lane addresses, the 16x16 accumulators are
//          converted to the 32x32 tile layout at the end (v_permlane16_swap + v_permlane32_swap).
// one 1 KiB LDS-DMA piece: 64 lanes x 16 bytes from (descriptor, lane * 16 + soff) to lds + lane * 16 (lds wave-uniform).
// (Plain functions, not the kernel template: with the address-space cast inside the template's lambda hipcc (ROCm 7.2) silently
// drops the HOST-side instantiation of the kernel -- no diagnostic, the launch stub is simply missing at link time.)
__device__ __forceinline__ void xlds_dma_piece(__amdgpu_buffer_rsrc_t r, unsigned char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void xlds_dma_piece_nt(__amdgpu_buffer_rsrc_t r, unsigned char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 2);
}

template <int NH, int SHAPE, int YAUX>
__global__ __launch_bounds__(512) void stream_lds8x_kernel(const uint4* __restrict__ Yt,   // [XT][KS][64]
                                                          const uint4* __restrict__ Ft,   // [KS][2][NH][64]
                                                          float* __restrict__ Out, int XT, int KS, int steps_per_split, int nsplit,
                                                          long long ldOut, int frag_out) {
    constexpr int NF = 2 * NH;
    constexpr int HS = NH / 4;                 // waves that share an x pair (h slices of 4 tiles)
    constexpr int XPW = 8 / HS;                // x pairs per workgroup
    constexpr int YT = 2 * XPW;                // Y tiles per workgroup
    constexpr int SLOT = (NF + YT) * 1024;     // bytes of one k-step: 24 KiB
    constexpr int NYD = 2 / HS;                // Y pieces this wave fetches per k-step
    constexpr int NFD = NF / 8;                // factor pieces this wave fetches per k-step
    constexpr int NDMA = NYD + NFD;            // = 3
    static_assert(NH == 4 || NH == 8, "wave tile is 2 x 4 tiles");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npair = XT >> 1;
    const int bps = (npair + XPW - 1) / XPW;
    const int split = blockIdx.x / bps, xb = blockIdx.x % bps;
    if (split >= nsplit) return;               // workgroup-uniform
    const int xp = wib / HS, hs = wib % HS;
    const int pair = xb * XPW + xp;
    const bool active = pair < npair;
    const int tile0 = (active ? pair : npair - 1) * 2;      // an idle wave streams the last pair again (it still feeds the ring)

    const long long ks0 = (long long)split * steps_per_split;
    const unsigned ybytes = (unsigned)(steps_per_split + PIPE_D) * 1024u;
    const unsigned fbytes = (unsigned)(steps_per_split + PIPE_D) * (NF * 1024u);
    __amdgpu_buffer_rsrc_t yr[NYD];
#pragma unroll
    for (int i = 0; i < NYD; ++i)
        yr[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(Yt + (((long long)(tile0 + (HS == 2 ? hs : i))) * KS + ks0) * 64), 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(Ft + ks0 * (NF * 64) + wib * (NFD * 64)), 0, fbytes, 0x00020000);
    const int voff = lane * 16;
    // this wave's pieces of k-step `step` into slot `slot`
    auto dma_step = [&](int step, int slot) __attribute__((always_inline)) {
        unsigned char* base = smem + slot * SLOT;
#pragma unroll
        for (int q = 0; q < NFD; ++q)
            xlds_dma_piece(fr, base + (wib * NFD + q) * 1024, voff, (step * NF + q) * 1024);
#pragma unroll
        for (int i = 0; i < NYD; ++i) {
            unsigned char* dst = base + (NF + xp * 2 + (HS == 2 ? hs : i)) * 1024;
            if constexpr (YAUX == 2) xlds_dma_piece_nt(yr[i], dst, voff, step * 1024);
            else xlds_dma_piece(yr[i], dst, voff, step * 1024);
        }
    };

    f32x16 acc[2][4];                          // SHAPE 0: 32 x 32 tiles; SHAPE 1: filled from the 16 x 16 tiles at the end
    f32x4 acq[SHAPE == 1 ? 2 : 1][SHAPE == 1 ? 4 : 1][2][2];     // SHAPE 1: [tx][th][hi][xi] 16 x 16 tiles
    if constexpr (SHAPE == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acq[i][h][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // SHAPE 1: a 16x16x32 operand is 16 rows x 32 k = lane (q, r): k group q, row r.  k group q <-> (k-step q & 1 of the stage,
    // lane half q >> 1 of the 32x32x16 fragment); the same assignment on both operands, so the products pair up.  Lane (q, r) of
    // sub-tile xi therefore reads the 16 bytes of fragment lane (half = q >> 1, c = 16 xi + r) in slot q & 1: byte offset
    // (q & 1) * SLOT + ((q >> 1) * 32 + 16 xi + r) * 16.  (Rows of 16 lanes stay 256 contiguous bytes: conflict-free.)
    const int q16 = lane >> 4, r16 = lane & 15;
    const int lo16 = (q16 & 1) * SLOT + ((q16 >> 1) * 32 + r16) * 16;      // + 256 for the upper 16 rows of a 32-wide tile

    const int nst = steps_per_split >> 1;      // stages of two k-steps
    dma_step(0, 0); dma_step(1, 1); dma_step(2, 2); dma_step(3, 3);
    const u32x4v* lds4 = reinterpret_cast<const u32x4v*>(smem);

    for (int st0 = 0; st0 < nst; st0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int st = st0 + u;
            if (st >= nst) break;              // wave-uniform
            // my pieces of stage st have landed (those of stage st + 1 may still be in flight) ...
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            // ... and after the barrier everybody's have; every wave is also done reading stage st - 1, whose slots are refilled now
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (SHAPE == 0) {
                // k-step 0 of the stage: its fragments cannot be requested before the barrier (exposed LDS latency, once per
                // stage); k-step 1's are requested between the MFMAs of k-step 0.  The pieces of stage st + 2 go out between
                // the MFMAs as well: one DMA costs its wave ~50-60 cycles of issue, which the partner wave on the SIMD fills.
                u32x4v fa[10], fb[10];
                const u32x4v* s0 = lds4 + (2 * u) * (SLOT / 16) + lane;
                const u32x4v* s1 = lds4 + (2 * u + 1) * (SLOT / 16) + lane;
                auto frag_at = [&](const u32x4v* sl, int i) __attribute__((always_inline)) {
                    return i < 2 ? sl[(NF + xp * 2 + i) * 64] : sl[(((i - 2) >> 2) * NH + hs * 4 + ((i - 2) & 3)) * 64];
                };
                auto mma = [&](const u32x4v (&f)[10]) __attribute__((always_inline)) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int th = 0; th < 4; ++th)
#pragma unroll
                            for (int tx = 0; tx < 2; ++tx)
                                acc[tx][th] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[2 + p * 4 + th]),
                                                                                    __builtin_bit_cast(bf16x8, f[tx]), acc[tx][th], 0, 0, 0);
                };
#pragma unroll
                for (int i = 0; i < 10; ++i) fa[i] = frag_at(s0, i);
#pragma unroll
                for (int i = 0; i < 10; ++i) fb[i] = frag_at(s1, i);
                mma(fa);
                dma_step(2 * (st + 2), 2 * ((u + 2) % 3));
                // pinned order: the 10 reads of k-step 0, then per MFMA of k-step 0 one read of k-step 1, then the 3 pieces
                __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                mma(fb);
                dma_step(2 * (st + 2) + 1, 2 * ((u + 2) % 3) + 1);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            } else {
                // both k-steps of the stage at once: 4 Y + 16 factor operands (16-byte reads at per-lane addresses), 64 MFMAs.
                // The factor's lo part is requested between the MFMAs of the hi part.
                const unsigned char* sb = smem + (2 * u) * SLOT + lo16;
                auto opnd = [&](int frag, int sub) __attribute__((always_inline)) {
                    return *reinterpret_cast<const u32x4v*>(sb + frag * 1024 + sub * 256);
                };
                u32x4v yv[2][2], f0[4][2], f1[4][2];
#pragma unroll
                for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                    for (int xi = 0; xi < 2; ++xi) yv[tx][xi] = opnd(NF + xp * 2 + tx, xi);
#pragma unroll
                for (int th = 0; th < 4; ++th)
#pragma unroll
                    for (int hi = 0; hi < 2; ++hi) f0[th][hi] = opnd(hs * 4 + th, hi);
#pragma unroll
                for (int th = 0; th < 4; ++th)
#pragma unroll
                    for (int hi = 0; hi < 2; ++hi) f1[th][hi] = opnd(NH + hs * 4 + th, hi);
                auto mma16 = [&](const u32x4v (&f)[4][2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int th = 0; th < 4; ++th)
#pragma unroll
                        for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                                for (int xi = 0; xi < 2; ++xi)
                                    acq[tx][th][hi][xi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf16x8, f[th][hi]), __builtin_bit_cast(bf16x8, yv[tx][xi]), acq[tx][th][hi][xi], 0, 0, 0);
                };
                mma16(f0);
                dma_step(2 * (st + 2), 2 * ((u + 2) % 3));
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                mma16(f1);
                dma_step(2 * (st + 2) + 1, 2 * ((u + 2) % 3) + 1);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // run-ahead pieces (slack of the tiled buffers) still target this LDS
    if (!active) return;
    if constexpr (SHAPE == 1) {
        // 16 x 16 tile (hi, xi): lane (q, c) holds column x = 16 xi + c, rows h = 16 hi + 4 q + t in register t.  The 32 x 32 tile
        // wants lane (half, c') = column c', register r = 4 g + t <-> row (r & 3) + 8 (r >> 2) + 4 half, i.e. half = q & 1,
        // g = 2 hi + (q >> 1).  With S0, S1 = register t of sub-tiles xi = 0, 1 as rows of 16 lanes [r0 r1 r2 r3]:
        // v_permlane16_swap -> [S0.r0 S1.r0 S0.r2 S1.r2], [S0.r1 S1.r1 S0.r3 S1.r3]; v_permlane32_swap of those two ->
        // [S0.r0 S1.r0 S0.r1 S1.r1] = register 4 (2 hi) + t and [S0.r2 S1.r2 S0.r3 S1.r3] = register 4 (2 hi + 1) + t.
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float s0 = acq[i][h][hi][0][t], s1 = acq[i][h][hi][1][t];
                        const auto a = __builtin_amdgcn_permlane16_swap(fbits(s0), fbits(s1), false, false);
                        const auto b = __builtin_amdgcn_permlane32_swap(a[0], a[1], false, false);
                        acc[i][h][4 * (2 * hi) + t] = bitsf(b[0]);
                        acc[i][h][4 * (2 * hi + 1) + t] = bitsf(b[1]);
                    }
    }
    if (frag_out) {
        float4* o4 = reinterpret_cast<float4*>(Out + (long long)split * (NH * 32) * ldOut);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float4* t = o4 + (((long long)(tile0 + i) * NH + hs * 4 + h) * 64 + lane) * 4;
                const f32x16 a = acc[i][h];
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = float4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            }
    } else {
        const int c = lane & 31, half = lane >> 5;
        float* o = Out + (long long)split * (NH * 32) * ldOut;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long x = (long long)(tile0 + i) * 32 + c;
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(long long)((hs * 4 + h) * 32 + rho(r, half)) * ldOut + x] = acc[i][h][r];
        }
    }
}

}  // namespace vbmf
