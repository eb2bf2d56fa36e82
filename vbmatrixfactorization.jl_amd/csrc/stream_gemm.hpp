// stream_gemm.hpp -- the streaming contraction both passes of the sweep run on.
//
//     Out[s][h][x] = sum_{k in split s} F[k][h] * Y[k][x]
//
//   pass 1 (updateA!, src/vbmf.jl:98  Y'*BHat):  x = column m, k = row l,    F = BHat,  split-K over l
//   pass 2 (updateB!, src/vbmf.jl:112 Y*AHat):   x = row l,    k = column m, F = AHat
//
// Design (MI355X): the pass is HBM-bound (bf16 Y: 2H flop per 2 bytes = 64 flop/B at H=64 against a
// machine balance of ~300), so the kernel is built as a pure stream:
//   * Y is read exactly once, from its pre-tiled copy, with 1 KiB fully-coalesced wave loads (nt)
//     straight into VGPRs -- each fragment is used by one wave only, so an LDS round trip would be
//     pure overhead.  Waves are independent: no LDS, no barriers, nothing to drain.
//   * The factor operand (H x k-step, L2-resident) is read as pre-tiled MFMA A-fragments; in the
//     bf16x2 mode as a hi and a lo bf16 fragment (two MFMAs) so that only Y's own storage rounding
//     remains -- the MFMA pipe has ~4x headroom at H=64.  Its L2->L1 traffic is what costs
//     bandwidth (measured: 4.2 -> 5.7 TB/s when it is removed), so a wave covers NXW = 8/NH
//     32-wide x tiles per factor fragment (128 accumulator registers).
//   * Two register rings, refilled in place right after use: DY k-steps of Y (HBM latency; a CU's
//     throughput is bytes-in-flight / latency, measured ~30 GB/s at 48 KiB) and DF k-steps of the
//     factor (L2 latency).  The rings start as zeros and the loop runs DY lead-in steps on them, so
//     the only load order the waitcnt pass sees is the loop's own and every wait is a counted vmcnt.
//   * Loads are SGPR-descriptor buffer loads (descriptor = the wave's stream, SGPR step offset, VGPR
//     lane*16): unlike plain loads, which hipcc folds back into a load-at-use (phi-of-loads ->
//     load-of-phi), the intrinsic keeps the rings in flight.  Run-ahead past a split reads the next
//     tiles / the buffers' slack and is discarded.
//   * Output tile: MFMA A operand = factor (rows h), B operand = Y (columns x), so every accumulator
//     register is 32 consecutive x of one h row -> two 128-byte segments per store instruction.
//   * Split-K partials go to per-split slabs with plain stores (deterministic; the consumer sums them
//     while it loads), not atomics.
//   * Workgroups 0 and 1 of a launch may run the sweep's H x H control chain (ctrl_kernels.hpp, CtrlArgs)
//     instead of streaming: its inputs are complete before the launch and its outputs are needed only
//     after it, so the chain overlaps the pass for free.
//   * Grid: per-CU throughput is latency-bound, so time = the most loaded CU: the planner (host) picks
//     the split factor that makes the block count one balanced wave over the 256 CUs
//     (measured at 100k x 10k: 240 blocks 5.7 TB/s, 260 blocks 3.3 TB/s).
#pragma once
#include "common.hpp"
#include "ctrl_kernels.hpp"
#include "post_kernels.hpp"

namespace vbmf {

// cache policy of the Y stream: 2 = nt (streamed once); 0 = default
constexpr int Y_AUX = 2;

// tuning switch of the register epilogue (A/B on the GPU, see profiles/): 1 = the previous factor's rows are loaded one tile
// ahead (costs 32 more live registers per column tile), 0 = at the head of each tile
#ifndef VBMF_EPI_PV_AHEAD
#define VBMF_EPI_PV_AHEAD 1
#endif

// Epilogue of the Y*A pass (EPI = 1, un-split pass, H <= 64): the product tiles never leave the registers -- B = (Y A) SigmaB
// / sigma2, its operand tiles, the fp32 factor and the Gram / delta-Gram partials are produced right here
// (post_kernels.hpp, post_gram_tile_regs), which removes the L x H round trip through HBM and one launch per sweep.
// SigmaB is computed by control workgroup 0 of this very launch: `sready` is its release flag.
// (The flag may only be released by a workgroup of THIS launch -- those are dispatched first.  A stand-alone kernel on
// another stream can be starved by the pass occupying every CU while its workgroups wait in their epilogues: tried for
// H >= 128 with SigmaB on the side stream, it deadlocked until the bounded spin gave up.  A product-only epilogue for
// H >= 128 (table streamed from L2, 2048 exact-f32 MFMAs per wave) was also measured: no faster than the separate
// post kernel at 1M x 128, so H >= 128 keeps the product in HBM.)
struct EpiArgs {
    const float* S; float* Fac; const float* Prev; uint4* Ft; float* slabs;
    const int* sready; int expect;           // expect < 0: the table was complete before the launch
    int spin_limit;                          // bounded wait for `sready` (polls of ~0.4 us each); on expiry *err = 2
    int* err;
    int store_fac;                           // 0 inside vbmf_run: the fp32 factor is rebuilt from the tiles once, at the end
    int frag_out;                            // EPI = 0 only: write the product in FRAGMENT-MAJOR order (below)
    double* trpart;                          // EPI = 1: per-wave shares of tr(B'YA) = sum (Y A) o BHat  [4 * workgroups]
    unsigned long long* stamp;               // EPI = 1: durations of workgroup 0's tail in 10 ns ticks: [0] wait for + load of the
                                             // SigmaB table, [1] the tiles' post / Gram work, [2] fold + slab store
    // EPI = 1: workgroup -> x-group map: workgroup b takes g_base x groups, the first g_rem workgroups one more (<= 4: one per
    // wave; the other waves idle).  Default: four per workgroup (782 groups of 100k rows = 196 workgroups).  Both ways of
    // putting the idle 60 CUs to work were measured and are SLOWER (profiles/r02_f_epilogue_pass_ab.txt): three groups per
    // workgroup on 254 CUs 0.372 vs 0.366 ms (with a 6- or a 12-deep Y ring alike), single tiles dealt to the waves (3 of the 4
    // accumulator tiles busy, the fourth behind a zero-record descriptor) 0.3725 vs 0.354 ms -- the four waves of a CU share
    // the factor operand's L1 lines, fewer / narrower waves per CU pay for that operand again.
    int g_base, g_rem;
};

// FDBG (tuning harness only): 1 = the factor ring re-reads one L1-hot k-step, 2 = no factor refills at all
template <int MODE, int NH, int NXW_, int DY, int DF, int RCTRL, int FDBG = 0, int EPI = 0>
__global__ __launch_bounds__(256) void stream_gemm_kernel(const uint4* __restrict__ Yt,   // [XT][KS][64]
                                                          const uint4* __restrict__ Ft,   // [KS][NPART][NH][64]
                                                          float* __restrict__ Out,        // [nsplit][NH*32][ldOut]
                                                          int XG, int KS, int steps_per_split, int nsplit,
                                                          long long ldOut, const int* __restrict__ stop,
                                                          CtrlArgs ctrl, int xcd_xb, EpiArgs epi = EpiArgs{}) {
    constexpr int NPART = ModeTraits<MODE>::NPART;
    constexpr int NF = NPART * NH;
    static_assert(PIPE_D % DY == 0 && DY % DF == 0, "ring depths must divide the padding quantum");
    constexpr bool FINE = (NH >= 4);
    int bid = blockIdx.x;
    if (ctrl.mode != 0) {                                 // launch carries two control workgroups (dispatched first)
        if (bid < 2) {
            if constexpr (RCTRL > 0) {
                extern __shared__ __attribute__((aligned(16))) unsigned char ctrl_lds[];
                ctrl_chain<RCTRL>(ctrl, ctrl_lds, bid);
            }
            return;
        }
        bid -= 2;
    }
    if (stop && *stop) return;

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform => SGPR addressing
    const int bps = (XG + 3) >> 2;                       // blocks per split
    int split, xb;
    if (xcd_xb > 0) {
        // XCD-aware map for split-K launches.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
        // with the linear map every XCD sees every split and fetches every split's factor k-range from HBM/MALL
        // (measured at 100k x 10k, H = 64: 2.28 GB fetched per launch against 2.03 GB algorithmic).  Here XCD g = bid % 8
        // takes the CONTIGUOUS range [g*per, (g+1)*per) of the split-major work list (per = xcd_xb), i.e. ~nsplit/8
        // splits: each split's factor tiles are then fetched by one or two XCDs only.
        const int g = bid & 7, j = bid >> 3;
        const int w = g * xcd_xb + j;
        if (j >= xcd_xb || w >= bps * nsplit) return;
        split = w / bps;
        xb = w % bps;
    } else {
        split = bid / bps;
        xb = bid % bps;
    }
    int xg = xb * 4 + wib;
    bool active = true;
    int t0 = xg * NXW_, nt = NXW_;                         // first x tile of this wave, number of its tiles
    __shared__ int epi_abort;                              // EPI: a wave's hand-off wait expired (see the epilogue)
    if constexpr (EPI) {
        // every wave of the workgroup meets in the epilogue's fold: a wave without tiles streams nothing
        if (split >= nsplit) return;                       // workgroup-uniform
        if (threadIdx.x == 0) epi_abort = 0;
        __syncthreads();
        const int gb = epi.g_base + (xb < epi.g_rem ? 1 : 0);              // x groups of this workgroup
        xg = xb * epi.g_base + (xb < epi.g_rem ? xb : epi.g_rem) + wib;
        active = wib < gb;
        t0 = xg * NXW_;
        if (!active) { xg = 0; t0 = 0; nt = 0; steps_per_split = -DY; }
    } else {
        if (xg >= XG || split >= nsplit) return;          // wave-uniform
    }

    const long long ks0 = (long long)split * steps_per_split;
    const unsigned ybytes = (unsigned)steps_per_split * 1024u;
    const unsigned fbytes = (unsigned)steps_per_split * (NF * 1024u);
    __amdgpu_buffer_rsrc_t yr[NXW_];
#pragma unroll
    for (int i = 0; i < NXW_; ++i)
        yr[i] = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(Yt + (((long long)(t0 + (i < nt ? i : 0))) * KS + ks0) * 64), 0, i < nt ? ybytes : 0u, 0x00020000);
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

    u32x4v yb[DY][NXW_];
    u32x4v fb[DF][NF];
#pragma unroll
    for (int d = 0; d < DY; ++d)
#pragma unroll
        for (int i = 0; i < NXW_; ++i) yb[d][i] = u32x4v{0u, 0u, 0u, 0u};
#pragma unroll
    for (int d = 0; d < DF; ++d)
#pragma unroll
        for (int j = 0; j < NF; ++j) fb[d][j] = u32x4v{0u, 0u, 0u, 0u};

    for (int s = -DY; s < steps_per_split; s += DY) {
#pragma unroll
        for (int d = 0; d < DY; ++d) {
            constexpr int UNUSED = 0; (void)UNUSED;
            const int fd = d % DF;
            // consume Y slot d and factor slot d % DF (lead-in iterations multiply zeros) ...
            if constexpr (FINE) {
                // H >= 128: a k-step is 32 MFMAs (~1000 cycles) on NF factor fragments.  Refill every factor
                // fragment right after ITS MFMAs (not after the whole step), so the DF-deep ring really
                // gives DF k-steps of L2 latency to each fragment; the Y fragments follow their last use.
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if constexpr (MODE == MODE_F32) {
                        const f32x4 fe = __builtin_bit_cast(f32x4, fb[fd][j]);
#pragma unroll
                        for (int i = 0; i < NXW_; ++i) {
                            const f32x4 ye = __builtin_bit_cast(f32x4, yb[d][i]);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fe[e], ye[e], acc[i][j], 0, 0, 0);
                        }
                    } else {
                        const bf16x8 fa = __builtin_bit_cast(bf16x8, fb[fd][j]);
#pragma unroll
                        for (int i = 0; i < NXW_; ++i)
                            acc[i][j % NH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                fa, __builtin_bit_cast(bf16x8, yb[d][i]), acc[i][j % NH], 0, 0, 0);
                    }
                    if constexpr (FDBG != 2)
                        fb[fd][j] = __builtin_amdgcn_raw_buffer_load_b128(fr, voff, ((FDBG == 1 ? d : s + d + DF) * NF + j) * 1024, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NXW_ * (MODE == MODE_F32 ? 4 : 1), 0);
                    if constexpr (FDBG != 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < NXW_; ++i)
                    yb[d][i] = __builtin_amdgcn_raw_buffer_load_b128(yr[i], voff, (s + DY + d) * 1024, Y_AUX);
                __builtin_amdgcn_sched_group_barrier(0x020, NXW_, 0);
            } else {
                if constexpr (MODE == MODE_F32) {
    #pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const f32x4 fe = __builtin_bit_cast(f32x4, fb[fd][h]);
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
                            const bf16x8 fa = __builtin_bit_cast(bf16x8, fb[fd][p * NH + h]);
    #pragma unroll
                            for (int i = 0; i < NXW_; ++i)
                                acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                    fa, __builtin_bit_cast(bf16x8, yb[d][i]), acc[i][h], 0, 0, 0);
                        }
                }
                // ... and refill both at once (Y: step +DY, factor: step +DF).  MFMAs read their operands at
                // issue and a wave issues in order, so the refill may target the registers just consumed.
                // The lead-in's factor steps are negative: they read the previous split's tiles or the PIPE_D
                // zero tiles in FRONT of the factor buffer (finite data times the zero Y ring).  The run-ahead
                // past the split lands in the next split's tiles or in the trailing slack and is never
                // consumed.  Nothing relies on the descriptor's bounds check (the SGPR offset is not part of
                // it).  (Clamping the lead-in step with max(.,0) instead cost 12 %: hipcc peels the lead-in
                // iteration and its waits serialise every wave's pipeline fill.)
    #pragma unroll
                for (int i = 0; i < NXW_; ++i)
                    yb[d][i] = __builtin_amdgcn_raw_buffer_load_b128(yr[i], voff, (s + DY + d) * 1024, Y_AUX);
    #pragma unroll
                for (int j = 0; j < NF; ++j)
                    fb[fd][j] = __builtin_amdgcn_raw_buffer_load_b128(fr, voff, ((s + d + DF) * NF + j) * 1024, 0);
                // pin that order in the emitted stream (otherwise hipcc sinks every refill to the loop bottom
                // and drains vmcnt(0) each iteration)
                constexpr int NMFMA = NXW_ * NF * (MODE == MODE_F32 ? 4 : 1);
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, NXW_ + NF, 0);
            }
        }
    }

    if constexpr (EPI) {
        static_assert(EPI == 0 || NH <= 2, "register epilogue keeps NH(NH+1)/2 pair tiles per Gram");
        constexpr int NPAIR = NH * (NH + 1) / 2;
        constexpr int Hp = NH * 32;
        __shared__ float fold[2 * NPAIR * 16 * 64];
        __shared__ float tbuf[4][32 * TB_LD];
        double trd = 0.0;
        const int c = lane & 31, half = lane >> 5;
        const unsigned long long e0 = wall_clock64();      // (stamps: the tail of workgroup 0, see VBMF_PEEK_CHAIN)
        if (epi.expect >= 0) {                             // SigmaB / sigma2 is written by workgroup 0 of this launch
            int spins = 0;
            while (__hip_atomic_load(epi.sready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epi.expect) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > epi.spin_limit) {                                                    // never hang the device
                    if (lane == 0) { atomicExch(epi.err, 2); atomicExch(&epi_abort, 1); }
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        // the SigmaB / sigma2 table goes to LDS once per workgroup and is read from there as the MFMA's B operand (bf16 factor
        // modes: pre-split into three bf16 parts, post_kernels.hpp load_sigma_table): as registers it pushed the tile body
        // over the register file
        __shared__ __attribute__((aligned(16))) float stab[sigma_lds_floats<MODE, NH>()];
        load_sigma_table<MODE, NH>(stab, epi.S);
        __syncthreads();
        // The table never came (VBMF_ERR_SYNC): leave -- workgroup-uniformly, the flag is in LDS -- WITHOUT running the tile
        // body: the previous factor's operand tiles, the fp32 factor and the Gram slabs keep their last valid contents instead of
        // being overwritten with products of a stale table (round 2 went on and destroyed them; the host reported the error
        // either way, but a caller that caught it was left with a corrupted device state).
        if (epi_abort) return;
        const unsigned long long e1 = wall_clock64();
        f32x16 G[NPAIR], D[NPAIR];
#pragma unroll
        for (int p = 0; p < NPAIR; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) { G[p][r] = 0.f; D[p][r] = 0.f; }
        if (active) {
            // the previous factor's rows of all this wave's tiles first (they are only needed for the delta-Gram at the
            // end of each tile: one round of memory latency instead of one per tile)
            // read from the factor's OPERAND TILES (they encode exactly the fp32 factor): 1 KiB wave loads
            // (one tile ahead, not all NXW at once: with the tr(B'YA) block in the tile body the full prefetch spilled)
#if VBMF_EPI_PV_AHEAD
            f32x16 pvn[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) read_factor_tiles<MODE, NH>(epi.Ft, pvn[h], t0, h, lane);
#pragma unroll
            for (int i = 0; i < NXW_; ++i) {
                if (i >= nt) break;                        // wave-uniform
                f32x16 pvc[NH];
#pragma unroll
                for (int h = 0; h < NH; ++h) pvc[h] = pvn[h];
                if (i + 1 < nt) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) read_factor_tiles<MODE, NH>(epi.Ft, pvn[h], t0 + i + 1, h, lane);
                }
                // (this tile's product registers leave the AGPR file here, by explicit reads: common.hpp, acc_read_tile)
                f32x16 qv[NH];
#pragma unroll
                for (int h = 0; h < NH; ++h) acc_read_tile(acc[i][h], qv[h]);
                post_gram_tile_regs<MODE, NH>(qv, stab, t0 + i, epi.Fac, epi.Prev, epi.Ft, lane, G, D, pvc,
                                              epi.store_fac, tbuf[wib], trd);
            }
#else
#pragma unroll
            for (int i = 0; i < NXW_; ++i) {               // loads issued at the head of the tile, consumed after its product
                if (i >= nt) break;                        // wave-uniform
                f32x16 pvc[NH];
#pragma unroll
                for (int h = 0; h < NH; ++h) read_factor_tiles<MODE, NH>(epi.Ft, pvc[h], t0 + i, h, lane);
                post_gram_tile_regs<MODE, NH>(acc[i], stab, t0 + i, epi.Fac, epi.Prev, epi.Ft, lane, G, D, pvc,
                                              epi.store_fac, tbuf[wib], trd);
            }
#endif
        }
        store_wave_dot(trd, epi.trpart + (long long)xb * 4 + wib, lane);   // (an idle wave contributes 0)
        const unsigned long long e2 = wall_clock64();
        // fold the four waves' partials (fixed order => deterministic), then one coalesced slab store per workgroup
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
        float* o = epi.slabs + (long long)xb * (2 * NPAIR * 1024);
        for (int i = threadIdx.x; i < 2 * NPAIR * 1024; i += 256) o[i] = fold[i];
        if (xb == 0 && threadIdx.x == 0 && epi.stamp != nullptr) {
            const unsigned long long e3 = wall_clock64();
            epi.stamp[0] = e1 - e0; epi.stamp[1] = e2 - e1; epi.stamp[2] = e3 - e2;
        }
    } else if (epi.frag_out) {
        // Fragment-major product for the H >= 128 post kernel: tile (x tile, h tile) is 64 lanes x 16 registers, each lane's
        // registers contiguous -- four 16-byte stores per tile here and four 16-byte loads there, instead of sixteen 4-byte
        // row accesses on both sides (the consumer takes register r as the operand of MFMA step r: post_frag_kernel).
        // (H >= 128: the accumulators leave the AGPR file by explicit reads, tile by tile -- common.hpp, acc_read_tile)
        float4* o4 = reinterpret_cast<float4*>(Out + (long long)split * (NH * 32) * ldOut);
#pragma unroll
        for (int i = 0; i < NXW_; ++i)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float4* t = o4 + (((long long)(xg * NXW_ + i) * NH + h) * 64 + lane) * 4;
                f32x16 a;
                if constexpr (FINE) acc_read_tile(acc[i][h], a);
                else a = acc[i][h];
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = float4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            }
    } else {
        const int c = lane & 31, half = lane >> 5;
        float* o = Out + (long long)split * (NH * 32) * ldOut;
#pragma unroll
        for (int i = 0; i < NXW_; ++i) {
            const long long x = (long long)(xg * NXW_ + i) * 32 + c;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                f32x16 a;
                if constexpr (FINE) acc_read_tile(acc[i][h], a);
                else a = acc[i][h];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(long long)(h * 32 + rho(r, half)) * ldOut + x] = a[r];
            }
        }
    }
}

// ===========================================================================================================================
// stream_lds8_kernel -- the H >= 128 streaming contraction (bf16x2 operands): 512-thread workgroups (two waves per SIMD),
// EVERY operand through LDS, filled by LDS-DMA (buffer_load ... lds: no VGPR destination, 1 KiB per wave-instruction).
//
// Why (round 3, profiles/r03_a_mfma_ceiling_probe.txt, r03_a_pmc_mfma_probe.json, r03_c_lds8_probe.txt): these passes are
// MFMA-bound (hi + lo factor parts), and a bare 32x32x16 loop on random operands sustains 1.82 PFLOP/s here (1.80 GHz under the
// chip's power management, matrix pipe 100 % busy; the 16x16x32 shape 2.06 PFLOP/s at 2.06 GHz).  The per-wave kernel above
// sat at 1.08-1.2 PFLOP/s with the pipe 56-63 % busy and its waves issue-stalled 70-75 % of their cycles: every
// buffer_load_dwordx4 costs the issuing wave ~50 cycles during which it issues no MFMA (H = 256: 18 loads per 32 MFMAs -> 1817
// cycles per k-step instead of 1024; with a 4 x 4 wave tile 12 loads -> 1625), and at one wave per SIMD nobody else feeds the
// pipe.  Here a wave issues 3 DMA loads per 16 MFMAs' worth of work, its operand reads are ds_read_b128 (~1.5 cycles of
// issue each), and the partner wave on the SIMD issues MFMAs meanwhile: pipe 66-74 % busy, 17 % fewer cycles; the chip answers
// with a lower clock (1.76 instead of 1.92 GHz), so the time gain is 8-11 %, and the 16x16x32 shape -- less accumulator traffic
// per flop, a higher sustained clock -- gives 4-7 % more: H = 256 0.755 / 0.799 ms against 0.88 / 0.945 ms per pass, H = 128
// 0.522 / 0.509 against 0.565 / 0.520.
//
// Geometry: wave tile = 2 x tiles x 4 h tiles (8 accumulator tiles, 128 registers).  H = 256: the 8 waves are 4 x pairs x 2 h
// halves (workgroup = 8 x tiles), H = 128: 8 x pairs (16 x tiles) -- the footprints of the per-wave kernel, so the host's plan
// (block counts, split-K, padding quanta) is unchanged.  Either way a k-step is 24 one-KiB pieces (NF factor fragments + the
// workgroup's Y tiles), 3 per wave.  LDS: 6 k-step slots = 3 stages of 2 k-steps = 144 KiB, one workgroup per CU.
// Pipeline: one raw s_barrier per stage; the pieces of stage s+2 are issued while stage s is consumed (two stages = 4 k-steps
// ~ 2 us in flight); a wave waits for its OWN pieces of stage s with a counted vmcnt and the barrier makes everybody's visible
// (LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier the reader has passed); the same
// barrier says every wave is done reading stage s-1, whose slots the new pieces overwrite.  No plain (VGPR-destination) load
// exists in the loop: beside LDS-DMA hipcc would wait vmcnt(0) at every use of one.  Fragments are lane-linear in HBM
// already, so the LDS image is a plain copy and every ds_read_b128 is conflict-free; the run-ahead past a split's end lands
// in the next split's tiles or the buffers' PIPE_D slack and is never consumed.
// MFMA shape: v_mfma_f32_16x16x32_bf16 over the stage's two k-steps.  A 16x16x32 operand is 16 rows x 32 k = lane (q, r):
// k group q, row r; k group q <-> (k-step q & 1 of the stage, lane half q >> 1 of the 32x32x16 fragment), the same assignment
// on both operands so the products pair up, which makes the operand of sub-tile xi a 16-byte read at the per-lane offset
// (q & 1) * SLOT + ((q >> 1) * 32 + 16 xi + r) * 16 of the unchanged image (rows of 16 lanes stay 256 contiguous bytes:
// conflict-free).  The 16 x 16 accumulators are converted to the 32 x 32 tile layout of common.hpp at the end
// (v_permlane16_swap + v_permlane32_swap), so every consumer sees the layouts it always saw.  Against the 32x32x16 kernel the
// results differ by fp32 summation order only (measured 1.5e-6 of the largest element); they are deterministic.
// Workgroups 0 and 1 may run the sweep's control chain on their first 256 threads (ctrl_kernels.hpp, ctrl_nthreads()).

// (lds_dma_piece / lds_dma_piece_nt: common.hpp)
constexpr int LDS8_SLOT_BYTES = 24 * 1024;            // one k-step: NF factor fragments + the workgroup's Y tiles
constexpr int LDS8_BYTES = 6 * LDS8_SLOT_BYTES;       // dynamic LDS of a launch (>= the control chain's need, ctrl_lds_bytes)

template <int NH, int RCTRL, bool SK = false>
__global__ __launch_bounds__(512) void stream_lds8_kernel(const uint4* __restrict__ Yt,   // [XT][KS][64]
                                                          const uint4* __restrict__ Ft,   // [KS][2][NH][64]
                                                          float* __restrict__ Out,        // [nsplit][NH*32][ldOut] or fragment-major
                                                          int XT, int KS, int steps_per_split, int nsplit, long long ldOut,
                                                          const int* __restrict__ stop, CtrlArgs ctrl, int xcd_xb, int frag_out,
                                                          int sk_per = 0, const float* __restrict__ Out2 = nullptr) {   // SK: sk_per = segments, Out2 = the segment list (int4 each)
    constexpr int NF = 2 * NH;
    constexpr int HS = NH / 4;                 // waves that share an x pair (h slices of 4 tiles)
    constexpr int XPW = 8 / HS;                // x pairs per workgroup
    constexpr int YT = 2 * XPW;                // Y tiles per workgroup
    constexpr int SLOT = (NF + YT) * 1024;     // bytes of one k-step
    constexpr int NYD = 2 / HS;                // Y pieces this wave fetches per k-step
    constexpr int NFD = NF / 8;                // factor pieces this wave fetches per k-step
    constexpr int NDMA = NYD + NFD;            // = 3
    static_assert(NH == 4 || NH == 8, "wave tile is 2 x 4 tiles");
    static_assert(SLOT == LDS8_SLOT_BYTES && NDMA == 3, "slot layout");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    int bid = blockIdx.x;
    if (ctrl.mode != 0) {                                 // launch carries two control workgroups (dispatched first)
        if (bid < 2) {
            if constexpr (RCTRL > 0) {
                if (threadIdx.x < 256) ctrl_chain<RCTRL>(ctrl, smem, bid);
            }
            return;
        }
        bid -= 2;
    }
    if (stop && *stop) return;

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npair = XT >> 1;
    const int bps = (npair + XPW - 1) / XPW;
    const int nst_all = steps_per_split >> 1;  // stages of two k-steps (the host pads k-steps to an even count)
    int split = 0, xb = 0;
    // SK (un-split launches whose block count is not a whole number of rounds of the chip -- config 5's Y*A pass: 391 blocks on
    // 254 CUs = two rounds at 77 %): the launch's workgroups follow a host-built SEGMENT LIST, one (x block, first stage, stages,
    // slab) per workgroup, in dispatch order: first one whole block per CU, then the remaining blocks cut T ways in k, the pieces
    // ordered k-major -- workgroups that run at the same time stream the same k range of the factor, which is what keeps the
    // factor in the XCDs' L2s (a stream-K cut at arbitrary stages ran every workgroup at a different k: 1.05 ms instead of 0.83 ms,
    // the factor came from beyond L2).  Piece t of a cut block goes to slab t; the host adds slabs 1 .. T-1 into slab 0 for those blocks.
    int s_lo = 0, nst = nst_all, slab = 0;
    if constexpr (SK) {
        if (bid >= sk_per) return;                        // (sk_per = number of segments)
        const int4 sg = reinterpret_cast<const int4*>(Out2)[bid];
        xb = sg.x; s_lo = sg.y; nst = sg.z; slab = sg.w;
    } else if (xcd_xb > 0) {                              // XCD-aware map for split-K launches (see stream_gemm_kernel)
        const int g = bid & 7, j = bid >> 3;
        const int w = g * xcd_xb + j;
        if (j >= xcd_xb || w >= bps * nsplit) return;
        split = w / bps;
        xb = w % bps;
    } else {
        split = bid / bps;
        xb = bid % bps;
    }
    if (split >= nsplit) return;                          // workgroup-uniform
    const int xp = wib / HS, hs = wib % HS;
    const int voff = lane * 16;
    const int q16 = lane >> 4, r16 = lane & 15;
    const int lo16 = (q16 & 1) * SLOT + ((q16 >> 1) * 32 + r16) * 16;      // (+ 256: the upper 16 rows of a 32-wide tile)
    const int pair = xb * XPW + xp;
    const bool active = pair < npair;
    const int tile0 = (active ? pair : npair - 1) * 2;    // a wave without a pair streams the last one again (it still feeds the ring)

    const long long ks0 = (long long)split * steps_per_split + 2 * s_lo;
    const unsigned ybytes = (unsigned)(steps_per_split + PIPE_D) * 1024u;
    const unsigned fbytes = (unsigned)(steps_per_split + PIPE_D) * (NF * 1024u);
    __amdgpu_buffer_rsrc_t yr[NYD];
#pragma unroll
    for (int i = 0; i < NYD; ++i)
        yr[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(Yt + (((long long)(tile0 + (HS == 2 ? hs : i))) * KS + ks0) * 64), 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(Ft + ks0 * (NF * 64) + wib * (NFD * 64)), 0, fbytes, 0x00020000);
    // this wave's pieces of k-step `step` into slot `slot`
    auto dma_step = [&](int step, int slot) __attribute__((always_inline)) {
        unsigned char* base = smem + slot * SLOT;
#pragma unroll
        for (int q = 0; q < NFD; ++q) lds_dma_piece(fr, base + (wib * NFD + q) * 1024, voff, (step * NF + q) * 1024);
#pragma unroll
        for (int i = 0; i < NYD; ++i) {
            unsigned char* dst = base + (NF + xp * 2 + (HS == 2 ? hs : i)) * 1024;
            if constexpr (Y_AUX == 2) lds_dma_piece_nt(yr[i], dst, voff, step * 1024);
            else lds_dma_piece(yr[i], dst, voff, step * 1024);
        }
    };

    f32x4 acq[2][4][2][2];                     // [tx][th][hi][xi] 16 x 16 tiles
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acq[i][h][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    dma_step(0, 0); dma_step(1, 1); dma_step(2, 2); dma_step(3, 3);

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
            // both k-steps of the stage at once: 4 Y + 16 factor operands, 64 MFMAs.  The first operands cannot be requested
            // before the barrier (exposed LDS latency, once per stage); the factor's lo part is requested between the MFMAs of the
            // hi part, and the pieces of stage st + 2 go out between the MFMAs as well (one DMA costs its wave ~50-60 cycles of
            // issue, which the partner wave on the SIMD fills).
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
            // pinned order: the 12 reads of the hi part; per 2 MFMAs one read of the lo part; then one piece per 4 MFMAs
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // run-ahead pieces (slack of the tiled buffers) still target this LDS
    if (active) {
    // 16 x 16 tile (hi, xi): lane (q, c) holds column x = 16 xi + c, rows h = 16 hi + 4 q + t in register t.  The 32 x 32 tile
    // wants lane (half, c') = column c', register r = 4 g + t <-> row (r & 3) + 8 (r >> 2) + 4 half, i.e. half = q & 1,
    // g = 2 hi + (q >> 1).  With S0, S1 = register t of sub-tiles xi = 0, 1 as rows of 16 lanes [r0 r1 r2 r3]:
    // v_permlane16_swap -> [S0.r0 S1.r0 S0.r2 S1.r2], [S0.r1 S1.r1 S0.r3 S1.r3]; v_permlane32_swap of those two ->
    // [S0.r0 S1.r0 S0.r1 S1.r1] = register 4 (2 hi) + t and [S0.r2 S1.r2 S0.r3 S1.r3] = register 4 (2 hi + 1) + t.
    float* o = Out + (long long)(SK ? slab : split) * (NH * 32) * ldOut;
    float4* o4 = reinterpret_cast<float4*>(o);
    const int c = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            f32x16 a;
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float s0 = acq[i][h][hi][0][t], s1 = acq[i][h][hi][1][t];
                    const auto pa = __builtin_amdgcn_permlane16_swap(fbits(s0), fbits(s1), false, false);
                    const auto pb = __builtin_amdgcn_permlane32_swap(pa[0], pa[1], false, false);
                    a[4 * (2 * hi) + t] = bitsf(pb[0]);
                    a[4 * (2 * hi + 1) + t] = bitsf(pb[1]);
                }
            if (frag_out) {                    // fragment-major product (see stream_gemm_kernel)
                float4* t4 = o4 + (((long long)(tile0 + i) * NH + hs * 4 + h) * 64 + lane) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[q] = float4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            } else {
                const long long x = (long long)(tile0 + i) * 32 + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(long long)((hs * 4 + h) * 32 + rho(r, half)) * ldOut + x] = a[r];
            }
        }
    }                                                     // active
}

// Fix-up of a segment-list launch: Out[block] += sum_{t = 1 .. nslab-1} slab_t[block] for the blocks that were cut (list: block
// indices; block_floats contiguous floats per block in the fragment-major product; slab stride = total_floats).  Fixed order.
__global__ __launch_bounds__(256) void streamk_fixup_kernel(float* __restrict__ Out, int nslab, const int* __restrict__ list, int nlist,
                                                            long long block_floats, long long total_floats,
                                                            const int* __restrict__ stop) {
    if (stop && *stop) return;
    const long long n4 = block_floats >> 2;
    for (int b = blockIdx.y; b < nlist; b += gridDim.y) {
        const long long base = (long long)list[b] * block_floats;
        float4* o = reinterpret_cast<float4*>(Out + base);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4 && base + 4 * i < total_floats; i += (long long)gridDim.x * 256) {
            float4 a = o[i];
            for (int t = 1; t < nslab; ++t) {
                const float4 b2 = reinterpret_cast<const float4*>(Out + (long long)t * total_floats + base)[i];
                a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            }
            o[i] = a;
        }
    }
}

}  // namespace vbmf
