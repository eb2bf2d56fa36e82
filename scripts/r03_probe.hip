// r03_probe.hip -- round-3 harness: where is the ceiling of the H >= 128 streaming passes?
//   (A) bare bf16 MFMA loops on random operands held in registers, both shapes (32x32x16, 16x16x32), one and two waves per
//       SIMD, every CU busy: TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime) -- what the chip sustains under its
//       power management when nothing but MFMAs issue;
//   (B) the library's stream_gemm_kernel at config 5's and config 4's pass shapes (hi + lo factor parts), and the variants of
//       scripts/experiments (wave tile 4 x 4 instead of 2 x 8; the round-1 LDS-shared factor), timed in one process and
//       compared bit for bit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vbmatrixfactorization.jl_amd/csrc -I scripts scripts/r03_probe.hip -o scripts/r03_probe.bin
// Run:   scripts/r03_probe.bin [bare|stream|all]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "common.hpp"
#include "stream_gemm.hpp"
#include "experiments/stream_gemm_lds.hpp"
#include "experiments/stream_gemm_v3.hpp"
using namespace vbmf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4v;

static void fill_random(uint4* d, size_t n) {
    const size_t chunk = std::min<size_t>(n, (size_t)1 << 22);
    std::vector<unsigned> h(chunk * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        unsigned lo = ((s >> 3) & 0x807F) | (0x3F00 - (((s >> 20) & 1) << 8));
        unsigned hi = ((s >> 11) & 0x807F) | (0x3F00 - (((s >> 21) & 1) << 8));
        h[i] = lo | (hi << 16);
    }
    CK(hipMemcpy(d, h.data(), chunk * 16, hipMemcpyHostToDevice));
    for (size_t off = chunk; off < n; off += chunk)
        CK(hipMemcpy(d + off, d, std::min(chunk, n - off) * 16, hipMemcpyDeviceToDevice));
}

template <class F>
static double time_ms(F&& launch, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / iters;
}

// ---- (A) bare MFMA loops -------------------------------------------------------------------------------------------
// SHAPE 0: v_mfma_f32_32x32x16_bf16, NT accumulator tiles of 16 registers; SHAPE 1: v_mfma_f32_16x16x32_bf16, 4 NT tiles of 4.
// Same accumulator footprint and the same FLOPs per outer iteration for both shapes; 4 x 4 distinct operand pairs.
template <int SHAPE, int WAVES, int NT>
__global__ __launch_bounds__(WAVES * 64) void bare_kernel(const uint4* __restrict__ opnd, float* __restrict__ out, int iters,
                                                          unsigned long long* __restrict__ stamps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, opnd[((size_t)(blockIdx.x * WAVES + wave) * 8 + i) * 64 + lane]);
        b[i] = __builtin_bit_cast(bf16x8, opnd[((size_t)(blockIdx.x * WAVES + wave) * 8 + 4 + i) * 64 + lane]);
    }
    float sum = 0.f;
    unsigned long long t0, r0, t1, r1;
    if constexpr (SHAPE == 0) {
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + rep) & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[j][r];
    } else {
        f32x4v acc[NT * 4];
#pragma unroll
        for (int j = 0; j < NT * 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NT * 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j >> 2) & 3], acc[j], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int j = 0; j < NT * 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += acc[j][r];
    }
    out[(size_t)blockIdx.x * WAVES * 64 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int WAVES, int NT>
static void run_bare(const uint4* opnd, float* out, unsigned long long* stamps, const char* tag, int blocks) {
    // per outer iteration and wave: 2 * NT MFMAs of 32x32x16 = 4 * NT of 16x16x32 = NT * 65536 flop
    const int iters = 120000 / NT * 8;
    const double flop = (double)blocks * WAVES * iters * NT * 65536.0;
    auto launch = [&] { hipLaunchKernelGGL((bare_kernel<SHAPE, WAVES, NT>), dim3(blocks), dim3(WAVES * 64), 0, 0, opnd, out, iters, stamps); };
    for (int rep = 0; rep < 3; ++rep) launch();                 // ~ settle the clocks under this load
    CK(hipDeviceSynchronize());
    const double ms = time_ms(launch, 3);
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk(blocks);
    for (int i = 0; i < blocks; ++i) clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;   // GHz (realtime ticks at 100 MHz)
    std::sort(clk.begin(), clk.end());
    const double cyc_per_mfma = (double)h[0] / ((double)iters * (SHAPE == 0 ? 2 * NT : 4 * NT));
    printf("  %-44s %7.2f ms  %7.1f TFLOP/s   in-kernel clock median %.3f GHz (min %.3f max %.3f)   %.1f cycles per MFMA and wave\n",
           tag, ms, flop / ms / 1e9, clk[blocks / 2], clk.front(), clk.back(), cyc_per_mfma);
    fflush(stdout);
}

// ---- (B) streaming variants ----------------------------------------------------------------------------------------
__global__ void k_diff(const float* a, const float* b, size_t n, unsigned long long* cnt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += (__float_as_uint(a[i]) != __float_as_uint(b[i]));
    if (c) atomicAdd(cnt, c);
}
static unsigned long long diff_words(const float* a, const float* b, size_t n) {
    unsigned long long* cnt; CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8));
    hipLaunchKernelGGL(k_diff, dim3(1024), dim3(256), 0, 0, a, b, n, cnt);
    unsigned long long h = 0; CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost)); CK(hipFree(cnt));
    return h;
}

__global__ void k_maxdiff(const float* a, const float* b, size_t n, unsigned* out) {     // out[0] = max |a - b|, out[1] = max |a| (as bits of non-negative floats)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float d = 0.f, m = 0.f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { d = fmaxf(d, fabsf(a[i] - b[i])); m = fmaxf(m, fabsf(a[i])); }
    atomicMax(out, __float_as_uint(d)); atomicMax(out + 1, __float_as_uint(m));
}
static void max_diff(const float* a, const float* b, size_t n, float* d, float* m) {
    unsigned* o; CK(hipMalloc(&o, 8)); CK(hipMemset(o, 0, 8));
    hipLaunchKernelGGL(k_maxdiff, dim3(1024), dim3(256), 0, 0, a, b, n, o);
    unsigned h[2]; CK(hipMemcpy(h, o, 8, hipMemcpyDeviceToHost)); CK(hipFree(o));
    memcpy(d, &h[0], 4); memcpy(m, &h[1], 4);
}

struct Shape { long long X, K; int H; int ns; const char* name; };

// current library kernel (fragment-major output) vs the re-shaped wave tile, same decomposition, interleaved rounds
template <int NH, int NXW, int DY, int DF, int PNHW, int PNXW, int PDY, int PDF>
static void run_stream(const Shape& sh, const uint4* Y, const uint4* F, float* O, size_t obytes) {
    constexpr int XQ = NXW > PNXW ? NXW : PNXW;
    const int XT = (int)(((sh.X + 31) / 32 + XQ - 1) / XQ * XQ);
    const int ns = sh.ns;
    const int KS0 = (int)((sh.K + 15) / 16);
    int sps = (KS0 + ns - 1) / ns; sps = (sps + 11) / 12 * 12;
    const int KS = sps * ns;
    const long long ld = (long long)XT * 32;
    const size_t n = (size_t)ns * NH * 32 * ld;
    if (2 * n * 4 > obytes) { printf("  (output too large)\n"); return; }
    float* O2 = O + n;
    CtrlArgs ca{}; ca.mode = 0;
    EpiArgs ea{}; ea.frag_out = 1;
    const int XG = XT / NXW, bps = (XG + 3) / 4;
    constexpr int HS = NH / PNHW, GPW = 4 / HS;
    const int PXG = XT / PNXW, pbps = (PXG + GPW - 1) / GPW;
    const double flop = 2.0 * sh.X * sh.K * sh.H * 2.0;     // hi + lo factor parts
    const double yb = (double)sh.X * sh.K * 2.0;
    auto cur = [&] { hipLaunchKernelGGL((stream_gemm_kernel<2, NH, NXW, DY, DF, 0>), dim3(bps * ns), dim3(256), 0, 0, Y, F, O, XG, KS, sps, ns, ld, (const int*)nullptr, ca, 0, ea); };
    auto pair = [&] { hipLaunchKernelGGL((stream_pair_kernel<NH, PNHW, PNXW, PDY, PDF, 0>), dim3(pbps * ns), dim3(256), 0, 0, Y, F, O2, PXG, KS, sps, ns, ld); };
    CK(hipMemset(O, 0xff, n * 4)); CK(hipMemset(O2, 0xee, n * 4));
    double tc[3], tp[3];
    for (int r = 0; r < 3; ++r) { tc[r] = time_ms(cur, 6); tp[r] = time_ms(pair, 6); }
    std::sort(tc, tc + 3); std::sort(tp, tp + 3);
    const unsigned long long dw = diff_words(O, O2, n);
    printf("  library  NXW%d x NH%d DY%d DF%d  blocks %4d: median %.3f ms (min %.3f)  %.0f GB/s of Y  %.0f TFLOP/s (hi+lo)\n", NXW, NH, DY, DF, bps * ns, tc[1], tc[0], yb / tc[1] / 1e6, flop / tc[1] / 1e9);
    printf("  pair     NXW%d x NHW%d DY%d DF%d blocks %4d: median %.3f ms (min %.3f)  %.0f GB/s of Y  %.0f TFLOP/s (hi+lo)   differing words %llu of %zu\n", PNXW, PNHW, PDY, PDF, pbps * ns, tp[1], tp[0],
           yb / tp[1] / 1e6, flop / tp[1] / 1e9, dw, n);
    fflush(stdout);
}

template <int NH, int NXW, int DYL, int GF>
static void run_lds(const Shape& sh, const uint4* Y, const uint4* F, float* O, size_t obytes) {
    const int XT = (int)(((sh.X + 31) / 32 + NXW - 1) / NXW * NXW);
    const int XG = XT / NXW, bps = (XG + 3) / 4, ns = sh.ns;
    const int KS0 = (int)((sh.K + 15) / 16);
    const long long ld = (long long)XT * 32;
    CtrlArgs ca{}; ca.mode = 0;
    int sps = (KS0 + ns - 1) / ns; sps = (sps + 11) / 12 * 12;
    const int KS = sps * ns, blocks = bps * ns;
    const size_t n = (size_t)ns * NH * 32 * ld;
    if (n * 4 > obytes) { printf("  (output too large)\n"); return; }
    const size_t lds = (size_t)LDS_STAGES * 2 * NH * 1024;
    CK(hipFuncSetAttribute((const void*)stream_gemm_lds_kernel<2, NH, NXW, DYL, GF, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const double flop = 2.0 * sh.X * sh.K * sh.H * 2.0, yb = (double)sh.X * sh.K * 2.0;
    double t[3];
    for (int r = 0; r < 3; ++r)
        t[r] = time_ms([&] { hipLaunchKernelGGL((stream_gemm_lds_kernel<2, NH, NXW, DYL, GF, 0>), dim3(blocks), dim3(256), lds, 0, Y, F, O, XG, KS, sps, ns, ld, (const int*)nullptr, ca, 0); }, 6);
    std::sort(t, t + 3);
    printf("  lds(r1)  NXW%d x NH%d DY%d GF%d  blocks %4d: median %.3f ms (min %.3f)  %.0f GB/s of Y  %.0f TFLOP/s (hi+lo)\n", NXW, NH, DYL, GF, blocks, t[1], t[0], yb / t[1] / 1e6, flop / t[1] / 1e9);
    fflush(stdout);
}

template __global__ void vbmf::stream_lds8x_kernel<8, 0, 2>(const uint4*, const uint4*, float*, int, int, int, int, long long, int);
template __global__ void vbmf::stream_lds8x_kernel<4, 0, 2>(const uint4*, const uint4*, float*, int, int, int, int, long long, int);
template __global__ void vbmf::stream_lds8x_kernel<8, 1, 2>(const uint4*, const uint4*, float*, int, int, int, int, long long, int);
template __global__ void vbmf::stream_lds8x_kernel<4, 1, 2>(const uint4*, const uint4*, float*, int, int, int, int, long long, int);

// the 8-wave LDS-DMA kernel against the library kernel at the same decomposition, interleaved rounds, bit-compared
template <int NH, int NXW, int DY, int DF, int SHAPE>
static void run_lds8(const Shape& sh, const uint4* Y, const uint4* F, float* O, size_t obytes) {
    const int XT = (int)(((sh.X + 31) / 32 + NXW - 1) / NXW * NXW);
    const int ns = sh.ns;
    const int KS0 = (int)((sh.K + 15) / 16);
    int sps = (KS0 + ns - 1) / ns; sps = (sps + 11) / 12 * 12;
    const int KS = sps * ns;
    const long long ld = (long long)XT * 32;
    const size_t n = (size_t)ns * NH * 32 * ld;
    if (2 * n * 4 > obytes) { printf("  (output too large)\n"); return; }
    float* O2 = O + n;
    CtrlArgs ca{}; ca.mode = 0;
    EpiArgs ea{}; ea.frag_out = 1;
    const int XG = XT / NXW, bps = (XG + 3) / 4;
    constexpr int XPW = 8 / (NH / 4);
    const int lbps = (XT / 2 + XPW - 1) / XPW;
    const int lds = 6 * 24 * 1024;
    CK(hipFuncSetAttribute((const void*)stream_lds8x_kernel<NH, SHAPE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const double flop = 2.0 * sh.X * sh.K * sh.H * 2.0, yb = (double)sh.X * sh.K * 2.0;
    auto cur = [&] { hipLaunchKernelGGL((stream_gemm_kernel<2, NH, NXW, DY, DF, 0>), dim3(bps * ns), dim3(256), 0, 0, Y, F, O, XG, KS, sps, ns, ld, (const int*)nullptr, ca, 0, ea); };
    auto l8 = [&] { hipLaunchKernelGGL((stream_lds8x_kernel<NH, SHAPE, 2>), dim3(lbps * ns), dim3(512), lds, 0, Y, F, O2, XT, KS, sps, ns, ld, 1); };
    CK(hipMemset(O, 0xff, n * 4)); CK(hipMemset(O2, 0xee, n * 4));
    double tc[3], tp[3];
    for (int r = 0; r < 3; ++r) { tc[r] = time_ms(cur, 6); tp[r] = time_ms(l8, 6); }
    std::sort(tc, tc + 3); std::sort(tp, tp + 3);
    const unsigned long long dw = diff_words(O, O2, n);
    float md = 0.f, mm = 0.f; max_diff(O, O2, n, &md, &mm);
    printf("  library  NXW%d x NH%d DY%d DF%d  blocks %4d: median %.3f ms (min %.3f)  %.0f GB/s of Y  %.0f TFLOP/s (hi+lo)\n", NXW, NH, DY, DF, bps * ns, tc[1], tc[0], yb / tc[1] / 1e6, flop / tc[1] / 1e9);
    printf("  lds8     shape %d, 8 waves x 2x4 tiles blocks %4d: median %.3f ms (min %.3f)  %.0f GB/s of Y  %.0f TFLOP/s (hi+lo)   differing words %llu of %zu, max |diff| %.3g of max |value| %.3g\n", SHAPE, lbps * ns, tp[1], tp[0],
           yb / tp[1] / 1e6, flop / tp[1] / 1e9, dw, n, md, mm);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    const bool do_bare = !strcmp(what, "all") || !strcmp(what, "bare");
    const bool do_stream = !strcmp(what, "all") || !strcmp(what, "stream");
    const bool do_lds8 = !strcmp(what, "all") || !strcmp(what, "lds8");
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    if (do_bare) {
        uint4* opnd; float* out; unsigned long long* stamps;
        CK(hipMalloc(&opnd, (size_t)256 * 8 * 8 * 64 * 16)); CK(hipMalloc(&out, (size_t)256 * 512 * 4)); CK(hipMalloc(&stamps, 2 * 256 * 8));
        fill_random(opnd, (size_t)256 * 8 * 8 * 64);
        printf("== (A) bare bf16 MFMA loops, random operands in registers, 256 workgroups (one per CU)\n");
        for (int rep = 0; rep < 2; ++rep) {
            run_bare<0, 4, 16>(opnd, out, stamps, "32x32x16, 1 wave/SIMD, 16 acc tiles", 256);
            run_bare<1, 4, 16>(opnd, out, stamps, "16x16x32, 1 wave/SIMD, 64 acc tiles of 4", 256);
            run_bare<0, 8, 8>(opnd, out, stamps, "32x32x16, 2 waves/SIMD, 8 acc tiles each", 256);
            run_bare<1, 8, 8>(opnd, out, stamps, "16x16x32, 2 waves/SIMD, 32 acc tiles of 4 each", 256);
        }
        CK(hipMemset(opnd, 0, (size_t)256 * 8 * 8 * 64 * 16));
        printf("== the same on all-zero operands\n");
        run_bare<0, 4, 16>(opnd, out, stamps, "32x32x16, 1 wave/SIMD, zeros", 256);
        run_bare<1, 4, 16>(opnd, out, stamps, "16x16x32, 1 wave/SIMD, zeros", 256);
        CK(hipFree(opnd)); CK(hipFree(out)); CK(hipFree(stamps));
    }
    if (do_stream || do_lds8) {
        Shape shapes[] = {{10000, 100000, 256, 12, "cfg5 pass1 (x=M=10k, k=L=100k, H=256), 12 k-slices"},
                          {100000, 10000, 256, 1, "cfg5 pass2 (x=L=100k, k=M=10k, H=256)"},
                          {10000, 125000, 128, 12, "cfg4/8 pass1 (x=M=10k, k=L=125k, H=128), 12 k-slices"},
                          {125000, 10000, 128, 1, "cfg4/8 pass2 (x=L=125k, k=M=10k, H=128)"}};
        const size_t nY = (size_t)(130000 / 32 + 8) * (size_t)(10000 / 16 + 64) * 64 * 2 + (1 << 20);
        const size_t nF = (size_t)(130000 / 16 + 64) * 16 * 64 + (1 << 16);
        const size_t obytes = (size_t)3 << 30;
        uint4 *Y, *F; float* O;
        CK(hipMalloc(&Y, nY * 16)); CK(hipMalloc(&F, nF * 16)); CK(hipMalloc(&O, obytes));
        fill_random(Y, nY); fill_random(F, nF);
        for (int si = 0; si < 4; ++si) {
            const Shape& sh = shapes[si];
            printf("== (B) %s\n", sh.name);
            if (do_lds8) {
                if (sh.H == 256) { run_lds8<8, 2, 2, 2, 0>(sh, Y, F, O, obytes); run_lds8<8, 2, 2, 2, 1>(sh, Y, F, O, obytes); }
                else { run_lds8<4, 4, 4, 2, 0>(sh, Y, F, O, obytes); run_lds8<4, 4, 4, 2, 1>(sh, Y, F, O, obytes); }
            }
            if (!do_stream) continue;
            if (sh.H == 256) {
                run_stream<8, 2, 2, 2, 4, 4, 2, 2>(sh, Y, F, O, obytes);
                run_stream<8, 2, 2, 2, 4, 4, 4, 2>(sh, Y, F, O, obytes);
                run_stream<8, 2, 2, 2, 4, 4, 3, 3>(sh, Y, F, O, obytes);
                run_lds<8, 2, 4, 2>(sh, Y, F, O, obytes);
            } else {
                run_stream<4, 4, 4, 2, 4, 4, 4, 2>(sh, Y, F, O, obytes);       // (identical tiling: the pair kernel's own overhead)
                run_stream<4, 4, 4, 2, 2, 8, 2, 2>(sh, Y, F, O, obytes);       // 8 x 2 tiles: 12 loads per 32 MFMAs as well, other aspect
                run_lds<4, 4, 4, 2>(sh, Y, F, O, obytes);
            }
        }
    }
    return 0;
}
