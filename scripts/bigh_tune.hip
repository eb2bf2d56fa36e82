// bigh_tune.hip -- developer harness for the H >= 128 shapes (NH = 4, 8) of the streaming contraction.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vbmatrixfactorization.jl_amd/csrc -I scripts scripts/bigh_tune.hip -o scripts/bigh_tune.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "common.hpp"
#include "stream_gemm.hpp"
#include "experiments/stream_gemm_lds.hpp"
using namespace vbmf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

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
    return ms / iters;
}

struct Shape { long long X, K; int H; const char* name; };

template <int NH, int NXW, int DY, int DF, int FDBG = 0>
static void run_variant(const Shape& sh, const uint4* Y, const uint4* F, float* O, size_t obytes, const char* tag) {
    const int XT = (int)(((sh.X + 31) / 32 + NXW - 1) / NXW * NXW);
    const int XG = XT / NXW, bps = (XG + 3) / 4;
    const int KS0 = (int)((sh.K + 15) / 16);
    const long long ld = (long long)XT * 32;
    CtrlArgs ca{}; ca.mode = 0;
    double best = 1e9; int bestns = 0;
    for (int ns : {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32}) {
        if (ns > 1 && bps * (ns + 1) <= 255 && bps * ns < 200) continue;
        const int blocks = bps * ns;
        if (blocks > 255 * 2 || (ns > 1 && blocks > 255) ) continue;
        if (blocks < 120 && ns < 32) { if (bps * (ns + 1) <= 255) continue; }
        int sps = (KS0 + ns - 1) / ns; sps = (sps + DY - 1) / DY * DY;
        const int KS = sps * ns;
        if ((size_t)ns * NH * 32 * ld * 4 > obytes) continue;
        const double yb = (double)sh.X * sh.K * 2.0;
        double ms = time_ms([&] { hipLaunchKernelGGL((stream_gemm_kernel<2, NH, NXW, DY, DF, 0, FDBG>), dim3(blocks), dim3(256), 0, 0, Y, F, O, XG, KS, sps, ns, ld, (const int*)nullptr, ca, 0); }, 5);
                printf("  %-28s ns=%2d blocks=%3d: %.3f ms  %.0f GB/s (Y only)\n", tag, ns, blocks, ms, yb / ms / 1e6);
        if (ms < best) { best = ms; bestns = ns; }
        if (false) {
            const int xx = (bps + 7) / 8, grid = 8 * xx * ns;
            if (grid <= 256) {
                ms = time_ms([&] { hipLaunchKernelGGL((stream_gemm_kernel<2, NH, NXW, DY, DF, 0>), dim3(grid), dim3(256), 0, 0, Y, F, O, XG, KS, sps, ns, ld, (const int*)nullptr, ca, xx); }, 5);
                printf("  %-28s ns=%2d blocks=%3d: %.3f ms  %.0f GB/s (Y only)  XCD-grouped (grid %d)\n", tag, ns, blocks, ms, yb / ms / 1e6, grid);
                if (ms < best) { best = ms; bestns = -ns; }
            }
        }
    }
    printf("  => %s best %.3f ms at ns=%d\n", tag, best, bestns);
}

__global__ void k_diff(const float* a, const float* b, size_t n, unsigned long long* cnt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += (__float_as_uint(a[i]) != __float_as_uint(b[i]));
    if (c) atomicAdd(cnt, c);
}

// LDS-shared-factor kernel vs the per-wave kernel at the same decomposition: time + bitwise comparison
template <int NH, int NXW, int DY, int DF, int DYL, int GF>
static void run_lds(const Shape& sh, const uint4* Y, const uint4* F, float* O, size_t obytes, int ns, const char* tag) {
    const int XT = (int)(((sh.X + 31) / 32 + NXW - 1) / NXW * NXW);
    const int XG = XT / NXW, bps = (XG + 3) / 4;
    const int KS0 = (int)((sh.K + 15) / 16);
    const long long ld = (long long)XT * 32;
    CtrlArgs ca{}; ca.mode = 0;
    int sps = (KS0 + ns - 1) / ns; sps = (sps + 11) / 12 * 12;
    const int KS = sps * ns, blocks = bps * ns;
    const size_t n = (size_t)ns * NH * 32 * ld;
    if (2 * n * 4 > obytes) { printf("  (output too large)\n"); return; }
    float* O2 = O + n;
    const size_t lds = (size_t)LDS_STAGES * 2 * NH * 1024;
    CK(hipFuncSetAttribute((const void*)stream_gemm_lds_kernel<2, NH, NXW, DYL, GF, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const double yb = (double)sh.X * sh.K * 2.0;
    CK(hipMemset(O, 0xff, n * 4)); CK(hipMemset(O2, 0xee, n * 4));
    double t0 = time_ms([&] { hipLaunchKernelGGL((stream_gemm_kernel<2, NH, NXW, DY, DF, 0>), dim3(blocks), dim3(256), 0, 0, Y, F, O, XG, KS, sps, ns, ld, (const int*)nullptr, ca, 0); }, 5);
    double t1 = time_ms([&] { hipLaunchKernelGGL((stream_gemm_lds_kernel<2, NH, NXW, DYL, GF, 0>), dim3(blocks), dim3(256), lds, 0, Y, F, O2, XG, KS, sps, ns, ld, (const int*)nullptr, ca, 0); }, 5);
    unsigned long long* cnt; CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8));
    hipLaunchKernelGGL(k_diff, dim3(1024), dim3(256), 0, 0, O, O2, n, cnt);
    unsigned long long h = 0; CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost)); CK(hipFree(cnt));
    printf("  %-22s ns=%2d blocks=%3d: per-wave %.3f ms (%.0f GB/s)   LDS-shared %.3f ms (%.0f GB/s)   differing words: %llu of %zu\n",
           tag, ns, blocks, t0, yb / t0 / 1e6, t1, yb / t1 / 1e6, h, n);
}

int main(int argc, char** argv) {
    Shape shapes[] = {{10000, 100000, 256, "cfg5 pass1 (x=M=10k, k=L=100k, H=256)"},
                      {100000, 10000, 256, "cfg5 pass2 (x=L=100k, k=M=10k, H=256)"},
                      {10000, 125000, 128, "cfg4/8 pass1 (x=M=10k, k=L=125k, H=128)"},
                      {125000, 10000, 128, "cfg4/8 pass2 (x=L=125k, k=M=10k, H=128)"}};
    const size_t nY = (size_t)(130000 / 32 + 8) * (size_t)(10000 / 16 + 64) * 64 * 2 + (1 << 20);
    const size_t nF = (size_t)(130000 / 16 + 64) * 16 * 64 + (1 << 16);
    const size_t obytes = (size_t)3 << 30;
    uint4 *Y, *F; float* O;
    CK(hipMalloc(&Y, nY * 16)); CK(hipMalloc(&F, nF * 16)); CK(hipMalloc(&O, obytes));
    fill_random(Y, nY); fill_random(F, nF);
    int which = argc > 1 ? atoi(argv[1]) : -1;
    for (int si = 0; si < 4; ++si) {
        if (which >= 0 && which != si) continue;
        const Shape& sh = shapes[si];
        printf("== %s\n", sh.name);
        if (sh.H == 256) {
            // round 2: ring depths again, now that the accumulators leave the AGPR file by explicit reads (no epilogue-induced
            // register pressure in the loop): profiles/r02_g_bigh_ring_depths.txt
            run_variant<8, 2, 2, 2>(sh, Y, F, O, obytes, "NH8 NXW2 DY2 DF2 (current)");
            run_variant<8, 2, 3, 3>(sh, Y, F, O, obytes, "NH8 NXW2 DY3 DF3");
            run_variant<8, 2, 4, 2>(sh, Y, F, O, obytes, "NH8 NXW2 DY4 DF2");
            run_variant<8, 2, 6, 3>(sh, Y, F, O, obytes, "NH8 NXW2 DY6 DF3");
            run_variant<8, 2, 6, 2>(sh, Y, F, O, obytes, "NH8 NXW2 DY6 DF2");
        } else {
            run_variant<4, 4, 4, 2>(sh, Y, F, O, obytes, "NH4 NXW4 DY4 DF2 (current)");
            run_variant<4, 4, 4, 4>(sh, Y, F, O, obytes, "NH4 NXW4 DY4 DF4");
            run_variant<4, 4, 6, 3>(sh, Y, F, O, obytes, "NH4 NXW4 DY6 DF3");
            run_variant<4, 4, 6, 2>(sh, Y, F, O, obytes, "NH4 NXW4 DY6 DF2");
            run_variant<4, 4, 3, 3>(sh, Y, F, O, obytes, "NH4 NXW4 DY3 DF3");
        }
    }
    return 0;
}
