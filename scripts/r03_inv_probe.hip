// r03_inv_probe.hip -- the blocked fp64 inverse of csrc/blk_inverse.hpp against a host fp64 Gauss-Jordan: accuracy, log det,
// latency of one inverse (in-kernel 100 MHz stamps) and throughput of many (one matrix per wave / per workgroup).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vbmatrixfactorization.jl_amd/csrc scripts/r03_inv_probe.hip -o scripts/r03_inv_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "blk_inverse.hpp"
using namespace vbmf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// one matrix per workgroup of NW waves; matrix m at A + m * n * n (row-major, order n), result to Out, log det to ld[m],
// stamps[m] = duration of the sweep in 10 ns ticks
template <int NB, int NW>
__global__ __launch_bounds__(NW * 64) void inv_wg_kernel(const double* __restrict__ A, double* __restrict__ Out, double* __restrict__ ldet,
                                                         unsigned long long* __restrict__ stamps, int n, int nmat) {
    extern __shared__ __attribute__((aligned(16))) double W[];
    constexpr int NP = 16 * NB, LD = NP + 2;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbu = (n + 15) / 16;
    for (int m = blockIdx.x; m < nmat; m += gridDim.x) {
        const double* a = A + (size_t)m * n * n;
        for (int t = threadIdx.x; t < NP * NP; t += NW * 64) {
            const int i = t / NP, j = t % NP;
            W[i * LD + j] = (i < n && j < n) ? a[i * n + j] : (i == j ? 1.0 : 0.0);
        }
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        PivAcc pv;
        blk_sweep<NB, NW>(W, LD, nbu, w, lane, pv);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        double* o = Out + (size_t)m * n * n;
        for (int t = threadIdx.x; t < n * n; t += NW * 64) {
            const int i = t / n, j = t % n;
            o[t] = -(i <= j ? W[i * LD + j] : W[j * LD + i]);       // upper blocks are authoritative... element-wise upper triangle here
        }
        if (threadIdx.x == 0) { ldet[m] = pv.bad ? -INFINITY : pv.logdet(); stamps[m] = t1 - t0; }
        __syncthreads();
    }
}

// one matrix per WAVE (NW waves per workgroup, each with its own LDS image): no barrier inside the sweep
template <int NB, int NW>
__global__ __launch_bounds__(NW * 64) void inv_wave_kernel(const double* __restrict__ A, double* __restrict__ Out, double* __restrict__ ldet,
                                                           unsigned long long* __restrict__ stamps, int n, int nmat) {
    extern __shared__ __attribute__((aligned(16))) double Wall[];
    constexpr int NP = 16 * NB, LD = NP + 2;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* W = Wall + (size_t)w * NP * LD;
    const int nbu = (n + 15) / 16;
    for (int m = blockIdx.x * NW + w; m < nmat; m += gridDim.x * NW) {
        const double* a = A + (size_t)m * n * n;
        for (int t = lane; t < NP * NP; t += 64) {
            const int i = t / NP, j = t % NP;
            W[i * LD + j] = (i < n && j < n) ? a[i * n + j] : (i == j ? 1.0 : 0.0);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        PivAcc pv;
        blk_sweep<NB, 1>(W, LD, nbu, 0, lane, pv);
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        double* o = Out + (size_t)m * n * n;
        for (int t = lane; t < n * n; t += 64) {
            const int i = t / n, j = t % n;
            o[t] = -(i <= j ? W[i * LD + j] : W[j * LD + i]);
        }
        if (lane == 0) { ldet[m] = pv.bad ? -INFINITY : pv.logdet(); stamps[m] = t1 - t0; }
    }
}

static void host_inverse(const std::vector<double>& a, int n, std::vector<double>& inv, double& logdet) {
    std::vector<long double> w(a.begin(), a.end());
    long double ld = 0;
    for (int k = 0; k < n; ++k) {
        const long double d = w[k * n + k]; ld += logl(d);
        const long double di = 1.0L / d;
        for (int j = 0; j < n; ++j) w[k * n + j] = (j == k ? 1.0L : w[k * n + j]) * di;
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const long double f = w[i * n + k];
            for (int j = 0; j < n; ++j) w[i * n + j] = (j == k ? 0.0L : w[i * n + j]) - f * w[k * n + j];
        }
    }
    inv.resize((size_t)n * n);
    for (int t = 0; t < n * n; ++t) inv[t] = (double)w[t];
    logdet = (double)ld;
}

template <int NB, int NW, bool PERWAVE>
static void run(int n, int nmat, int grid, const char* tag) {
    std::vector<double> A((size_t)nmat * n * n);
    unsigned s = 777u + n;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    // SPD with a wide spectrum: G G' / n + diag(10^(-3..3))
    std::vector<double> G((size_t)n * n);
    for (int m = 0; m < std::min(nmat, 4); ++m) {
        for (auto& g : G) g = rnd();
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double v = 0; for (int k = 0; k < n; ++k) v += G[i * n + k] * G[j * n + k];
                // matrices 0, 1: diag 1e-3 .. 1e3; matrices 2, 3: an ARD-like precision, diag entries from 1e-8 up to 1e10 in shuffled order
                const double dg = (m < 2) ? pow(10.0, -3.0 + 6.0 * i / std::max(1, n - 1)) : pow(10.0, -8.0 + 18.0 * ((i * 7) % n) / std::max(1, n - 1));
                A[(size_t)m * n * n + i * n + j] = v / n + (i == j ? dg : 0.0);
            }
    }
    for (int m = 4; m < nmat; ++m) std::copy(A.begin() + (size_t)(m % 4) * n * n, A.begin() + (size_t)(m % 4 + 1) * n * n, A.begin() + (size_t)m * n * n);
    double *dA, *dO, *dL; unsigned long long* dS;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dO, A.size() * 8)); CK(hipMalloc(&dL, nmat * 8)); CK(hipMalloc(&dS, nmat * 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    constexpr int NP = 16 * NB, LD = NP + 2;
    const size_t lds = (size_t)(PERWAVE ? NW : 1) * NP * LD * 8;
    auto launch = [&] {
        if constexpr (PERWAVE) hipLaunchKernelGGL((inv_wave_kernel<NB, NW>), dim3(grid), dim3(NW * 64), lds, 0, dA, dO, dL, dS, n, nmat);
        else hipLaunchKernelGGL((inv_wg_kernel<NB, NW>), dim3(grid), dim3(NW * 64), lds, 0, dA, dO, dL, dS, n, nmat);
    };
    if constexpr (PERWAVE) CK(hipFuncSetAttribute((const void*)inv_wave_kernel<NB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    else CK(hipFuncSetAttribute((const void*)inv_wg_kernel<NB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    launch(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    std::vector<double> O(A.size()), L(nmat); std::vector<unsigned long long> S(nmat);
    CK(hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(L.data(), dL, nmat * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(S.data(), dS, nmat * 8, hipMemcpyDeviceToHost));
    double worst = 0, worst_ld = 0;
    for (int m = 0; m < std::min(nmat, 4); ++m) {
        std::vector<double> a(A.begin() + (size_t)m * n * n, A.begin() + (size_t)(m + 1) * n * n), inv; double ld;
        host_inverse(a, n, inv, ld);
        double wm = 0;                                       // max_ij |err_ij| / sqrt(inv_ii inv_jj): scale-free for ARD-like spectra
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double d = fabs(O[(size_t)m * n * n + i * n + j] - inv[i * n + j]);
                wm = std::max(wm, d / sqrt(inv[i * n + i] * inv[j * n + j]));
            }
        worst = std::max(worst, wm);
        worst_ld = std::max(worst_ld, fabs(L[m] - ld) / std::max(1.0, fabs(ld)));
    }
    std::sort(S.begin(), S.end());
    printf("  %-40s n=%3d  %6d matrices, grid %4d: rel.err %.2e  logdet err %.2e   sweep %.2f us (median of in-kernel stamps, min %.2f)   kernel %.3f ms = %.3f us per matrix\n",
           tag, n, nmat, grid, worst, worst_ld, S[nmat / 2] * 0.01, S[0] * 0.01, ms, ms * 1e3 / nmat);
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dO)); CK(hipFree(dL)); CK(hipFree(dS));
}

int main() {
    printf("== one matrix, one workgroup (latency)\n");
    run<1, 1, false>(10, 1, 1, "1 wave");
    run<2, 1, false>(32, 1, 1, "1 wave");
    run<2, 4, false>(32, 1, 1, "4 waves");
    run<4, 1, false>(64, 1, 1, "1 wave");
    run<4, 4, false>(64, 1, 1, "4 waves");
    run<4, 4, false>(50, 1, 1, "4 waves (n = 50, padded to 64)");
    run<8, 1, false>(128, 1, 1, "1 wave");
    run<8, 4, false>(128, 1, 1, "4 waves");
    run<8, 16, false>(128, 1, 1, "16 waves");
    run<8, 4, false>(100, 1, 1, "4 waves (n = 100, padded)");
    printf("== 10 000 matrices (throughput)\n");
    run<4, 4, true>(64, 10000, 256, "one matrix per wave, 4 waves per WG");
    run<4, 4, true>(64, 10000, 512, "one matrix per wave, 4 waves per WG");
    run<4, 4, false>(64, 10000, 1024, "one matrix per 4-wave WG");
    run<2, 8, true>(32, 10000, 512, "one matrix per wave, 8 waves per WG");
    run<1, 8, true>(10, 10000, 512, "one matrix per wave, 8 waves per WG");
    return 0;
}
