// r03_inv256_probe.hip -- inv256_blk (csrc/blk_inverse.hpp: the 256 x 256 blocked sweep with the matrix in registers, 1024 threads) against a host
// long-double Gauss-Jordan: accuracy (scale-free), pivots / log det, latency (in-kernel 100 MHz stamps).  Orders 130, 200, 256 (identity-padded).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vbmatrixfactorization.jl_amd/csrc scripts/r03_inv256_probe.hip -o scripts/r03_inv256_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "blk_inverse.hpp"
using namespace vbmf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(1024) void inv256_kernel(const double* __restrict__ Kg, double* __restrict__ Out, double* __restrict__ piv_out,
                                                      unsigned long long* __restrict__ stamp) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* pivs = lds + INV256_LDS_DOUBLES;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    inv256_blk(Kg, Out, lds, pivs);
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x < 256) piv_out[threadIdx.x] = pivs[threadIdx.x];
    if (threadIdx.x == 0) *stamp = t1 - t0;
}

static void host_inverse(const std::vector<double>& a, int n, std::vector<double>& inv, std::vector<double>& piv) {
    std::vector<long double> w(a.begin(), a.end());
    piv.resize(n);
    for (int k = 0; k < n; ++k) {
        const long double d = w[k * n + k]; piv[k] = (double)d;
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
}

int main() {
    const int N = 256;
    for (int n : {130, 200, 256})
        for (int kind = 0; kind < 2; ++kind) {
            std::vector<double> A((size_t)N * N, 0.0), G((size_t)n * n);
            unsigned s = 4242u + n + kind;
            auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
            for (auto& g : G) g = rnd();
            for (int i = 0; i < N; ++i) A[(size_t)i * N + i] = 1.0;                  // identity padding
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double v = 0; for (int k = 0; k < n; ++k) v += G[i * n + k] * G[j * n + k];
                    const double dg = kind == 0 ? pow(10.0, -3.0 + 6.0 * i / (n - 1.0)) : pow(10.0, -8.0 + 18.0 * ((i * 7) % n) / (n - 1.0));   // ARD-like: 1e-8 .. 1e10
                    A[(size_t)i * N + j] = v / n + (i == j ? dg : 0.0);
                }
            double *dA, *dO, *dP; unsigned long long* dS;
            CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dO, A.size() * 8)); CK(hipMalloc(&dP, 256 * 8)); CK(hipMalloc(&dS, 8));
            CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
            const size_t lds = (size_t)(INV256_LDS_DOUBLES + 256) * 8;
            CK(hipFuncSetAttribute((const void*)inv256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(inv256_kernel, dim3(1), dim3(1024), lds, 0, dA, dO, dP, dS);
            CK(hipDeviceSynchronize()); CK(hipGetLastError());
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(inv256_kernel, dim3(1), dim3(1024), lds, 0, dA, dO, dP, dS);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            std::vector<double> O(A.size()), P(256); unsigned long long st;
            CK(hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(P.data(), dP, 256 * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&st, dS, 8, hipMemcpyDeviceToHost));
            std::vector<double> inv, piv; host_inverse(A, N, inv, piv);
            double worst = 0, ld_dev = 0, ld_ref = 0, sym = 0;
            for (int i = 0; i < N; ++i) { ld_dev += log(P[i]); ld_ref += log(piv[i]); }
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    worst = std::max(worst, fabs(O[(size_t)i * N + j] - inv[(size_t)i * N + j]) / sqrt(inv[(size_t)i * N + i] * inv[(size_t)j * N + j]));
                    sym = std::max(sym, fabs(O[(size_t)i * N + j] - O[(size_t)j * N + i]));
                }
            printf("n=%3d %-28s rel.err %.2e  asym %.1e  logdet err %.2e   sweep %.1f us (in-kernel)   kernel %.1f us\n", n,
                   kind == 0 ? "diag 1e-3..1e3" : "ARD-like diag 1e-8..1e10", worst, sym, fabs(ld_dev - ld_ref) / std::max(1.0, fabs(ld_ref)), st * 0.01, ms * 1e3);
            fflush(stdout);
            CK(hipFree(dA)); CK(hipFree(dO)); CK(hipFree(dP)); CK(hipFree(dS));
        }
    return 0;
}
