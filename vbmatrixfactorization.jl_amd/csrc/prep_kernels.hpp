// prep_kernels.hpp -- scaleY / preprocess (src/util.jl:36-54, 73-86) as device pre-passes over a resident fp64 Y
// (column-major, leading dimension ld): row mean, row variance (two-pass, n-1), the scaled absolute row sums the
// row filter looks at, and a tiling source that applies (Y - mu)/den * lambda to the kept rows on the fly.
#pragma once
#include "common.hpp"

namespace vbmf {

constexpr int PREP_CHUNKS = 16;    // column chunks: partial row sums [PREP_CHUNKS][L], folded in fixed order

// what = 0: sum y;  1: sum (y - mu)^2;  2: sum |s|, s = zero_small(y - mu) / den   (src/util.jl:38-39, 47-51, 75)
__global__ __launch_bounds__(256) void prep_row_partial_kernel(const double* __restrict__ Y, long long L, long long M,
                                                               long long ld, int what, const double* __restrict__ mu,
                                                               const double* __restrict__ den,
                                                               double* __restrict__ part) {
    const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const long long cols = (M + PREP_CHUNKS - 1) / PREP_CHUNKS;
    const long long m0 = (long long)blockIdx.y * cols, m1 = m0 + cols < M ? m0 + cols : M;
    const double mean = what ? mu[l] : 0.0, dn = what == 2 ? den[l] : 1.0;
    double a0 = 0.0, a1 = 0.0;
    long long m = m0;
    for (; m + 1 < m1; m += 2) {
        double y0 = Y[m * ld + l], y1 = Y[(m + 1) * ld + l];
        if (what == 1) { y0 -= mean; y1 -= mean; y0 *= y0; y1 *= y1; }
        if (what == 2) {
            y0 -= mean; y1 -= mean;
            y0 = fabs(y0) <= 1e-8 ? 0.0 : fabs(y0 / dn);
            y1 = fabs(y1) <= 1e-8 ? 0.0 : fabs(y1 / dn);
        }
        a0 += y0; a1 += y1;
    }
    if (m < m1) {
        double y0 = Y[m * ld + l];
        if (what == 1) { y0 -= mean; y0 *= y0; }
        if (what == 2) { y0 -= mean; y0 = fabs(y0) <= 1e-8 ? 0.0 : fabs(y0 / dn); }
        a0 += y0;
    }
    part[(long long)blockIdx.y * L + l] = a0 + a1;
}

// what = 0: mu = sum / M;  1: den = sqrt(var), var = sum/(M-1), |var| <= 1e-15 -> 1 (src/util.jl:44-47);  2: keep flag
__global__ __launch_bounds__(256) void prep_row_fold_kernel(const double* __restrict__ part, long long L, long long M,
                                                            int what, double* __restrict__ out,
                                                            unsigned char* __restrict__ keep) {
    const long long l = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    double s = 0.0;
    for (int q = 0; q < PREP_CHUNKS; ++q) s += part[(long long)q * L + l];
    if (what == 0) out[l] = s / (double)M;
    else if (what == 1) {
        double v = s / (double)(M - 1);            // Julia var(): corrected; NaN for M = 1, like the reference
        if (fabs(v) <= 1e-15) v = 1.0;
        out[l] = sqrt(v);
    } else {
        out[l] = s;
        keep[l] = s >= 1e-5 ? 1 : 0;               // src/util.jl:78
    }
}

// tiling source: local row l of the context = kept row rows[row_offset + l] of the resident matrix
struct PrepSrc {
    const double* buf; long long ld;
    const long long* rows; const double* mu; const double* den;
    double lambda; long long L, M, row_offset;
    __device__ __forceinline__ double operator()(long long l, long long m) const {
        if (l >= L || m >= M) return 0.0;
        const long long r = rows[row_offset + l];
        const double nom = buf[m * ld + r] - mu[r];
        return fabs(nom) <= 1e-8 ? 0.0 : lambda * (nom / den[r]);
    }
};

}  // namespace vbmf
