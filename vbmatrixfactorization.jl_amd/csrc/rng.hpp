// rng.hpp -- Philox4x32-10 counter-based generator + Box-Muller, device side.
// Used only by the synthetic-data generator (vbmf_set_Y_synthetic): every element of Y is a pure
// function of (seed, global row, column), so the matrix is identical for any row-sharding.
#pragma once
#include "common.hpp"

namespace vbmf {

struct u32x4 { unsigned x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                               unsigned k1) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return {c0, c1, c2, c3};
}

__device__ __forceinline__ float normal_from(unsigned a, unsigned b) {
    const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;   // (0,1)
    const float u2 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// toy_matrix of examples/toy_data.jl:7-18: Y[l,m] = Bstar[l, col(m)] + std * N(0,1)
struct SynthGen {
    unsigned long long seed;
    long long Hstar, M;
    float std;
    __device__ __forceinline__ float operator()(long long lg, long long m) const {
        const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
        const u32x4 a = philox4x32_10((unsigned)m, (unsigned)(m >> 32), 2u, 0u, k0, k1);
        const long long col = (long long)(a.x % (unsigned)Hstar);
        const unsigned long long bi = (unsigned long long)lg * (unsigned long long)Hstar + (unsigned long long)col;
        const u32x4 b = philox4x32_10((unsigned)bi, (unsigned)(bi >> 32), 1u, 0u, k0, k1);
        const unsigned long long ni = (unsigned long long)lg * (unsigned long long)M + (unsigned long long)m;
        const u32x4 n = philox4x32_10((unsigned)ni, (unsigned)(ni >> 32), 3u, 0u, k0, k1);
        return normal_from(b.x, b.y) + std * normal_from(n.x, n.y);
    }
};

}  // namespace vbmf
