// tile_kernels.hpp -- build the two MFMA-fragment-tiled device copies of Y, and read Y back.
//
// Tiled format ("Y tiles"): a fragment is the 16 bytes one lane feeds to one MFMA as the B operand;
// 64 lanes = 1 KiB = one tile of 32 x-columns by one k-step (16 k in bf16, 8 k in f32).  Tiles are
// stored [x-tile][k-step][lane], so a wave that walks the contraction reads one contiguous stream
// with fully coalesced 1 KiB wave-loads.  Two copies exist because the two passes contract over
// different indices (288 GB of HBM make the second copy free; neither pass re-reads the other's):
//     pass 1  P = Y'B :  x = column m, k = row l      (Y1)
//     pass 2  Q = Y A :  x = row l,    k = column m   (Y2)
#pragma once
#include "common.hpp"
#include "rng.hpp"

namespace vbmf {

// source functors: value of Y at (local row l, column m); rows/cols out of range are zero padding
struct ColMajorF64Src {
    const double* buf; long long ld; long long m0, mc;   // staging chunk: columns [m0, m0+mc)
    long long L, M;
    __device__ __forceinline__ double operator()(long long l, long long m) const {
        if (l >= L || m >= M) return 0.0;
        return buf[(m - m0) * ld + l];
    }
};
struct SynthSrc {
    SynthGen g; long long L, M, row_offset;
    __device__ __forceinline__ double operator()(long long l, long long m) const {
        if (l >= L || m >= M) return 0.0;
        return (double)g(l + row_offset, m);
    }
};

// One thread = one 16-byte fragment.  TRANSPOSED=false: x=m,k=l (Y1); true: x=l,k=m (Y2).
// Fragments in [xt0,xt1) x [ks0,ks1) are produced.  When sumsq != nullptr each block writes the fp64 sum of
// its squared stored values to sumsq[blockIdx.x] -- done on exactly one of the two copies.
template <int MODE, bool TRANSPOSED, class Src>
__global__ __launch_bounds__(256) void tile_y_kernel(uint4* __restrict__ out, Src src, int xt0, int xt1, int ks0,
                                                     int ks1, int KSpad, double* sumsq) {
    constexpr int KSTEP = (MODE == MODE_F32) ? 8 : 16;
    constexpr int NE = (MODE == MODE_F32) ? 4 : 8;
    const long long nks = ks1 - ks0;
    const long long total = (long long)(xt1 - xt0) * nks * 64;
    double acc = 0.0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        const long long q = t >> 6;
        const int ks = ks0 + (int)(q % nks);
        const int xt = xt0 + (int)(q / nks);
        const int c = lane & 31, half = lane >> 5;
        const long long x = (long long)xt * 32 + c;
        // values are rounded ONCE, from the source's fp64, to the device dtype (round-to-nearest-even)
        double v[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const long long k = (long long)ks * KSTEP + kperm(MODE, half, e);
            v[e] = TRANSPOSED ? src(x, k) : src(k, x);
        }
        uint4 o;
        if (MODE == MODE_F32) {
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { f[e] = (float)v[e]; acc += (double)f[e] * (double)f[e]; }
            o.x = fbits(f[0]); o.y = fbits(f[1]); o.z = fbits(f[2]); o.w = fbits(f[3]);
        } else {
            unsigned short b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 q = (__bf16)v[e];
                b[e] = __builtin_bit_cast(unsigned short, q);
                const float r = bf2f(b[e]);
                acc += (double)r * (double)r;
            }
            o.x = b[0] | ((unsigned)b[1] << 16); o.y = b[2] | ((unsigned)b[3] << 16);
            o.z = b[4] | ((unsigned)b[5] << 16); o.w = b[6] | ((unsigned)b[7] << 16);
        }
        out[((long long)xt * KSpad + ks) * 64 + lane] = o;
    }
    if (sumsq) {                                  // one partial per block, summed later in a fixed order
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        __shared__ double part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) sumsq[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
    }
}

// *dst += sum(partials[0..n)) in a fixed order (run-to-run reproducible ||Y||^2)
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partials, int n,
                                                           double* __restrict__ dst) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += partials[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *dst += sh[0];
}

// Decode the pass-2 copy (x = l, k = m) back to column-major fp64: out[(l-row0) + m*ld].
template <int MODE>
__global__ __launch_bounds__(256) void untile_y_kernel(const uint4* __restrict__ Y2, double* __restrict__ out,
                                                       long long ld, long long row0, long long nrows, long long M,
                                                       int KSpad) {
    constexpr int KSTEP = (MODE == MODE_F32) ? 8 : 16;
    const long long total = nrows * M;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long li = t % nrows, m = t / nrows;
        const long long l = row0 + li;
        const int xt = (int)(l >> 5), c = (int)(l & 31);
        const int ks = (int)(m / KSTEP), w = (int)(m % KSTEP);
        int half, e;
        if (MODE == MODE_F32) { half = w >> 2; e = w & 3; }
        else { half = (w >> 2) & 1; e = 4 * (w >> 3) + (w & 3); }
        const uint4 f = Y2[((long long)xt * KSpad + ks) * 64 + half * 32 + c];
        const unsigned wd[4] = {f.x, f.y, f.z, f.w};
        float v;
        if (MODE == MODE_F32) v = bitsf(wd[e]);
        else v = bf2f((unsigned short)((wd[e >> 1] >> (16 * (e & 1))) & 0xFFFFu));
        out[li + m * ld] = (double)v;
    }
}

}  // namespace vbmf
