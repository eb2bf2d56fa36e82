// common.hpp -- shared device types and the fragment/tile index algebra (gfx950 only).
//
// Everything in this library is organised around ONE register layout: the 32x32 fp32 accumulator
// tile of the CDNA4 MFMA (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32):
//     lane = 32*half + c      holds column c, and in register r (0..15) row rho(r, half)
//     rho(r, half) = (r & 3) + 8*(r >> 2) + 4*half
// The tiled HBM formats of Y and of the factor operands use a k-order inside each MFMA k-step that
// equals rho, so a kernel that holds a result tile in accumulator registers can emit the NEXT
// kernel's MFMA operand fragments with plain 16-byte stores (no LDS transpose, no shuffles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vbmf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4v;

// streamed-once 16-byte load (global_load_dwordx4 ... nt): Y is never re-read within a pass
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
    const u32x4v v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(p));
    return __builtin_bit_cast(uint4, v);
}

enum : int { MODE_F32 = 0, MODE_BF16 = 1, MODE_BF16X2 = 2 };

// contraction elements consumed by one fragment (= one 16-byte load per lane)
template <int MODE> struct ModeTraits;
template <> struct ModeTraits<MODE_F32>    { static constexpr int KSTEP = 8;  static constexpr int NPART = 1; };
template <> struct ModeTraits<MODE_BF16>   { static constexpr int KSTEP = 16; static constexpr int NPART = 1; };
template <> struct ModeTraits<MODE_BF16X2> { static constexpr int KSTEP = 16; static constexpr int NPART = 2; };

__host__ __device__ inline int kstep_of(int mode) { return mode == MODE_F32 ? 8 : 16; }
__host__ __device__ inline int npart_of(int mode) { return mode == MODE_BF16X2 ? 2 : 1; }

__device__ __forceinline__ int rho(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// k offset (inside one k-step) of element e of the fragment held by lane-half `half`.
//   bf16 (16-wide step, 8 elements / lane): 8*(e>>2) + 4*half + (e&3)
//   f32  ( 8-wide step, 4 elements / lane): 4*half + e
__host__ __device__ __forceinline__ int kperm(int mode, int half, int e) {
    return mode == MODE_F32 ? (4 * half + e) : (8 * (e >> 2) + 4 * half + (e & 3));
}

// NOTE: never apply __builtin_bit_cast directly to an ext_vector element (v[i]): clang (ROCm 7.2)
// then reads element 0.  Always go through a by-value scalar, as these helpers do.
__device__ __forceinline__ unsigned fbits(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float bitsf(unsigned u) { return __builtin_bit_cast(float, u); }

__device__ __forceinline__ unsigned short f2bf(float f) {      // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) {
    return __builtin_bit_cast(float, ((unsigned)u) << 16);
}

// One accumulator register taken out of the AGPR file at THIS point of the program.  Left to the compiler, every accumulator
// value that is used by vector instructions after an MFMA loop is copied to a VGPR at the loop exit, all at once: with 256
// accumulator registers per wave that runs the whole kernel -- the MFMA loop's operand rings included -- out of registers
// (post_frag2_kernel: 1 484 bytes of scratch per lane -> 0, 256 -> 114 VGPRs).  The caller keeps >= 18 wait states between the
// last MFMA and the first read (acc_read_fence): hand-placed reads are outside the compiler's hazard bookkeeping.
__device__ __forceinline__ float acc_read(float a) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(a));
    return t;
}
__device__ __forceinline__ void acc_read_fence() { asm volatile("s_nop 15\n\ts_nop 15"); }

// PIPE_D: upper bound of the streaming kernel's ring depths (its run-ahead over-reads at most that many
// tiles: every tiled buffer carries PIPE_D tiles of slack).  The actual padding quanta are per rank class
// (host planner): x tiles to the per-wave tile count NXW = 8/NH, k-steps to the Y ring depth.
constexpr int PIPE_D = 12;

struct Dims {               // one streaming pass: Out[h][x] = sum_k F[k][h] * Y[k][x]
    int XT;                 // 32-wide x tiles, multiple of the per-wave tile count
    int KS;                 // k-steps, multiple of nsplit * (Y ring depth)
    int nsplit;             // split-K factor
    int steps_per_split;    // KS / nsplit, multiple of the Y ring depth
};

}  // namespace vbmf
