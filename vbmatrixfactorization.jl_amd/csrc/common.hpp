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

// One 32 x 32 accumulator tile (16 registers) taken out of the AGPR file at THIS point of the program.  Left to the compiler,
// every accumulator value that is used by vector instructions after an MFMA loop is copied to a VGPR at the loop exit, all at
// once: with 256 accumulator registers per wave that runs the whole kernel -- the MFMA loop's operand rings included -- out of
// registers (post_frag2_kernel: 1 484 bytes of scratch per lane -> 0, 256 -> 114 VGPRs).
// Hand-placed v_accvgpr_read is outside the compiler's hazard bookkeeping (an MFMA result may be read by a vector instruction
// only >= 18 wait states after the MFMA issued), and the scheduler is free to move an MFMA -- a pure register operation -- past a
// separate "wait" statement.  So the wait states live INSIDE the statement that reads: its inputs are registers of the tile,
// which every MFMA of that tile writes, so all of them issue before it, and it begins with 32 wait states.  (First version: a
// separate s_nop statement before the reads; with the MFMA loop fully unrolled the last, smallest product terms were then
// sometimes missing from the value read -- seen as 1e-6 differences of a replicated factor between ranks.)
__device__ __forceinline__ void acc_read_tile(const f32x16& src, f32x16& dst) {
    float o[16];
    const float i0 = src[0], i1 = src[1], i2 = src[2], i3 = src[3], i4 = src[4], i5 = src[5], i6 = src[6], i7 = src[7];
    const float i8 = src[8], i9 = src[9], i10 = src[10], i11 = src[11], i12 = src[12], i13 = src[13], i14 = src[14], i15 = src[15];
    asm volatile("s_nop 15\n\ts_nop 15\n\t"
                 "v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %9\n\tv_accvgpr_read_b32 %2, %10\n\tv_accvgpr_read_b32 %3, %11\n\t"
                 "v_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\tv_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "a"(i0), "a"(i1), "a"(i2), "a"(i3), "a"(i4), "a"(i5), "a"(i6), "a"(i7));
    asm volatile("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %9\n\tv_accvgpr_read_b32 %2, %10\n\tv_accvgpr_read_b32 %3, %11\n\t"
                 "v_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\tv_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15"
                 : "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
                 : "a"(i8), "a"(i9), "a"(i10), "a"(i11), "a"(i12), "a"(i13), "a"(i14), "a"(i15));
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r] = o[r];
}

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

// one 1 KiB LDS-DMA piece: 64 lanes x 16 bytes from (descriptor, lane * 16 + soff) to lds + lane * 16 (lds wave-uniform).
// (Plain functions, not inside the kernel template: with the address-space cast in the template's lambda hipcc (ROCm 7.2)
// silently drops the HOST-side instantiation of the kernel -- no diagnostic, the launch stub is simply missing at link time.)
__device__ __forceinline__ void lds_dma_piece(__amdgpu_buffer_rsrc_t r, unsigned char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void lds_dma_piece_nt(__amdgpu_buffer_rsrc_t r, unsigned char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 2);
}

}  // namespace vbmf
