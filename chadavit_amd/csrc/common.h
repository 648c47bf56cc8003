// chadavit_amd HIP kernels for gfx950 (MI355X / CDNA4) -- shared device helpers.
// Wavefront = 64 lanes everywhere; MFMA tiles are v_mfma_f32_16x16x32_bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chadavit_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define CHADA_LDS __attribute__((address_space(3)))

// Entry check: a HIP error that is already pending when an entry point is called (an asynchronous fault of an earlier
// launch, a failed launch of another library) is REPORTED as 1000 + hipError_t, not swallowed -- the caller would otherwise
// see it attributed to a later, innocent call or never.  hipErrorNotReady is not a fault: it is what hipEventQuery /
// hipStreamQuery leave behind when polled (PyTorch's caching allocator polls events all the time), so it alone is dropped.
#define CHADA_ENTRY()                                                       \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess && e__ != hipErrorNotReady) return 1000 + (int)e__; \
  } while (0)

#define CHADA_CHECK_LAUNCH()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return 1000 + (int)e__;   \
  } while (0)

namespace chada {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- MFMA: D(16x16) += A(16x32) * B(32x16).
// A frag: lane l holds A[row = l&15][k = (l>>4)*8 + j], j<8
// B frag: lane l holds B[k = (l>>4)*8 + j][col = l&15]
// C frag: lane l holds C[row = (l>>4)*4 + r][col = l&15], r<4
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- LDS reads
__device__ __forceinline__ bf16x8 lds_read8(const bf16_t* p) {  // 16-byte aligned
  return *reinterpret_cast<const bf16x8*>(p);
}
// Hardware transpose read (ds_read_b64_tr_b16).  Within each 16-lane group the 16 lanes address a
// [4 rows][16 cols] bf16 block: lane ii supplies &blk[ii>>2][(ii&3)*4] (8 bytes); it receives
// column ii of the block: {blk[0][ii], blk[1][ii], blk[2][ii], blk[3][ii]}.
__device__ __forceinline__ bf16x4 lds_read_tr4(const bf16_t* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((CHADA_LDS s16x4*)(p));
  return __builtin_bit_cast(bf16x4, v);
}
// 8 k-slots for an MFMA operand whose k axis is the ROW axis of a row-major LDS tile.
// k-slot order of lane group g = l>>4:  rows {4g..4g+3} then {16+4g..16+4g+3} of the 32-row k-step.
// `tile` points at (row0 of the k-step, col0 of the 16-wide block); ld = row stride in elements.
__device__ __forceinline__ bf16x8 lds_read_tr8(const bf16_t* tile, int ld) {
  const int l = lane_id();
  const int ii = l & 15, g = l >> 4;
  const bf16_t* p = tile + (4 * g + (ii >> 2)) * ld + (ii & 3) * 4;
  bf16x4 lo = lds_read_tr4(p);
  bf16x4 hi = lds_read_tr4(p + 16 * ld);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- LDS-DMA through a buffer resource (buffer_load_dwordx4 ... lds) rather than global_load_lds_dwordx4.  Same data
// path; the difference is in hipcc's wait insertion: global_load_lds is a FLAT-class instruction, and while one is
// pending the compiler degrades EVERY lgkmcnt wait to lgkmcnt(0) -- no LDS read of the consuming stage can then overlap the
// MFMAs that use the previous one.  The MUBUF form leaves the counted waits alone.  Address = base (wave-uniform, in the
// resource) + per-lane byte offset (VGPR) + wave-uniform byte offset (SGPR): the per-stage part usually needs no VALU.
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc make_rsrc(const void* base) {
  // raw buffer (stride 0), no bounds (callers clamp rows themselves), gfx9 dword 3 = DATA_FORMAT 32
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ void lds_dma16(BufRsrc rs, bf16_t* lds_dst, unsigned lane_bytes, unsigned uniform_bytes) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (CHADA_LDS void*)lds_dst, 16, lane_bytes, uniform_bytes, 0, 0);
}
__device__ __forceinline__ void lds_dma4(BufRsrc rs, void* lds_dst, unsigned lane_bytes, unsigned uniform_bytes) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (CHADA_LDS void*)lds_dst, 4, lane_bytes, uniform_bytes, 0, 0);
}

// ---- cross-lane exchange without the LDS crossbar (ds_bpermute): DPP inside a 16-lane row, the gfx950 permlane swaps
// across rows.  Every helper returns, in every participating lane, the reduction over the lanes it names.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// {v[l], v[l^16]} as (a, b) for the combine: v_permlane16_swap exchanges the odd rows of one register with the even rows
// of the other (v_permlane32_swap: upper half <-> lower half).  Written as inline asm on two read-write operands: the
// builtin's second result is mis-tracked by this compiler when both inputs carry the same value.  The s_nops cover the
// VALU-write -> permlane-swap and permlane-swap -> DPP-read wait states the hazard recogniser cannot see through asm.
__device__ __forceinline__ void swap16(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// The same exchanges on two DIFFERENT registers (epilogue store widening): afterwards a = (a.row0, b.row0, a.row2, b.row2) and
// b = (a.row1, b.row1, a.row3, b.row3) for the 16-lane rows (swap16x), a = (a.lo, b.lo), b = (a.hi, b.hi) for the halves (swap32x).
__device__ __forceinline__ void swap16x(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32x(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// reductions over the 4 lanes {l, l^16, l^32, l^48} (same l & 15)
__device__ __forceinline__ float rows_sum(float v) {
  float a, b;
  swap16(v, a, b); v = a + b;
  swap32(v, a, b); return a + b;
}
__device__ __forceinline__ float rows_max(float v) {
  float a, b;
  swap16(v, a, b); v = fmaxf(a, b);
  swap32(v, a, b); return fmaxf(a, b);
}
// reductions over the 16 lanes of a row: xor 1, xor 2 (quad_perm), then mirror pairings (quads / halves already uniform)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
// ---- wave reductions (all 64 lanes)
__device__ __forceinline__ float wave_sum(float v) { return rows_sum(row16_sum(v)); }
__device__ __forceinline__ float wave_max(float v) { return rows_max(row16_max(v)); }

// ReLU as ONE instruction on an MFMA result: fmaxf(x, 0.f) compiles to two v_max_f32 -- hipcc first canonicalises the accumulator register
// (v_max_f32 x, x, x: it cannot know an MFMA output is not a signalling NaN), then takes the maximum; in the block kernels that was 16 of the 83 vector
// instructions of an FFN chunk (round 6).  The integer maximum of the bit pattern against 0 is the same function for every non-NaN input (positive floats
// are positive integers, negative floats and -0 negative ones) and returns +0 for them; a NaN with the sign bit clear passes through (torch.relu's
// behaviour -- fmaxf dropped it), one with the sign bit set becomes 0.
__device__ __forceinline__ float relu_f(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  bf16x4 r;
  r[0] = (bf16_t)a; r[1] = (bf16_t)b; r[2] = (bf16_t)c; r[3] = (bf16_t)d;
  return r;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

}  // namespace chada
