// gcl_tiles.h — tile geometry and small types shared by the MFMA pipelines of gcl.hip (the three products of a GCL
// layer) and linear.hip (plain linear layers on the same pipelines).
#pragma once
#include "common.h"
#include "tile_order.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// One product of the split chain.  H2 = false: operands are planes of the exact three-term bf16 split (six products per
// fp32 product); H2 = true: the two planes of the fp16 pair format of common.h (three products).  The fragments travel as
// 16-byte registers either way.
template <bool H2>
__device__ inline f32x16 gcl_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#define GCL_OOB ((int)0x80000000)     // byte offset >= num_records: the buffer load returns 0 / the store is dropped
#ifndef GCL_BDEPTH
#define GCL_BDEPTH 2          // k-steps of weight fragments in flight per MFMA wave
#endif

namespace {
constexpr int BM = PM_TILE_ROWS;   // rows per workgroup
constexpr int CH = 128;       // features per chunk of the chunked pipelines
constexpr int ROWB = CH * 2;  // bytes of one row of a chunk image (one plane)
constexpr int PLANE = BM * ROWB;
constexpr int IMG = 3 * PLANE;

// ---- weight-gradient tiles (k_gcl_dw of gcl.hip, k_rows_tn of linear.hip): C += A^T B over the node rows, 128 x 128
// output tiles, both operands staged as [node][column] bf16 planes and read TRANSPOSED (ds_read_b64_tr_b16)
constexpr int DW_T = 128;                 // output tile edge
constexpr int DW_KT = 32;                 // node rows per LDS tile
#ifndef GCL_DW_PAD
#define GCL_DW_PAD 64
#endif
constexpr int DW_PITCH = DW_T * 2 + GCL_DW_PAD;   // bytes per image row (+64: the 4 rows x 2 column halves a 32-lane transposing read touches fall into 64 different banks)
constexpr int DW_PLANE = DW_KT * DW_PITCH;
constexpr int DW_STAGE = 2 * 3 * DW_PLANE;          // A' image + dh image, three planes each
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
// operand fragment of the 32-wide block starting at image column `c0` for k-step ks (16 nodes): 8 consecutive nodes per lane
__device__ inline bf16x8 dw_frag(const char* S, int c0, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2;
  const int k = ks * 16 + 8 * (g >> 1) + q;
  const int colb = (c0 + 16 * (g & 1) + 4 * (i & 3)) * 2;
  typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4;
  const char* p0 = S + k * DW_PITCH + colb;
  const s16x4_t t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
  const s16x4_t t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * DW_PITCH));
  const s16x8_t t = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, t);
}

}  // namespace
