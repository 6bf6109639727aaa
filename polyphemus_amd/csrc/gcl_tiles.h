// gcl_tiles.h — tile geometry and small types shared by the MFMA pipelines of gcl.hip (the three products of a GCL
// layer) and linear.hip (plain linear layers on the same pipelines).
#pragma once
#include "common.h"
#include "tile_order.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
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

}  // namespace
