// gemm.hip — fp32-accurate GEMM on the CDNA4 matrix cores.
//
// Replaces every `addmm` / `mm` on the path: nn.Linear layers, and the 6 relation GEMMs + root GEMM of GCL.forward
// (model.py:112,116), which become ONE grouped launch on the compact aggregate [track block | onset | next | x]
// (K = 4d, row lists per track relation, stacked weights) — or one plain call on [h_0|...|h_5|x] (K = 7d).
// Plain bf16 MFMA cannot hold the 1e-4 parity bar through 16 BatchNorm'd layers; three arithmetic modes keep fp32
// accuracy (one kernel template, 4 waves (2x2) of 32x32 MFMA tiles per workgroup unless noted):
//   MODE 0  fp32 MFMA   v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak)      C0 64x64x16, C1 128x128x16, C2 64x64x32, C3 128x128x32
//   MODE 1  split "x6"  v_mfma_f32_32x32x16_bf16, fp32 operands split in the kernel by 4 extra producer waves
//                                                                      C4 128x128x16, C5 128x64x16, C6 64x64x32, C7 128x128x32
//   MODE 2  planes      v_mfma_f32_32x32x16_bf16, operands PRE-SPLIT by the kernels that produced them
//                                                                      C8 64x64x32 (C9 128x64x32, C10 128x128x32 measured slower)
// Split arithmetic: every fp32 value is EXACTLY the sum of three bf16 terms (x = x1 + x2 + x3, 8 significand bits each,
// same exponent range) and the six partial products of weight >= 2^-16
//   x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1)
// are accumulated in the fp32 MFMA accumulators; the three dropped products are <= 2^-24 relative, i.e. below the
// rounding of an fp32 FMA chain (tests: the same 5e-6 bound against float64 as the fp32 mode, measured ~1e-7).
// fp32 mode: k-major LDS tiles (an MFMA operand read is 32 consecutive floats per half-wave, conflict-free
// ds_read_b32), global loads staged through registers one k-tile ahead, double-buffered LDS, one barrier per k-tile.
// Split modes: three bf16 plane images per operand, fragments by ds_read_b128 (k-contiguous operands) or
// ds_read_b64_tr_b16 (row-contiguous operands, hardware transpose).  Workgroup ids are remapped so that tiles sharing
// an operand panel run on the same XCD (shared L2).  The gathered dimension may be indirect (row map + device-side
// count): drum / non-drum routing of the content decoder (model.py:552-576), per-relation row lists of the GCL.
#include "common.h"
#include "prof.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef PM_BD_DUAL
#define PM_BD_DUAL 0
#endif
#ifndef PM_TN_DUAL
#define PM_TN_DUAL 1
#endif
#ifndef PM_PL_WAVES
#define PM_PL_WAVES 5          // waves per SIMD the 64x64 planes kernel is register-budgeted for
#endif
#ifndef PM_BD_WAVES
#define PM_BD_WAVES 5
#endif
struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias;
  const int32_t* rowmap; const int32_t* dyn_entries;
  int M, N, K, lda, ldb, ldc, rpe, flags, kper, ntm, ntn;
  // grouped launch: blockIdx.y = group; every group has its own operand bases, row-map slice and live count
  int64_t a_boff, b_boff, c_boff, bias_boff; int map_boff, dyn_boff;
  // stacked operands: stored rows >= split are shared by all groups and live at base + hi (elements)
  int b_split, c_split; int64_t b_hi, c_hi;
  // ngroups > 1 with packed != 0: the groups partition the gathered rows (device-side counts), and the live row
  // panels of ALL groups are enumerated along blockIdx.x (gridDim.y == 1) so that no dead workgroup is launched
  int ngroups, packed;
  int64_t a_ps, b_ps;                        // planes mode: distance between the bf16 planes of A / B (elements)
  // row classes of the compact GCL (see PmGemmDesc.class_ptr): boundaries [ngroups][5], block size, masked dimension
  const int32_t* cls_ptr; int cls_blk, cls_dim;
  double* colstats;                          // optional [2][N]: += column sums of the stored values and of their squares
  // MODE 3: B (a weight matrix) as fragment-major planes (pm_split_planes_frag); bf_n = K/16 (transB) or N/32 tiles
  const char* bfrag; int bf_n;
  float* colsum_a;                           // optional (transA, fp32 tiles): += sum over K of A[k, m]  (bias gradient)
  unsigned* gate;                            // deterministic mode (common.h): the MFMA waves take turns in the epilogue
};

__device__ static inline int64_t map_row(const int32_t* map, int rpe, int r) {
  return map ? (rpe == 1 ? (int64_t)map[r] : (int64_t)map[r / rpe] * rpe + (r % rpe)) : (int64_t)r;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define PM_OOB ((int)0x80000000)     // byte offset >= num_records: the buffer load returns 0, no branch, no fault

// Stage one operand tile (R rows x BK k) through registers into its k-major LDS image S[k][r].
//  KC = true : stored k-contiguous, element (r,k) at P[row(r)*ld + k]   (A when !transA, B when transB)
//  KC = false: stored r-contiguous, element (r,k) at P[row(k)*ld + r]   (A when transA, B when !transB)
// Loads are raw buffer loads (32-bit byte offsets, hardware range check): out-of-tile elements get the
// offset PM_OOB and read as 0, so the k-loop has no predicated branches and no early s_waitcnt, and the
// prefetch of tile t+1 stays in flight behind the MFMAs of tile t.
template <int R, int BK, int THREADS, bool KC, bool VEC>
struct TileStage {
  static constexpr int NV = VEC ? (R * BK / 4) / THREADS : 0;
  static constexpr int NS = VEC ? 0 : (R * BK) / THREADS;
  static constexpr int NE = VEC ? NV : NS;
  static constexpr int LD = KC ? R + 1 : R + 4;      // KC: transposing scalar stores spread over banks; else 16-B rows
  static_assert(!VEC || (R * BK / 4) % THREADS == 0, "tile does not divide over the workgroup");
  u32x4 v[NV > 0 ? NV : 1];
  unsigned s[NS > 0 ? NS : 1];
  int base[NE];                                       // tile-invariant byte offset of each element, or PM_OOB
  int krow[NE];                                       // !KC + row map: physical row of the next tile's k (prefetched)
  int tail[KC && VEC ? NE : 1];                        // KC + VEC: k values left from this float4 on (< 4: the K tail, zeroed in store)
  __amdgpu_buffer_rsrc_t rsrc;
  const int32_t* map;
  int rpe, ld, tid;                                   // tid: index of this thread among the THREADS staging threads
  int split, lo_b, hi_b;                              // stacked operand: byte offset of stored rows < split / >= split

  __device__ inline int stack_off(int row) const { return row < split ? lo_b : hi_b; }
  __device__ inline void init(const float* P, int ld_, int r0, int rmax, const int32_t* map_, int rpe_, int split_ = 0,
                              int lo_b_ = 0, int hi_b_ = 0) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, PM_OOB, 0x00020000);
    map = map_; rpe = rpe_; ld = ld_; tid = threadIdx.x % THREADS;
    split = split_; lo_b = lo_b_; hi_b = hi_b_;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int f = tid + j * THREADS;
      if (KC) {                                        // row fixed per thread: resolve the row map once
        const int r = r0 + (VEC ? f / (BK / 4) : f / BK);
        const int kb = VEC ? (f % (BK / 4)) * 16 : (f % BK) * 4;
        base[j] = r < rmax ? (int)(map_row(map, rpe, r) * ld * 4) + kb + stack_off(r) : PM_OOB;
      } else {
        const int r = r0 + (VEC ? (f % (R / 4)) * 4 : f % R);
        base[j] = r < rmax ? r * 4 : PM_OOB;
      }
    }
  }
  __device__ inline void load(int k0, int kmax) { load_to(v, k0, kmax); }
  template <int NW>
  __device__ inline void load_to(u32x4 (&w)[NW], int k0, int kmax) {
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int f = tid + j * THREADS;
      int off;
      if (KC) {
        const int k = k0 + (VEC ? (f % (BK / 4)) * 4 : f % BK);
        off = (k < kmax && base[j] >= 0) ? base[j] + k0 * 4 : PM_OOB;
        if constexpr (VEC) tail[j] = kmax - k;
      } else {
        const int k = k0 + (VEC ? f / (R / 4) : f / R);
        // gathered K rows: the row-map entry of THIS tile was fetched while the previous tile was being
        // multiplied (krow), so no index load sits between the MFMAs and the data load
        const int prow = map ? krow[j] : k;
        off = (k < kmax && base[j] >= 0) ? base[j] + prow * (ld * 4) + stack_off(k) : PM_OOB;
        if (map) {
          const int kn = k + BK;
          krow[j] = kn < kmax ? (rpe == 1 ? map[kn] : map[kn / rpe] * rpe + kn % rpe) : 0;
        }
      }
      if (VEC) w[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
      else s[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
    }
  }
  // first tile of a gathered-K operand: resolve its row-map entries up front
  __device__ inline void prime(int k0, int kmax) {
    if (KC || !map) return;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int f = tid + j * THREADS;
      const int k = k0 + (VEC ? f / (R / 4) : f / R);
      krow[j] = k < kmax ? (rpe == 1 ? map[k] : map[k / rpe] * rpe + k % rpe) : 0;
    }
  }
  // ---- split mode: three bf16 planes of the tile.
  //  KC : image [r][BK], fragments read with ds_read_b128 (8 consecutive k of the lane's row);
  //  !KC: image [k][R] (as loaded), fragments read with ds_read_b64_tr_b16 (hardware 4x16 transpose).
  // Row strides are padded so that both the 8-byte stores and the fragment reads are bank-conflict free.
  static constexpr int XROWB = KC ? 2 * BK + 16 : 2 * R + 32;
  static constexpr int XPLANE = (KC ? R : BK) * XROWB;
  static constexpr int XBYTES = 3 * XPLANE;
  __device__ inline void store_x6(char* __restrict__ S) const { store_x6_from(v, S); }
  template <int NW>
  __device__ inline void store_x6_from(const u32x4 (&v)[NW], char* __restrict__ S) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = tid + j * THREADS;
      const int off = KC ? (f / (BK / 4)) * XROWB + (f % (BK / 4)) * 8 : (f / (R / 4)) * XROWB + (f % (R / 4)) * 8;
      unsigned l1, l2, l3, u1, u2, u3;
      pm_split3_pair(__uint_as_float(v[j].x), __uint_as_float(v[j].y), l1, l2, l3);
      pm_split3_pair(__uint_as_float(v[j].z), __uint_as_float(v[j].w), u1, u2, u3);
      const u32x2 p1 = {l1, u1}, p2 = {l2, u2}, p3 = {l3, u3};
      *reinterpret_cast<u32x2*>(S + off) = p1;
      *reinterpret_cast<u32x2*>(S + XPLANE + off) = p2;
      *reinterpret_cast<u32x2*>(S + 2 * XPLANE + off) = p3;
    }
  }
  // byte offset (inside a plane) of this lane's fragment for 32-row block 0 / k-step 0
  __device__ static inline int x6_lane_off(int row0, int lane) {
    if (KC) return (row0 + (lane & 31)) * XROWB + 16 * (lane >> 5);
    const int g = lane >> 4, i = lane & 15;
    return (8 * (g >> 1) + (i >> 2)) * XROWB + (row0 + 16 * (g & 1) + 4 * (i & 3)) * 2;
  }
  static constexpr int XSTEP_I = KC ? 32 * XROWB : 64;          // next 32-row block
  static constexpr int XSTEP_K = KC ? 32 : 16 * XROWB;          // next 16-wide k-step
  __device__ static inline bf16x8 x6_frag(const char* p) {
    if (KC) return *reinterpret_cast<const bf16x8*>(p);
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * XROWB));
    const s16x8 t = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, t);
  }

  __device__ inline void store(float* __restrict__ S) const {
    if (VEC) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int f = tid + j * THREADS;
        if (KC) {
          // (K not a multiple of 4: the last float4 of a row reaches past K into the row's next columns — zeroed here)
          const int r = f / (BK / 4), k = (f % (BK / 4)) * 4, tl = tail[j];
          S[(k + 0) * LD + r] = __uint_as_float(v[j].x); S[(k + 1) * LD + r] = tl > 1 ? __uint_as_float(v[j].y) : 0.f;
          S[(k + 2) * LD + r] = tl > 2 ? __uint_as_float(v[j].z) : 0.f; S[(k + 3) * LD + r] = tl > 3 ? __uint_as_float(v[j].w) : 0.f;
        } else {
          const int k = f / (R / 4), r = (f % (R / 4)) * 4;
          *reinterpret_cast<u32x4*>(S + k * LD + r) = v[j];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int e = tid + j * THREADS;
        if (KC) S[(e % BK) * LD + e / BK] = __uint_as_float(s[j]);
        else S[(e / R) * LD + e % R] = __uint_as_float(s[j]);
      }
    }
  }
};

// Operand given as three bf16 planes (the exact split x = x1 + x2 + x3 done once by the kernel that PRODUCED the
// tensor): the tile is staged with plain 16-byte loads / ds_write_b128 into the same LDS images as split mode, no
// conversion arithmetic in the GEMM.  Element = bf16, `ld` in elements, rows addressed like TileStage.
template <int R, int BK, int THREADS, bool KC>
struct PlaneStage {
  using Img = TileStage<R, BK, THREADS, KC, true>;      // padded image geometry (shared with the in-kernel split mode)
  // XOR-swizzled images where the row is exactly 64 bytes (KC, BK = 32) or 128 bytes (!KC, R = 64): no pad, and
  // both the 16-byte stores and the fragment reads are bank-conflict free —
  //   KC : 16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3): the 16 rows a ds_read_b128 group reads hit 16
  //        different 16-byte slots of the 256-byte bank window, a row stays one contiguous 64-byte line for the stores;
  //   !KC: the two 64-byte halves of k-row k are swapped when (k >> 1) & 1: the four k-rows of a ds_read_b64_tr_b16
  //        block land in four different 64-byte slots.
  static constexpr bool SWZ = KC ? (BK == 32) : (R == 64);
  static constexpr int PROWB = SWZ ? (KC ? 64 : 128) : Img::XROWB;
  static constexpr int PPLANE = (KC ? R : BK) * PROWB;
  static constexpr int PBYTES = 3 * PPLANE;
  // fragment of the 32-row block starting at tile row `row0` for k-step ks (16 wide), plane base S
  __device__ static inline bf16x8 frag(const char* S, int row0, int ks, int lane) {
    if (KC) {
      const int rr = row0 + (lane & 31), c = ks * 2 + (lane >> 5);
      const int cc = SWZ ? (c ^ ((rr >> 2) & 3)) : c;
      return *reinterpret_cast<const bf16x8*>(S + rr * PROWB + cc * 16);
    }
    const int g = lane >> 4, i = lane & 15, q = i >> 2;
    const int k = ks * 16 + 8 * (g >> 1) + q;
    int colb = (row0 + 16 * (g & 1) + 4 * (i & 3)) * 2;
    if (SWZ) colb ^= ((k >> 1) & 1) << 6;                 // (same for k + 4)
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const char* p0 = S + k * PROWB + colb;
    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * PROWB));
    const s16x8 t = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, t);
  }
  static constexpr int NV = (R * BK / 8) / THREADS;     // 16-byte chunks per plane and thread
  static_assert((R * BK / 8) % THREADS == 0 && NV >= 1, "tile does not divide over the workgroup");
  u32x4 v[3][NV];
  int base[NV], krow[NV];
  __amdgpu_buffer_rsrc_t rsrc[3];
  const int32_t* map;
  int rpe, ld, tid, split, lo_b, hi_b;

  __device__ inline int stack_off(int row) const { return row < split ? lo_b : hi_b; }
  __device__ inline void init(const char* P, int64_t plane_bytes, int ld_, int r0, int rmax, const int32_t* map_,
                              int rpe_, int split_ = 0, int lo_b_ = 0, int hi_b_ = 0) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
      rsrc[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P + p * plane_bytes), 0, PM_OOB, 0x00020000);
    map = map_; rpe = rpe_; ld = ld_; tid = threadIdx.x % THREADS;
    split = split_; lo_b = lo_b_; hi_b = hi_b_;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = tid + j * THREADS;
      if (KC) {
        const int r = r0 + f / (BK / 8);
        base[j] = r < rmax ? (int)(map_row(map, rpe, r) * ld * 2) + (f % (BK / 8)) * 16 + stack_off(r) : PM_OOB;
      } else {
        const int r = r0 + (f % (R / 8)) * 8;
        base[j] = r < rmax ? r * 2 : PM_OOB;
      }
    }
  }
  __device__ inline void prime(int k0, int kmax) {
    if (KC || !map) return;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int k = k0 + (tid + j * THREADS) / (R / 8);
      krow[j] = k < kmax ? (rpe == 1 ? map[k] : map[k / rpe] * rpe + k % rpe) : 0;
    }
  }
  __device__ inline void load(int k0, int kmax) { load_to(v, k0, kmax); }
  __device__ inline void load_to(u32x4 (&v)[3][NV], int k0, int kmax) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = tid + j * THREADS;
      int off;
      if (KC) {
        const int k = k0 + (f % (BK / 8)) * 8;
        off = (k < kmax && base[j] >= 0) ? base[j] + k0 * 2 : PM_OOB;
      } else {
        const int k = k0 + f / (R / 8);
        const int prow = map ? krow[j] : k;
        off = (k < kmax && base[j] >= 0) ? base[j] + prow * (ld * 2) + stack_off(k) : PM_OOB;
        if (map) {
          const int kn = k + BK;
          krow[j] = kn < kmax ? (rpe == 1 ? map[kn] : map[kn / rpe] * rpe + kn % rpe) : 0;
        }
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) v[p][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc[p], off, 0, 0);
    }
  }
  __device__ inline void store(char* __restrict__ S) const { store_from(v, S); }
  __device__ inline void store_from(const u32x4 (&v)[3][NV], char* __restrict__ S) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = tid + j * THREADS;
      int off;
      if (KC) {
        const int r = f / (BK / 8), kc = f % (BK / 8);
        off = r * PROWB + (SWZ ? (kc ^ ((r >> 2) & 3)) : kc) * 16;
      } else {
        const int k = f / (R / 8), rc = f % (R / 8);
        off = k * PROWB + (SWZ ? ((rc * 16) ^ (((k >> 1) & 1) << 6)) : rc * 16);
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(S + p * PPLANE + off) = v[p][j];
    }
  }
};

// MODE 0: fp32 MFMA.  MODE 1 ("x6"): fp32 operands split in the kernel.  MODE 2 ("planes"): operands pre-split.
// (planes mode, 64x64 tile: capped at 100 VGPRs so that FIVE workgroups fit a CU — the 1036 tiles of a GCL
//  contraction then run as one resident wave of workgroups instead of 1024 + a 12-tile tail)
template <int BM, int BN, int BK, int WVM, int WVN, bool TA, bool TB, bool VA, bool VB, int MODE>
__global__ void __launch_bounds__(64 * WVM * WVN * (MODE == 1 ? 2 : 1))
    __attribute__((amdgpu_waves_per_eu((MODE == 2 && BM * BN <= 64 * 64) ? PM_PL_WAVES : (MODE == 3 ? (BK > 32 ? 4 : PM_BD_WAVES) : 1),
                                       (MODE == 2 && BM * BN <= 64 * 64) ? PM_PL_WAVES : (MODE == 3 ? (BK > 32 ? 4 : PM_BD_WAVES) : 8))))
    k_gemm(GemmArgs g) {
  constexpr bool X6 = MODE == 1, PL = MODE >= 2, BD = MODE == 3;
  static_assert(MODE == 0 || (VA && VB && BK % 16 == 0), "split modes stage with 16-byte loads");
  // THREADS = MFMA threads = staging threads.  fp32 mode: the same waves do both.  Split mode: the block has
  // 2*THREADS threads, waves [0, WVM*WVN) multiply and waves [WVM*WVN, 2*WVM*WVN) load + split + store, so the
  // conversion arithmetic of tile t+1 runs on the SIMD's vector ALU while its matrix core works on tile t.
  constexpr int THREADS = 64 * WVM * WVN;
  constexpr int WM = BM / WVM, WN = BN / WVN, TM = WM / 32, TN = WN / 32;
  using StA = TileStage<BM, BK, THREADS, !TA, VA>;
  using StB = TileStage<BN, BK, THREADS, TB, VB>;
  constexpr int LDA_S = StA::LD, LDB_S = StB::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As0 = smem;                              // two A buffers, then two B buffers
  float* const Bs0 = smem + 2 * BK * LDA_S;
  char* const Ax0 = reinterpret_cast<char*>(smem);      // split mode: two A images (3 planes each), then two B images
  char* const Bx0 = Ax0 + 2 * StA::XBYTES;

  // deterministic mode: a workgroup that leaves early hands the turns of its MFMA waves on
#define GEMM_SKIP do { pm_turn_skip_block(g.gate, WVM * WVN); return; } while (0)
  int grp = blockIdx.y, t = blockIdx.x, zs = blockIdx.z, nwg;
  if (TA) {
    // weight gradients: XCD-aware order over the WHOLE grid (tiles x groups x K-slices).  Workgroup L runs on XCD
    // L % 8; each XCD gets a contiguous run of (group, K-slice) slabs, so a K range of the two operands is
    // fetched into ONE L2 instead of all eight (measured: 312 MB -> memory-side reads per launch before).
    const int nx = g.ntm * g.ntn, ny = gridDim.y, total = nx * ny * (int)gridDim.z;
    const int L = t + nx * (grp + ny * zs);
    const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    const int V = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    t = V % nx;
    grp = (V / nx) % ny;
    zs = (V / nx) / ny;
  }
  if (!TA && g.packed) {                     // packed groups: find the group of this tile from the device-side counts
    int tot = 0;
    for (int q = 0; q < g.ngroups; ++q) tot += (g.dyn_entries[q * g.dyn_boff] * g.rpe + BM - 1) / BM;
    nwg = tot * g.ntn;
    if (t >= nwg) GEMM_SKIP;
    {                                        // XCD-aware order over the live tiles of all groups (see below)
      const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    for (grp = 0; grp < g.ngroups - 1; ++grp) {
      const int nt = ((g.dyn_entries[grp * g.dyn_boff] * g.rpe + BM - 1) / BM) * g.ntn;
      if (t < nt) break;
      t -= nt;
    }
  }
  if (g.ngroups > 1) {                       // grouped launch (uniform branch): shift everything to this group
    const int64_t bi = grp;
    if (PL) {                                  // bf16 planes: element = 2 bytes
      g.A = reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.A) + bi * g.a_boff * 2);
      if (g.b_split == 0) g.B = reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.B) + bi * g.b_boff * 2);
    } else {
      g.A += bi * g.a_boff;
      if (g.b_split == 0) g.B += bi * g.b_boff;
    }
    if (g.c_split == 0) g.C += bi * g.c_boff;
    if (g.bias) g.bias += bi * g.bias_boff;
    if (g.rowmap) g.rowmap += bi * g.map_boff;
    if (g.dyn_entries) g.dyn_entries += bi * g.dyn_boff;
  }
  int M = g.M, K = g.K, kper = g.kper;
  if (g.dyn_entries) {                       // data-dependent size of the gathered dimension, read on device
    const int n = *g.dyn_entries * g.rpe;
    if (TA) {                                // split-K ranges follow the LIVE length, so every z-slice has work
      K = n < K ? n : K;
      kper = (((K + (int)gridDim.z - 1) / (int)gridDim.z + BK - 1) / BK) * BK;
    } else M = n < M ? n : M;
  }
  // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch), so hand each XCD a
  // contiguous run of tiles; tiles of one row panel (n fastest) then share the panel through one L2.
  // (with a device-side row count only the live row panels take part, so they still spread over all 8 XCDs)
  if (!TA && !g.packed) {
    nwg = ((M + BM - 1) / BM) * g.ntn;
    if (t >= nwg) GEMM_SKIP;
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (t / g.ntn) * BM, n0 = (t % g.ntn) * BN;
  if (m0 >= M) GEMM_SKIP;
  // Row classes (compact GCL): rows [b1,b3) of the group's list receive onset edges, rows [b2,b4) next edges; for all
  // other rows that block of the aggregate is identically zero.  Forward (cls_dim 1): skip those K blocks for a row
  // tile without such rows; input gradient (2): skip the output tiles nobody reads; weight gradient (3): contract
  // only over the rows whose block is non-zero.
  bool use_on = true, use_nx = true;
  if (g.cls_dim) {
    const int* cb = g.cls_ptr + grp * 5;
    const int on_lo = cb[1], on_hi = cb[3], nx_lo = cb[2], nx_hi = cb[4];
    if (TA) {
      const int mb = m0 / g.cls_blk;
      const int lo = mb == 1 ? on_lo : (mb == 2 ? nx_lo : 0), hi = mb == 1 ? on_hi : (mb == 2 ? nx_hi : K);
      g.rowmap += lo;
      K = hi - lo;
      if (K <= 0) GEMM_SKIP;
      kper = (((K + (int)gridDim.z - 1) / (int)gridDim.z + BK - 1) / BK) * BK;
    } else {
      use_on = m0 < on_hi && m0 + BM > on_lo;
      use_nx = m0 < nx_hi && m0 + BM > nx_lo;
      if (g.cls_dim == 2) {
        const int nb_ = n0 / g.cls_blk;
        if ((nb_ == 1 && !use_on) || (nb_ == 2 && !use_nx)) GEMM_SKIP;
      }
    }
  }
  const int kbeg = zs * kper;
  int kend = kbeg + kper;
  if (kend > K) kend = K;
  if (kbeg >= kend) GEMM_SKIP;

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) % (WVM * WVN);
  const bool producer = X6 && threadIdx.x >= THREADS;
  const unsigned my_turn = pm_linear_block() * (WVM * WVN) + wave;      // deterministic mode (MFMA waves only)
  bool in_turn = false;
  const int wr = wave / WVN, wc = wave % WVN;
  const int li = lane & 31, lh = lane >> 5;

  // !TA: A k-contiguous, rows = M (gathered).  TA: A r-contiguous, stored rows = K (gathered).
  // TB : B k-contiguous (stored [N,K]), never gathered.  !TB: B stored [K,N]; its K rows are gathered iff TA.
  const int32_t* mapA = g.rowmap;
  const int32_t* mapB = (TA && !TB) ? g.rowmap : nullptr;
  StA sa;
  StB sb;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (PL) {
    // Pre-split operands: single LDS image per operand (two barriers per k-tile, 4 workgroups per CU cover them),
    // the loads of tile t+1 are in flight during the MFMAs of tile t.
    using PA_ = PlaneStage<BM, BK, THREADS, !TA>;
    using PB_ = PlaneStage<BN, BK, THREADS, TB>;
    PA_ pa;
    PB_ pb;
    pa.init(reinterpret_cast<const char*>(g.A), g.a_ps * 2, g.lda, m0, M, mapA, g.rpe);
    if constexpr (!BD)
      pb.init(reinterpret_cast<const char*>(g.B), g.b_ps * 2, g.ldb, n0, g.N, mapB, g.rpe, g.b_split,
              (int)(grp * g.b_boff * 2), (int)(g.b_hi * 2));
    // k sequence: all of [kbeg, kend), or (forward with row classes) only the K blocks this row tile needs
    int kblk = kend - kbeg, kv_end = kend - kbeg, s1 = 1, s2 = 2;          // (scalars: no indexed private array)
    if (!TA && g.cls_dim == 1) {
      kblk = g.cls_blk;
      s1 = use_on ? 1 : (use_nx ? 2 : 3);
      s2 = use_on ? (use_nx ? 2 : 3) : 3;
      kv_end = (2 + (use_on ? 1 : 0) + (use_nx ? 1 : 0)) * kblk;
    }
    auto kmap = [&](int v) {                     // virtual k (multiple of BK) -> k; past the end -> kend (loads return 0)
      if (v >= kv_end) return kend;
      if (kblk == kend - kbeg) return kbeg + v;
      const int q = v / kblk;
      return (q == 0 ? 0 : q == 1 ? s1 : q == 2 ? s2 : 3) * kblk + (v - q * kblk);
    };
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    f32x16 accb[PM_BD_DUAL ? TM : 1][PM_BD_DUAL ? TN : 1];
#pragma unroll
    for (int i = 0; i < (PM_BD_DUAL ? TM : 1); ++i)
#pragma unroll
      for (int j = 0; j < (PM_BD_DUAL ? TN : 1); ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][j][r] = 0.f;
    if constexpr (BD) {
      // B direct (MODE 3): B is a weight matrix kept as FRAGMENT-MAJOR planes (pm_split_planes_frag): the 1 KiB that a
      // wave needs for one (32-row tile, 16-wide k-step, plane) MFMA operand is contiguous, so every wave takes its B
      // fragments straight from L2 into registers with one coalesced 16-byte load per lane — B never touches LDS
      // (half the LDS bytes of the LDS-staged tile), and only the A image (12 KB) is staged.  The fragment registers
      // of k-step t are refilled with those of k-step t+1 as soon as their MFMAs have issued.
      static_assert(!TA, "B direct: the weight operand of the forward / input-gradient products");
      const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g.bfrag), 0, PM_OOB, 0x00020000);
      const int shift_lo = g.b_split > 0 ? (int)(grp * g.b_boff / g.ldb) : 0, shift_hi = g.b_split > 0 ? (int)(g.b_hi / g.ldb) : 0;
      int bbase[TN];                                    // TB: byte offset of the row tile; !TB: index of the column tile
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nrow = n0 + wc * WN + j * 32;
        if (nrow >= g.N) bbase[j] = -1;
        else if (TB) bbase[j] = ((nrow + (nrow < g.b_split ? shift_lo : shift_hi)) >> 5) * g.bf_n * 3072 + lane * 16;
        else bbase[j] = (nrow >> 5) * 3072 + lane * 16;
      }
      auto bload = [&](bf16x8 (&dst)[3][TN], int k) {   // fragments of the 16-wide k-step starting at real k
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          int off = PM_OOB;
          if (k < kend && bbase[j] >= 0) {
            if (TB) off = bbase[j] + (k >> 4) * 3072;
            else off = bbase[j] + ((k + (k < g.b_split ? shift_lo : shift_hi)) >> 4) * g.bf_n * 3072;
          }
#pragma unroll
          for (int p = 0; p < 3; ++p)
            dst[p][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(brs, off + p * 1024, 0, 0));
        }
      };
      bf16x8 bq[BK / 16][3][TN];
      pa.prime(kbeg, kend);
      pa.load(kmap(0), kend);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const int k = kmap(0);
        bload(bq[ks], k < kend ? k + ks * 16 : kend);
      }
      pa.store(Ax0);
      __syncthreads();
      for (int v0 = 0; v0 < kv_end; v0 += BK) {
        const int kn = kmap(v0 + BK);
        pa.load(kn, kend);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
          bf16x8 a[3][TM];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i) a[p][i] = PA_::frag(Ax0 + p * PA_::PPLANE, wr * WM + i * 32, ks, lane);
          constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
#pragma unroll
          for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                if (PM_BD_DUAL && (t6 & 1))      // odd products to a second accumulator set: twice the independent chains
                  accb[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t6]][i], bq[ks][PB[t6]][j], accb[i][j], 0, 0, 0);
                else
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t6]][i], bq[ks][PB[t6]][j], acc[i][j], 0, 0, 0);
              }
          bload(bq[ks], kn < kend ? kn + ks * 16 : kend);
        }
        __syncthreads();
        if (v0 + BK < kv_end) pa.store(Ax0);
        __syncthreads();
      }
      if (PM_BD_DUAL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += accb[PM_BD_DUAL ? i : 0][PM_BD_DUAL ? j : 0][r];
      }
    } else {
    pa.prime(kbeg, kend);
    pb.prime(kbeg, kend);
    pa.load(kmap(0), kend);
    pb.load(kmap(0), kend);
    char* const Bx1 = Ax0 + PA_::PBYTES;
    pa.store(Ax0);
    pb.store(Bx1);
    __syncthreads();
    for (int v0 = 0; v0 < kv_end; v0 += BK) {
      pa.load(kmap(v0 + BK), kend);            // (past the end: out-of-range offsets, returns 0, no traffic)
      pb.load(kmap(v0 + BK), kend);
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};    // smallest terms first
      if constexpr (PM_TN_DUAL && TM * TN == 1 && BK == 32) {
        // one 32x32 tile per wave: its six products per k-step form ONE dependent MFMA chain; the two k-steps of the
        // tile go to two accumulators (added in the epilogue), so two independent chains interleave
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            a[ks][p] = PA_::frag(Ax0 + p * PA_::PPLANE, wr * WM, ks, lane);
            b[ks][p] = PB_::frag(Bx1 + p * PB_::PPLANE, wc * WN, ks, lane);
          }
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA[t6]], b[0][PB[t6]], acc[0][0], 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA[t6]], b[1][PB[t6]], acc2, 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[p][i] = PA_::frag(Ax0 + p * PA_::PPLANE, wr * WM + i * 32, ks, lane);
#pragma unroll
          for (int j = 0; j < TN; ++j) b[p][j] = PB_::frag(Bx1 + p * PB_::PPLANE, wc * WN + j * 32, ks, lane);
        }
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t6]][i], b[PB[t6]][j], acc[i][j], 0, 0, 0);
      }
      }
      __syncthreads();
      if (v0 + BK < kv_end) {
        pa.store(Ax0);
        pb.store(Bx1);
      }
      __syncthreads();
    }
    }
    if constexpr (PM_TN_DUAL && TM * TN == 1 && BK == 32 && !BD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[r];
    }
  } else {
  sa.init(g.A, g.lda, m0, M, mapA, g.rpe);
  sb.init(g.B, g.ldb, n0, g.N, mapB, g.rpe, g.b_split, (int)(grp * g.b_boff * 4), (int)(g.b_hi * 4));
  sa.prime(kbeg, kend);
  sb.prime(kbeg, kend);
  if (!X6 || producer) {
    sa.load(kbeg, kend);
    sb.load(kbeg, kend);
    if constexpr (X6) { sa.store_x6(Ax0); sb.store_x6(Bx0); }
    else { sa.store(As0); sb.store(Bs0); }
  }

  int buf = 0;
  if constexpr (X6) {
    // Producer waves keep the global loads TWO k-tiles ahead (two register sets): at bf16 rate a k-tile of MFMAs
    // is ~0.3-0.6 us, about one HBM round trip.  Iteration t: producers issue the loads of tile t+2 and split +
    // store tile t+1 (loaded during iteration t-1) into LDS buffer (t+1)&1 while the consumers multiply tile t out
    // of buffer t&1; one barrier per k-tile.
    constexpr int NVA = StA::NV, NVB = StB::NV;
    u32x4 ra[2][NVA], rb[2][NVB];
    const int a_off = StA::x6_lane_off(wr * WM, lane), b_off = StB::x6_lane_off(wc * WN, lane);
    if (producer) { sa.load_to(ra[1], kbeg + BK, kend); sb.load_to(rb[1], kbeg + BK, kend); }
    __syncthreads();
    auto phase = [&](int k0, u32x4 (&la)[NVA], u32x4 (&lb)[NVB], u32x4 (&na)[NVA], u32x4 (&nb)[NVB], int cb) {
      // la/lb: register set the loads of tile k0+2BK go to; na/nb: set holding tile k0+BK; cb: LDS buffer of tile k0
      if (producer) {
        // unconditional (past the end every offset is out of range and the load returns 0 without touching
        // memory): a conditional issue would make the vmcnt bookkeeping ambiguous and drain the prefetch
        sa.load_to(la, k0 + 2 * BK, kend); sb.load_to(lb, k0 + 2 * BK, kend);
        if (k0 + BK < kend) {
          sa.store_x6_from(na, Ax0 + (cb ^ 1) * StA::XBYTES);
          sb.store_x6_from(nb, Bx0 + (cb ^ 1) * StB::XBYTES);
        }
      } else {
        const char* as = Ax0 + cb * StA::XBYTES + a_off;
        const char* bs = Bx0 + cb * StB::XBYTES + b_off;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
          bf16x8 a[3][TM], b[3][TN];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[p][i] = StA::x6_frag(as + p * StA::XPLANE + i * StA::XSTEP_I + ks * StA::XSTEP_K);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[p][j] = StB::x6_frag(bs + p * StB::XPLANE + j * StB::XSTEP_I + ks * StB::XSTEP_K);
          }
          // smallest terms first: (3,1) (2,2) (1,3) | (2,1) (1,2) | (1,1)
          constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]][j], acc[i][j], 0, 0, 0);
        }
      }
      __syncthreads();
    };
    for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
      phase(k0, ra[0], rb[0], ra[1], rb[1], 0);
      if (k0 + BK < kend) phase(k0 + BK, ra[1], rb[1], ra[0], rb[0], 1);
    }
  } else {
  __syncthreads();
  // bias gradient riding along with a weight gradient dW = dy^T x (PmGemmDesc.a_colsum): the waves of the first column
  // of the first column tile add up the A fragments they feed to the matrix core anyway (lane (li, lh) holds
  // A[k + lh, m + li]); one atomic per output row and workgroup in the epilogue
  const bool do_cs = TA && MODE == 0 && g.colsum_a != nullptr && n0 == 0 && wc == 0;
  float csa[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) csa[i] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      sa.load(k0 + BK, kend);
      sb.load(k0 + BK, kend);
    }
    const float* as = As0 + buf * (BK * LDA_S) + wr * WM + li + lh * LDA_S;
    const float* bs = Bs0 + buf * (BK * LDB_S) + wc * WN + li + lh * LDB_S;
    // fragments of step kk+2 are read while the MFMAs of step kk run (explicit LDS software pipeline)
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = bs[j * 32];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
      if (kk + 2 < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[nxt][i] = as[(kk + 2) * LDA_S + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[nxt][j] = bs[(kk + 2) * LDB_S + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the next step's ds_reads ahead of this step's MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      if constexpr (TA && MODE == 0) {
        if (do_cs) {
#pragma unroll
          for (int i = 0; i < TM; ++i) csa[i] += a[cur][i];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      sa.store(As0 + (buf ^ 1) * (BK * LDA_S));
      sb.store(Bs0 + (buf ^ 1) * (BK * LDB_S));
    }
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (TA && MODE == 0) {
    if (do_cs) {
      pm_turn_enter(g.gate, my_turn);
      in_turn = true;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float v = csa[i] + __shfl_xor(csa[i], 32);
        const int row = m0 + wr * WM + i * 32 + li;
        if (lh == 0 && row < M) atomicAdd(g.colsum_a + row, v);
      }
    }
  }
  }

  }
  if (producer) return;
#undef GEMM_SKIP
  if (!in_turn) pm_turn_enter(g.gate, my_turn);
  // Epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
  const bool atomic = gridDim.z > 1;
  const bool accum = (g.flags & PM_GEMM_ACCUM) != 0;
  const bool relu = (g.flags & PM_GEMM_RELU) != 0;
  const bool relu_add = (g.flags & PM_GEMM_RELU_ADD) != 0;      // C = C + relu(acc + bias): residual GCL tail, eval mode
  const bool add_bias = g.bias != nullptr && zs == 0;
  const int32_t* mapC = TA ? nullptr : g.rowmap;
  const bool stats = g.colstats != nullptr;
  double cs[TN], cq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { cs[j] = 0.0; cq[j] = 0.0; }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= M) continue;
      float* crow = g.C + map_row(mapC, g.rpe, row) * g.ldc;
      bool at = atomic;
      if (g.c_split > 0) {                   // stacked C (weight gradient): shared rows collect every group's term
        const bool shared = row >= g.c_split;
        crow += shared ? g.c_hi : (int64_t)grp * g.c_boff;
        at = atomic || (shared && g.ngroups > 1);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * WN + j * 32 + li;
        if (col >= g.N) continue;
        float v = acc[i][j][r];
        if (add_bias) v += g.bias[col];
        if (at) atomicAdd(crow + col, v);
        else {
          if (relu_add) v = fmaxf(v, 0.f) + crow[col];
          else {
            if (accum) v += crow[col];
            if (relu) v = fmaxf(v, 0.f);
          }
          crow[col] = v;
          if (stats) { cs[j] += (double)v; cq[j] += (double)v * (double)v; }
        }
      }
    }
  }
  // Column statistics of the tile for the BatchNorm that follows (nn.BatchNorm1d over the node rows, model.py:203):
  // fp64 partial sums per lane, the two half-waves (same columns, different rows) combined by a lane exchange,
  // one fp64 atomic per column and per wave-row.
  if (stats && !atomic) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const double s2 = cs[j] + __shfl_xor(cs[j], 32), q2 = cq[j] + __shfl_xor(cq[j], 32);
      const int col = n0 + wc * WN + j * 32 + li;
      if (lh == 0 && col < g.N) {              // replica by row panel: <= ~64 serialized atomics per address
        double* dst = g.colstats + (int64_t)(((m0 / BM) * 2 + wr) % PM_BN_REPL) * 2 * g.N;
        atomicAdd(dst + col, s2);
        atomicAdd(dst + g.N + col, q2);
      }
    }
  }
  pm_turn_leave(g.gate, my_turn);
}

template <int BM, int BN, int BK, int WVM, int WVN, bool TA, bool TB, bool VA, bool VB, int MODE>
static void launch_one(dim3 grid, hipStream_t st, const GemmArgs& g) {
  using StA = TileStage<BM, BK, 64 * WVM * WVN, !TA, VA>;
  using StB = TileStage<BN, BK, 64 * WVM * WVN, TB, VB>;
  size_t lds;
  if constexpr (MODE == 3)
    lds = (size_t)PlaneStage<BM, BK, 64 * WVM * WVN, !TA>::PBYTES;
  else if constexpr (MODE == 2)
    lds = (size_t)(PlaneStage<BM, BK, 64 * WVM * WVN, !TA>::PBYTES + PlaneStage<BN, BK, 64 * WVM * WVN, TB>::PBYTES);
  else
    lds = MODE == 1 ? (size_t)2 * (StA::XBYTES + StB::XBYTES) : sizeof(float) * 2 * BK * (StA::LD + StB::LD);
  auto kern = k_gemm<BM, BN, BK, WVM, WVN, TA, TB, VA, VB, MODE>;
  static bool attr_done = false;             // > 64 KiB of dynamic LDS needs the attribute (once per instantiation)
  if (lds > 64 * 1024 && !attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WVM * WVN * (MODE == 1 ? 2 : 1)), lds, st, g);
}
template <int BM, int BN, int BK, int WVM, int WVN, bool TA, bool TB, int MODE>
static void launch_v(bool va, bool vb, dim3 grid, hipStream_t st, const GemmArgs& g) {
  if constexpr (MODE != 0) launch_one<BM, BN, BK, WVM, WVN, TA, TB, true, true, MODE>(grid, st, g);
  else if (va && vb) launch_one<BM, BN, BK, WVM, WVN, TA, TB, true, true, 0>(grid, st, g);
  else if (va) launch_one<BM, BN, BK, WVM, WVN, TA, TB, true, false, 0>(grid, st, g);
  else if (vb) launch_one<BM, BN, BK, WVM, WVN, TA, TB, false, true, 0>(grid, st, g);
  else launch_one<BM, BN, BK, WVM, WVN, TA, TB, false, false, 0>(grid, st, g);
}
template <int BM, int BN, int BK, int WVM, int WVN, int MODE>
static void launch_t(int ta, int tb, bool va, bool vb, dim3 grid, hipStream_t st, const GemmArgs& g) {
  if (!ta && !tb) launch_v<BM, BN, BK, WVM, WVN, false, false, MODE>(va, vb, grid, st, g);
  else if (!ta && tb) launch_v<BM, BN, BK, WVM, WVN, false, true, MODE>(va, vb, grid, st, g);
  else if constexpr (MODE != 3) launch_v<BM, BN, BK, WVM, WVN, true, false, MODE>(va, vb, grid, st, g);
}

enum { PM_GEMM_NCFG = 11 };                   // 0..3 fp32 MFMA, 4..7 split mode, 8 / 10 pre-split planes, 9 planes with B direct
static const int CFG_BM[PM_GEMM_NCFG] = {64, 128, 64, 128, 128, 128, 64, 128, 64, 64, 128};
static const int CFG_BN[PM_GEMM_NCFG] = {64, 128, 64, 128, 128, 64, 64, 128, 64, 128, 128};
static const int CFG_BK[PM_GEMM_NCFG] = {16, 16, 32, 32, 16, 16, 32, 32, 32, 32, 32};   // (config 9: 64 for the long-K forward product)

// Tile configuration: an explicit override (pm_gemm_force_config, for A/B timing in one process) or the shape rule.
static int g_forced_cfg = -1;
extern "C" int pm_gemm_force_config(int32_t cfg) { g_forced_cfg = (cfg >= 0 && cfg < 8) ? cfg : -1; return PM_OK; }
static int pick_config(int transA, int M, int N, int K, bool x6_ok = false, bool ungrouped = false) {
  if (g_forced_cfg >= 0 && (g_forced_cfg < 4 || x6_ok)) return g_forced_cfg;
  // Measured on MI355X (tools/bench_gemm.py, shapes of the training step, interleaved A/B in one process):
  //  - NN / NT (the node dimension is M): 64x64 tiles win or tie everywhere: 4 workgroups per CU = 4 waves per
  //    SIMD hide the LDS / barrier latency; the panels they re-read sit in L2 / Infinity Cache.  With a long K
  //    (>= 1024: GCL forward, chord encoder, d(x_L)) BK = 32 halves the barriers per flop: 107-119 TFLOP/s
  //    against 102-112; with K = d = 256 the BK = 16 variant stays ahead (110-114 against 106-108).
  //  - TN (weight gradients, K = node dimension, split-K): in isolation 128x128 and 64x64 tiles tie (98-108
  //    TFLOP/s), inside the training step 64x64x32 tiles with ~1024 workgroups (4 per CU, split-K sized for it)
  //    are 18 % faster than 128x128x16 with 512 (65.9 us against 80.3 us average over the 30 TN launches).
  //  8-wave 64x256 / 256x64 shapes (operand streamed exactly once) measured 5-20 % slower at these sizes.
  //  - Split mode (x6, needs 16-byte aligned operands) wins on every large shape of the step (same harness):
  //    NN/NT K >= 1024: 128x128x32 126-173 TFLOP/s (fp32 mode 96-118); NN/NT short K: 128x128x16 114-140 (90-108);
  //    TN: 128x64x16 123-155 (77-107).  Small problems stay on the fp32 tiles (fewer, cheaper workgroups).
  //    Inside the training step (row-gathered, grouped launches, one 8-wave workgroup per CU) the gain did not
  //    materialise (10.71 ms against 10.78 ms per step), so split mode is opt-in: PM_GEMM_SPLIT=1 or a forced config.
  constexpr bool split_on = false;
  // default: the large UNGROUPED NN / NT products (chord encoder / decoder and their input gradients) run in split
  // mode (78-90 us against 99-112 us in the step); weight gradients stay on the fp32 tiles (split mode: 132 against 106 us)
  constexpr bool split_ungrouped = true;
  // (N < 128: the 128-wide split tiles would be partly empty — the duration un-embedding, N = 99: 60 us against 46 us on the fp32 tiles)
  // ... since round 4 the large weight gradients too: they now run beside the head chain / in the step's tail, not beside the
  // one-workgroup-per-CU GCL kernels, and there the split tiles win (step 4.892 -> 4.859 ms; PM_GEMM_SPLIT_TN=0: fp32 tiles)
  constexpr bool split_tn = true;
  if ((split_on || (split_ungrouped && ungrouped && !transA && N >= 128) || (split_tn && transA)) && x6_ok && (double)M * N * K >= 1.0e9) {
    constexpr int tn_cfg = 5;   // (development A/B: 4, 5, 7)
    if (transA) return (tn_cfg == 4 || tn_cfg == 7) ? tn_cfg : 5;
    return K >= 1024 ? 7 : 4;
  }
  return K >= 1024 ? 2 : 0;
}

extern "C" int pm_gemm_config(int32_t transA, int32_t M, int32_t N, int32_t K) { return pick_config(transA, M, N, K); }

static inline bool ldc_contig(const PmGemmDesc* q, int N) { return q->ldc == N; }

extern "C" int pm_gemm_f32_desc(const PmGemmDesc* q, pm_stream_t stream) {
  if (!q) return PM_E_INVALID;
  const int transA = q->transA, transB = q->transB, M = q->M, N = q->N, K = q->K, n_groups = q->n_groups;
  if (n_groups < 1 || n_groups > 65535) return PM_E_INVALID;
  if (!q->A || !q->B || !q->C || M <= 0 || N <= 0 || K <= 0 || q->lda <= 0 || q->ldb <= 0 || q->ldc <= 0) return PM_E_INVALID;
  if (transA && transB) return PM_E_UNSUPPORTED;
  if ((q->rowmap || q->dyn_entries) && q->rows_per_entry <= 0) return PM_E_INVALID;
  if (q->b_split_rows < 0 || q->c_split_rows < 0 || (q->c_split_rows > 0 && !transA)) return PM_E_INVALID;
  // operands are addressed with 32-bit byte offsets (buffer loads): each must span < 2 GiB
  if (!q->rowmap && ((int64_t)(transA ? K : M) * q->lda * 4 >= ((int64_t)1 << 31) ||
                     (int64_t)(transB ? N : K) * q->ldb * 4 >= ((int64_t)1 << 31))) return PM_E_UNSUPPORTED;
  if (q->b_split_rows > 0 && (q->b_group_stride * (n_groups - 1) * 4 >= ((int64_t)1 << 30) ||
                              q->b_shared_off * 4 >= ((int64_t)1 << 30) || q->b_shared_off < 0)) return PM_E_UNSUPPORTED;
  GemmArgs g;
  g.A = q->A; g.B = q->B; g.C = q->C; g.bias = q->bias; g.rowmap = q->rowmap; g.dyn_entries = q->dyn_entries;
  g.M = M; g.N = N; g.K = K; g.lda = q->lda; g.ldb = q->ldb; g.ldc = q->ldc;
  g.rpe = q->rows_per_entry > 0 ? q->rows_per_entry : 1; g.flags = q->flags;
  g.a_boff = q->a_group_stride; g.b_boff = q->b_group_stride; g.c_boff = q->c_group_stride;
  g.bias_boff = q->bias_group_stride; g.map_boff = q->map_group_stride; g.dyn_boff = q->dyn_group_stride;
  g.b_split = q->b_split_rows; g.b_hi = q->b_shared_off; g.c_split = q->c_split_rows; g.c_hi = q->c_shared_off;
  g.colstats = q->col_stats;
  g.colsum_a = nullptr;
  if (q->col_stats && (transA || (q->flags & PM_GEMM_ACCUM))) return PM_E_INVALID;   // stats of plainly stored tiles only
  // float4 staging needs 16-byte aligned rows and a contiguous extent that is a multiple of 4
  const bool va = ((uintptr_t)q->A % 16 == 0) && (q->lda % 4 == 0) && ((transA ? M : K) % 4 == 0) && (q->a_group_stride % 4 == 0);
  const bool vb = ((uintptr_t)q->B % 16 == 0) && (q->ldb % 4 == 0) && ((transB ? K : N) % 4 == 0) &&
                  (q->b_group_stride % 4 == 0) && (q->b_shared_off % 4 == 0);
  // the fp32 tile kernels (configs 0-3) stage with 16-byte loads whenever the rows are 16-byte aligned: an extent that is
  // not a multiple of 4 makes the last float4 of a row reach into the row's own padding / next columns (lda % 4 == 0: never
  // past the row) — K tails are zeroed when staged, M / N tails only reach accumulator rows / columns that are not stored
  const bool va_rows = ((uintptr_t)q->A % 16 == 0) && (q->lda % 4 == 0) && (q->a_group_stride % 4 == 0);
  const bool vb_rows = ((uintptr_t)q->B % 16 == 0) && (q->ldb % 4 == 0) && (q->b_group_stride % 4 == 0) && (q->b_shared_off % 4 == 0);
  const bool planes = q->operand_planes != 0;
  if (planes) {                              // bf16 planes: 16-byte chunks = 8 elements along the contiguous extent
    if (((uintptr_t)q->A % 16) || ((uintptr_t)q->B % 16) || (q->lda % 8) || (q->ldb % 8) || ((transA ? M : K) % 8) ||
        ((transB ? K : N) % 8) || (q->a_group_stride % 8) || (q->b_group_stride % 8) || (q->b_shared_off % 8) ||
        (q->a_plane_stride % 8) || (q->b_plane_stride % 8) || q->a_plane_stride <= 0 || q->b_plane_stride <= 0)
      return PM_E_INVALID;
  }
  g.a_ps = q->a_plane_stride; g.b_ps = q->b_plane_stride;
  g.bfrag = reinterpret_cast<const char*>(q->b_frag); g.bf_n = transB ? q->ldb / 16 : q->ldb / 32;
  g.cls_ptr = q->class_ptr; g.cls_blk = q->class_block; g.cls_dim = 0;
  // planes mode: 64x64x32 tiles (measured in the step: 128x64x32 85 us, 128x128x32 97-130 us against 70-80 us)
  // B direct (config 9): B given additionally as fragment-major planes; forward / input-gradient products whose tiles
  // are aligned with the fragment grid (and, with row classes, with the 128-wide column tile)
  constexpr bool bdirect_on = true;
  const bool bdirect = planes && q->b_frag && bdirect_on && !transA && N % 128 == 0 && K % 32 == 0 && q->ldb % 32 == 0 &&
                       (n_groups == 1 || q->b_split_rows > 0) && q->b_split_rows % 32 == 0 &&
                       (q->b_split_rows == 0 || (q->b_group_stride % q->ldb == 0 && q->b_shared_off % q->ldb == 0)) &&
                       ((uintptr_t)q->b_frag % 16 == 0) && (!q->class_ptr || q->class_block % 128 == 0);
  const int cfg = planes ? (bdirect ? 9 : 8) : pick_config(transA, M, N, K, va && vb, n_groups == 1 && !q->rowmap);
  const int BM = CFG_BM[cfg], BN = CFG_BN[cfg], BK = (cfg == 9 && !transB) ? 64 : CFG_BK[cfg];
  if (q->class_ptr && q->class_block > 0 && q->rowmap && q->dyn_entries && q->rows_per_entry == 1) {
    const int blk = q->class_block;           // which dimension carries the [track | onset | next | x] blocks
    if (transA) { if (M == 4 * blk && blk % BM == 0) g.cls_dim = 3; }
    else if (transB) { if (N == 4 * blk && blk % BN == 0) g.cls_dim = 2; }
    else if (planes && K == 4 * blk && blk % BK == 0 && q->split_k == 1) g.cls_dim = 1;
  }
  // bias gradient with the weight gradient: the fp32 tile kernels fold it into the product, any other route takes the
  // column-sum kernel (same result, one more launch)
  if (q->a_colsum) {
    if (!transA || q->rowmap || q->dyn_entries || n_groups != 1 || planes) return PM_E_INVALID;
    if (cfg <= 3) g.colsum_a = q->a_colsum;
    else {
      const int rc = pm_colsum_acc(q->A, K, M, q->lda, q->a_colsum, stream);
      if (rc != PM_OK) return rc;
    }
  }
  g.ntm = (int)pm_cdiv(M, BM); g.ntn = (int)pm_cdiv(N, BN);
  const int64_t tiles = (int64_t)g.ntm * g.ntn;
  // a gathered dimension with a device-side count per group: the groups PARTITION the rows, so the live work is
  // about 1/n_groups of the bound the grid is sized for
  const bool partitioned = q->dyn_entries && n_groups > 1 && (q->flags & PM_GEMM_PARTITION);
  g.ngroups = n_groups;
  g.packed = (partitioned && q->rowmap && !transA && n_groups <= 16) ? 1 : 0;   // row lists that partition <= M rows
  int split_k = q->split_k;
  if (split_k <= 0) {                                                  // auto: fill the 256 CUs when K is long
    split_k = 1;
    if (transA && tiles * n_groups < 768) {
      // (768 since the large weight gradients run on the 128x64 split tiles: 4.637 against 4.667 ms per step with 1024; 256: 5.18 ms)
      constexpr int tn_target = 768;
      split_k = (int)(tn_target / (tiles * n_groups));
      const int maxs = (int)pm_cdiv(partitioned ? K / n_groups : K, 8 * BK);
      if (split_k > maxs) split_k = maxs;
      if (split_k < 1) split_k = 1;
    }
  }
  int flags = q->flags;
  hipStream_t st = (hipStream_t)stream;
  // Latency-bound small products (the heads: a few dozen tiles, K of 256 or more): the K loop is the critical path, so
  // split it over the otherwise idle CUs.  C is cleared first and accumulated with atomics (bias by slice 0);
  // only for a plainly stored contiguous C (no ReLU / statistics / row map / accumulate).
  // (a C with a row pitch only when the caller vouches that it is already zero: the clear here is one contiguous memset)
  if (!transA && q->split_k == 1 && n_groups == 1 && tiles <= 64 && K >= 256 && (ldc_contig(q, N) || (flags & PM_GEMM_ZEROED)) &&
      !(flags & (PM_GEMM_RELU | PM_GEMM_ACCUM | PM_GEMM_RELU_ADD)) && !q->col_stats && !q->rowmap && !planes) {
    split_k = K / 64 < 8 ? K / 64 : 8;
    if (split_k > 1) {
      if (!(flags & PM_GEMM_ZEROED)) hipMemsetAsync(q->C, 0, sizeof(float) * (size_t)M * N, st);
      flags |= PM_GEMM_ACCUM;
      g.flags = flags;
    } else split_k = 1;
  }
  if (split_k > 1 && ((flags & (PM_GEMM_RELU | PM_GEMM_RELU_ADD)) || !(flags & PM_GEMM_ACCUM))) return PM_E_INVALID;  // needs += semantics
  if (g.c_split > 0 && n_groups > 1 && !(flags & PM_GEMM_ACCUM)) return PM_E_INVALID;
  const int kper = (int)pm_cdiv(pm_cdiv(K, split_k), BK) * BK;
  g.kper = kper;
  if (!(transA && q->dyn_entries)) split_k = (int)pm_cdiv(K, kper);     // (device-side K: the kernel re-derives kper)
  dim3 grid((unsigned)tiles, (unsigned)n_groups, (unsigned)split_k);
  if (g.packed) grid = dim3((unsigned)((g.ntm + n_groups) * g.ntn), 1, (unsigned)split_k);
  // deterministic mode: K slices, groups that share stacked rows, column statistics and the bias gradient add in turn
  g.gate = (split_k > 1 || g.colstats || g.colsum_a || (g.c_split > 0 && n_groups > 1)) ? pm_det_gate(st) : nullptr;
  const double work = 2.0 * M * N * K * (partitioned ? 1.0 : (double)n_groups);
  const int pe = pm_prof_open(st, PM_PROF_GEMM0 + cfg * 3 + (transA ? 2 : (transB ? 1 : 0)), work);
  switch (cfg) {
    case 0: launch_t<64, 64, 16, 2, 2, 0>(transA, transB, va_rows, vb_rows, grid, st, g); break;
    case 1: launch_t<128, 128, 16, 2, 2, 0>(transA, transB, va_rows, vb_rows, grid, st, g); break;
    case 2: launch_t<64, 64, 32, 2, 2, 0>(transA, transB, va_rows, vb_rows, grid, st, g); break;
    case 3: launch_t<128, 128, 32, 2, 2, 0>(transA, transB, va_rows, vb_rows, grid, st, g); break;
    case 4: launch_t<128, 128, 16, 2, 2, 1>(transA, transB, va, vb, grid, st, g); break;
    case 5: launch_t<128, 64, 16, 2, 2, 1>(transA, transB, va, vb, grid, st, g); break;
    case 6: launch_t<64, 64, 32, 2, 2, 1>(transA, transB, va, vb, grid, st, g); break;
    case 7: launch_t<128, 128, 32, 2, 2, 1>(transA, transB, va, vb, grid, st, g); break;
    case 8: launch_t<64, 64, 32, 2, 2, 2>(transA, transB, va, vb, grid, st, g); break;
    case 9:        // B direct: k-tiles of 64 for the forward (K = 4d: half the barriers, 62.2 -> 59.8 us), of 32 for the input gradient (K = d)
      if (transB) launch_t<64, 128, 32, 1, 4, 3>(transA, transB, va, vb, grid, st, g);
      else launch_t<64, 128, 64, 1, 4, 3>(transA, transB, va, vb, grid, st, g);
      break;
    default: launch_t<128, 128, 32, 2, 2, 2>(transA, transB, va, vb, grid, st, g); break;
  }
  pm_prof_close(st, pe);
  return pm_check_launch();
}

extern "C" int pm_gemm_f32_grouped(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda,
                                   const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias, int flags,
                                   int split_k, const int32_t* rowmap, int32_t rows_per_entry,
                                   const int32_t* dyn_entries, int32_t n_groups, int64_t a_group_stride,
                                   int64_t b_group_stride, int64_t c_group_stride, int64_t bias_group_stride,
                                   int32_t map_group_stride, int32_t dyn_group_stride, pm_stream_t stream) {
  PmGemmDesc q;
  q.transA = transA; q.transB = transB; q.M = M; q.N = N; q.K = K; q.A = A; q.lda = lda; q.B = B; q.ldb = ldb;
  q.C = C; q.ldc = ldc; q.bias = bias; q.flags = flags; q.split_k = split_k; q.rowmap = rowmap;
  q.rows_per_entry = rows_per_entry; q.dyn_entries = dyn_entries; q.n_groups = n_groups;
  q.a_group_stride = a_group_stride; q.b_group_stride = b_group_stride; q.c_group_stride = c_group_stride;
  q.bias_group_stride = bias_group_stride; q.map_group_stride = map_group_stride; q.dyn_group_stride = dyn_group_stride;
  q.b_split_rows = 0; q.b_shared_off = 0; q.c_split_rows = 0; q.c_shared_off = 0; q.col_stats = nullptr;
  q.operand_planes = 0; q.a_plane_stride = 0; q.b_plane_stride = 0; q.a_colsum = nullptr;
  q.class_ptr = nullptr; q.class_block = 0; q.b_frag = nullptr;
  return pm_gemm_f32_desc(&q, stream);
}

extern "C" int pm_gemm_f32(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda,
                           const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias, int flags,
                           int split_k, const int32_t* rowmap, int32_t rows_per_entry, const int32_t* dyn_entries,
                           pm_stream_t stream) {
  return pm_gemm_f32_grouped(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, split_k, rowmap,
                             rows_per_entry, dyn_entries, 1, 0, 0, 0, 0, 0, 0, stream);
}

// fp32 -> three bf16 planes (x = x1 + x2 + x3 exactly): the operand format of the planes mode.  Used once per step on
// the GCL weights; activations are written as planes directly by the kernels that produce them.
__global__ void __launch_bounds__(256) k_split_planes(const float* __restrict__ src, int64_t n4, uint16_t* __restrict__ planes,
                                                      int64_t plane_stride) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    pm_store_planes4(planes, plane_stride, i * 4, v.x, v.y, v.z, v.w);
  }
}
extern "C" int pm_split_planes(const float* src, int64_t n, uint16_t* planes, int64_t plane_stride, pm_stream_t stream) {
  if (!src || !planes || n <= 0 || (n & 3) || plane_stride < n || (plane_stride & 3) || ((uintptr_t)src % 16) ||
      ((uintptr_t)planes % 8))
    return PM_E_INVALID;
  int64_t grid = pm_cdiv(n / 4, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_split_planes, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, src, n / 4, planes, plane_stride);
  return pm_check_launch();
}

// Fragment-major planes of a weight matrix W [rows, cols] (fp32) for the B-direct GEMM mode: per (32-wide tile of the
// n dimension, 16-wide k-step, plane) one contiguous 1 KiB block holding, for lane l, the 8 bf16 of
// (n = tile*32 + l % 32, k = step*16 + 8 * (l / 32) .. + 8) — exactly the B operand of v_mfma_f32_32x32x16_bf16.
//   kind 0 (the product uses W as B[n][k], transB): n = W row, k = W column; blocks ordered [row tile][k-step][plane]
//   kind 1 (the product uses W as B[k][n])        : k = W row, n = W column; blocks ordered [k-step][column tile][plane]
// rows % 32 == 0 and cols % 32 == 0; `n_mats` matrices `src_stride` floats apart go to blocks `dst_stride` bf16 apart.
// H2: the fp16 pair format of common.h (two planes of W * w_scale; plane 2 of the block is left alone)
template <bool H2>
__global__ void __launch_bounds__(256) k_split_planes_frag(const float* __restrict__ W, int rows, int cols, int kind,
                                                           int64_t src_stride, int64_t dst_stride,
                                                           uint16_t* __restrict__ out, float w_scale, unsigned* clamps) {
  bool cut = false;                                          // (H2: a weight times w_scale beyond fp16's range)
  const float* src = W + (int64_t)blockIdx.y * src_stride;
  uint16_t* dst = out + (int64_t)blockIdx.y * dst_stride;
  const int64_t chunks = (int64_t)rows * cols / 8;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * blockDim.x) {
    float x[8];
    int64_t blk;
    int lane;
    if (kind == 0) {                                       // chunk = 8 consecutive columns of one row
      const int n = (int)(c / (cols / 8)), k0 = (int)(c % (cols / 8)) * 8;
      const float4 u = *reinterpret_cast<const float4*>(src + (int64_t)n * cols + k0);
      const float4 w = *reinterpret_cast<const float4*>(src + (int64_t)n * cols + k0 + 4);
      x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = w.x; x[5] = w.y; x[6] = w.z; x[7] = w.w;
      blk = (int64_t)(n >> 5) * (cols / 16) + (k0 >> 4);
      lane = ((k0 >> 3) & 1) * 32 + (n & 31);
    } else {                                               // chunk = 8 consecutive rows of one column (n fastest: coalesced)
      const int n = (int)(c % cols), k0 = (int)(c / cols) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = src[(int64_t)(k0 + e) * cols + n];
      blk = (int64_t)(k0 >> 4) * (cols / 32) + (n >> 5);
      lane = ((k0 >> 3) & 1) * 32 + (n & 31);
    }
    unsigned p1[4], p2[4], p3[4];
    uint16_t* o = dst + blk * 1536 + lane * 8;             // 3 planes x 512 bf16 per block
    if constexpr (H2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) pm_split2h_pair(pm_clamp_f16(x[2 * e] * w_scale, cut), pm_clamp_f16(x[2 * e + 1] * w_scale, cut), p1[e], p2[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) pm_split3_pair(x[2 * e], x[2 * e + 1], p1[e], p2[e], p3[e]);
      *reinterpret_cast<u32x4*>(o + 1024) = u32x4{p3[0], p3[1], p3[2], p3[3]};
    }
    *reinterpret_cast<u32x4*>(o) = u32x4{p1[0], p1[1], p1[2], p1[3]};
    *reinterpret_cast<u32x4*>(o + 512) = u32x4{p2[0], p2[1], p2[2], p2[3]};
  }
  if (H2 && cut && clamps) atomicAdd(clamps, 1u);
}
static int split_planes_frag_impl(const float* W, int32_t rows, int32_t cols, int32_t kind, int32_t n_mats,
                                  int64_t src_stride, int64_t dst_stride, float w_scale, uint16_t* out, pm_stream_t stream) {
  if (!W || !out || rows <= 0 || cols <= 0 || (rows % 32) || (cols % 32) || (kind != 0 && kind != 1) || n_mats <= 0 ||
      ((uintptr_t)W % 16) || ((uintptr_t)out % 16) || (src_stride % 4) || (dst_stride % 8))
    return PM_E_INVALID;
  int64_t grid = pm_cdiv((int64_t)rows * cols / 8, 256);
  if (grid > 2048) grid = 2048;
  if (w_scale > 0.f)
    hipLaunchKernelGGL(k_split_planes_frag<true>, dim3((unsigned)grid, (unsigned)n_mats), dim3(256), 0, (hipStream_t)stream, W,
                       rows, cols, kind, src_stride, dst_stride, out, w_scale, pm_h2_clamp_word());
  else
    hipLaunchKernelGGL(k_split_planes_frag<false>, dim3((unsigned)grid, (unsigned)n_mats), dim3(256), 0, (hipStream_t)stream, W,
                       rows, cols, kind, src_stride, dst_stride, out, 0.f, nullptr);
  return pm_check_launch();
}
extern "C" int pm_split_planes_frag(const float* W, int32_t rows, int32_t cols, int32_t kind, int32_t n_mats,
                                    int64_t src_stride, int64_t dst_stride, uint16_t* out, pm_stream_t stream) {
  return split_planes_frag_impl(W, rows, cols, kind, n_mats, src_stride, dst_stride, 0.f, out, stream);
}
extern "C" int pm_split_planes_frag_h2(const float* W, int32_t rows, int32_t cols, int32_t kind, int32_t n_mats,
                                       int64_t src_stride, int64_t dst_stride, float w_scale, uint16_t* out, pm_stream_t stream) {
  if (!(w_scale > 0.f)) return PM_E_INVALID;
  return split_planes_frag_impl(W, rows, cols, kind, n_mats, src_stride, dst_stride, w_scale, out, stream);
}
// max |x| of a tensor as float bits (PmH2.absmax_in): one atomic per wave
__global__ void __launch_bounds__(256) k_absmax(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
  __shared__ unsigned sm;
  if (threadIdx.x == 0) sm = 0u;
  __syncthreads();
  float m = 0.f;
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
  pm_absmax_block(out, m, &sm);
}
extern "C" int pm_absmax(const float* x, int64_t n, uint32_t* out, pm_stream_t stream) {
  if (!x || !out || n <= 0 || ((uintptr_t)x % 16)) return PM_E_INVALID;
  int64_t grid = pm_cdiv(n / 4 + 1, 256 * 4);
  if (grid > 512) grid = 512;
  hipLaunchKernelGGL(k_absmax, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n, out);
  return pm_check_launch();
}

// ---------------------------------------------------------------- launch-duration profiler (see prof.h)
PmProfState g_pm_prof = {false, 0, 0, nullptr, ~0ull, 1, {0}};

// Which launches pm_prof_begin .. pm_prof_end bracket: classes of `class_mask`, every `stride`-th launch of each.
extern "C" int pm_prof_configure(int64_t class_mask, int32_t stride) {
  if (stride < 1) return PM_E_INVALID;
  g_pm_prof.mask = (uint64_t)class_mask;
  g_pm_prof.stride = stride;
  return PM_OK;
}


extern "C" int pm_prof_begin(int32_t max_events) {
  PmProfState& p = g_pm_prof;
  if (max_events <= 0) return PM_E_INVALID;
  if (p.cap < max_events) {
    PmProfEvent* ev = (PmProfEvent*)realloc(p.ev, sizeof(PmProfEvent) * max_events);
    if (!ev) return PM_E_INVALID;
    p.ev = ev;
    for (int i = p.cap; i < max_events; ++i) {
      if (hipEventCreate(&p.ev[i].a) != hipSuccess || hipEventCreate(&p.ev[i].b) != hipSuccess) return PM_E_LAUNCH;
    }
    p.cap = max_events;
  }
  p.n = 0;
  for (int c = 0; c < PM_PROF_NCLASS; ++c) p.seen[c] = 0;
  p.on = true;
  return PM_OK;
}
// Stops recording, waits for the recorded events and sums them per class: ms[c], work[c], count[c] (c < PM_PROF_NCLASS).
extern "C" int pm_prof_end(double* ms, double* work, int64_t* count) {
  PmProfState& p = g_pm_prof;
  p.on = false;
  if (!ms || !work || !count) return PM_E_INVALID;
  for (int c = 0; c < PM_PROF_NCLASS; ++c) { ms[c] = 0; work[c] = 0; count[c] = 0; }
  for (int i = 0; i < p.n; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(p.ev[i].b) != hipSuccess || hipEventElapsedTime(&t, p.ev[i].a, p.ev[i].b) != hipSuccess)
      return PM_E_LAUNCH;
    ms[p.ev[i].cls] += t; work[p.ev[i].cls] += p.ev[i].work; count[p.ev[i].cls] += 1;
  }
  p.n = 0;
  return PM_OK;
}
