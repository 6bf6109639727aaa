// wide.h — host entry points of wide.hip (the MFMA pipelines of gcl.hip / linear.hip for 512-wide layers); called by
// the C-ABI functions of those files when d = 512, argument meaning as theirs.  Internal to the library.
#pragma once
#include "common.h"

// h = A'(x) @ [W_t; W_4; W_5; root] + bias in one kernel (pm_gcl_forward_fused at d = 512).  `a_planes_in` non-null:
// the aggregate is not built by the kernel but read from these A' planes (dense graphs: pm_segreduce_fwd_planes first).
int pm_wide_gcl_forward(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                        float dropout_p, uint32_t seed, uint32_t layer_uid, const uint16_t* w_frag, const float* bias,
                        int32_t use_classes, float* h, double* col_stats, uint16_t* planes, int64_t plane_stride,
                        const uint16_t* a_planes_in, hipStream_t st, const PmH2* h2 = nullptr);
// dA' = dh @ [W_t; W_4; W_5; root]^T (pm_gcl_input_grad_fused at d = 512)
int pm_wide_gcl_input_grad(const uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan, int32_t N, int32_t E,
                           int32_t G, const uint16_t* w_frag_t, int32_t use_classes, float* dA, hipStream_t st,
                           const float* dh_scale = nullptr, float w_scale = 1.f);   // (dh_scale: planes in the fp16 pair format)
// C[N, Nout] = X[N, 512] @ W (+ bias), Nout a multiple of 512 (pm_rows_times_weight at K = 512)
int pm_wide_rows_times_weight(const float* X, int32_t ldx, int32_t N, const uint16_t* w_frag, int32_t kind,
                              int32_t w_tiles, int32_t Nout, const float* bias, float* C, int32_t ldc, hipStream_t st);
// C[N, 512] = X[N, K] @ W, K a multiple of 128 (pm_rows_times_weight_longk at Nout = 512)
int pm_wide_rows_times_weight_longk(const float* X, int32_t ldx, int32_t N, int32_t K, const uint16_t* w_frag,
                                    int32_t kind, int32_t w_pitch, float* C, int32_t ldc, hipStream_t st);
