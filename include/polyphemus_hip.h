/*
 * polyphemus_hip.h — C ABI of libpolyphemus_hip.so: the MI355X (gfx950) kernels of
 * the Polyphemus graph-VAE hot path.
 *
 * The reference has no FFI of its own (it is pure Python over torch /
 * torch_geometric / torch_scatter); the drop-in boundary is the
 * `nn.Module` surface of `model.VAE`.  Each entry point below replaces the
 * chain of ATen / torch_scatter kernels that one stretch of the reference's
 * Python launches today; the stretch is cited as `file:line` into the
 * reference tree.  The Python mirror (`polyphemus_amd/model.py`) binds these
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by
 *     the caller (PyTorch); the library never allocates, frees or synchronises;
 *   - work is enqueued on the caller's `stream` (graph-capturable);
 *   - floats are fp32, row-major, contiguous unless a leading dimension is given;
 *     indices are int32 unless stated;
 *   - return value: 0 = PM_OK, <0 = error (PM_E_*); no exceptions cross the ABI.
 */
#ifndef POLYPHEMUS_HIP_H
#define POLYPHEMUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pm_stream_t; /* hipStream_t */

enum {
  PM_OK = 0,
  PM_E_INVALID = -1, /* bad argument (shape, alignment, null pointer) */
  PM_E_LAUNCH = -2,  /* hipGetLastError() != hipSuccess after a launch */
  PM_E_UNSUPPORTED = -3
};

enum { PM_N_REL = 6, PM_N_DIST = 32, PM_N_SLOTS = 15, PM_N_PITCH = 131, PM_N_DUR = 99, PM_N_TOK = 230 };

/* Library / device identification (host only, no GPU needed).  pm_abi_version() returns the PM_ABI_VERSION the library
 * was built with; a host must refuse a library whose version differs from the header it was compiled against (struct
 * layouts — PmBatch, PmGemmDesc, PmVaeLayout —, argument lists and the dropout stream are part of the version).
 *   1: round 1.  2: round 2 (PmBatch.ce_scale, pm_attnpool_bwd +2 arguments, one hash per four channels in the dropout
 *   stream).  3: round 3 (d = 512 in the gcl / linear entry points, pm_gcl_forward_from_planes).
 *   4: round 3, second half (PmGemmDesc.a_colsum — the bias gradient riding in a weight-gradient GEMM —, pm_unembed_ce's
 *   w_planes argument, PM_PLAN_TRK_CNT grown to 32 + 4 W ints by the tile schedule, pm_unembed_dh, pm_bn_small_*,
 *   pm_rows_tn_weight_grad, pm_gcl_tile_order / pm_row_tile_order, pm_vae_step_reload_switches).
 *   5: round 4 (pm_set_deterministic / pm_get_deterministic; plan scratch field grown; PmVaeLayout.flags / .dropout — the
 *   C++ step covers batch_norm = False and cfg.dropout —; pm_vae_step_info writes 16 ints; pm_relu_bwd_planes;
 *   pm_vae_step_backward_encoder_heads, pm_vae_step_join_decoder_grads; pm_head_chain — the latter removed again in ABI 8).
 *   6: round 4, second half (pm_gcl_input_grad_bn / PmBnBwd, pm_bn_bwd_sums: the norm backward inside the input gradient;
 *   pm_chord_pad_vec, pm_chord_tables_fwd, pm_chord_sum_fwd / _bwd, pm_chord_tables_bwd_w / _x: the chord encoder as table algebra).
 *   7: round 5 (the fp16 pair operand format: PmH2, pm_absmax, pm_split_planes_frag_h2, pm_gcl_forward_fused_h2,
 *   pm_gcl_input_grad_bn_h2, pm_gcl_weight_grad_fused_h2; PmNormSums.absmax_out; pm_bn_apply_fused_absmax; pm_vae_step_set_output_grads: the drop-in module's
 *   `model(graph)` + autograd runs the C++ step; pm_vae_step_saved, pm_bn_relu_decisions: parity introspection; pm_batch_flags; pm_unembed_bias_grads; pm_deterministic_faults).
 *   8: round 6 (pm_bar_aggregate_fwd / _bwd: bar-resident aggregation of dense graphs; pm_gcl_forward_from_planes_h2: the dense
 *   route's product in the fp16 pair format; pm_h2_clamp_events: saturation counter of the pair format; PmBatch.flags bit 3 and
 *   pm_vae_step_output_views: the drop-in module's outputs and gradients as views of the arena;
 *   pm_unembed_row_lists / pm_unembed_ce_rows / pm_unembed_dh_rows: the decoder head without its PAD-target rows;
 *   pm_unembed_dw: the three un-embedding weight gradients in one launch). */
#define PM_ABI_VERSION 8
int pm_abi_version(void);
const char* pm_build_info(void);

/* Deterministic mode (environment PM_DETERMINISTIC=1 at first use, or pm_set_deterministic).  The reference's CPU step
 * is reproducible run to run (training.py:137-166 on ATen's CPU kernels); the HIP step by default is not, because sums
 * across workgroups (split-K slices of the head products, column statistics of the norms, weight / bias / table
 * gradients) are float atomics whose order follows the hardware's scheduling.  With the mode on, every launch that
 * contains such atomics orders them by a gate (workgroups take turns in dispatch order, csrc/common.h) and the kernels
 * whose workgroups race on an LDS table run one wave per workgroup: two runs of a step on the same inputs are then
 * BIT-IDENTICAL in outputs, losses and gradients (tests/test_deterministic_gpu.py).  Slower (serialised epilogues):
 * for parity tests and debugging.  Host calls, no GPU work. */
int pm_set_deterministic(int32_t on);
int pm_get_deterministic(void);
/* Faults of the mode since the library was loaded (synchronises the device; < 0: could not be read): gates that could not be set
 * up (the launch ran un-gated) + waves that gave up waiting for their turn after 4 s and went ahead unordered.  0 = every gated
 * launch so far was ordered.  While the mode is on the step keeps every launch on the caller's stream (two gated kernels on two
 * streams could starve each other's turns). */
int pm_deterministic_faults(void);
/* Saturation events of the fp16 pair operand format (PmH2 below): the threads of the kernels that split an fp32 tensor into
 * scaled fp16 hi + lo planes (pm_gcl_input_grad_bn_h2, pm_bn_bwd_fused_h2, pm_split_planes_frag_h2) whose value exceeded
 * +-65504 after scaling and was cut — the |max| BOUND the power-of-two scale is taken from was too small by more than 2^3 (the
 * dh bound holds for standardised values |xhat| <= 14; a near-constant BatchNorm column reaches sqrt(N)).  Counted since the
 * library was loaded or since the last call with reset != 0 (synchronises the device; < 0: could not be read).  0 = no gradient
 * was clipped; the parity tests assert it. */
int pm_h2_clamp_events(int32_t reset);

/* ------------------------------------------------------------------ graph plan
 * Replaces the per-layer boolean-mask edge selection `edge_index[:, edge_type == i]`
 * (model.py:30-38,104-105, 96 times per forward) and the host syncs of
 * `torch.unique(distinct_bars)` (model.py:543) and `batch[-1].item()` (PyG
 * GlobalAttention, model.py:409) by ONE device-side pass per batch:
 * CSR by (dst, relation), CSC by src, per-edge 1/in-degree, bar offsets,
 * drum / non-drum node lists, token histograms.  All integer work, bit-exact.
 *
 * The plan lives in one caller-allocated int32 buffer; `pm_plan_layout` returns the
 * field offsets (in int32 elements) so the host can slice views of it. */
enum {
  PM_PLAN_ROWPTR = 0,   /* [N*6+1] CSR offsets, key = dst*6 + relation            */
  PM_PLAN_CSR_SRC,      /* [E] source node of the edge at this CSR slot           */
  PM_PLAN_CSR_DIST,     /* [E] timestep distance 0..31                            */
  PM_PLAN_CSR_EID,      /* [E] position of the edge in the input edge_index       */
  PM_PLAN_COLPTR,       /* [N+1] CSC offsets, key = src                           */
  PM_PLAN_CSC_DST,      /* [E]                                                    */
  PM_PLAN_CSC_RELDIST,  /* [E] relation | distance << 8                           */
  PM_PLAN_CSC_EID,      /* [E]                                                    */
  PM_PLAN_CSC_INVCNT,   /* [E] float bits: 1 / max(#in-edges of (dst, relation),1) */
  PM_PLAN_NODE_BAR,     /* [N] distinct bar id = bars + n_bars*batch (model.py:403) */
  PM_PLAN_BAR_PTR,      /* [G+1] node offsets of each bar                         */
  PM_PLAN_GROUP_LIST,   /* [2N] drum nodes (ascending) in [0,n_drum); non-drum nodes in [N, N+n_non_drum) */
  PM_PLAN_GROUP_CNT,    /* [4] {n_drum, n_non_drum, 15*n_drum, 15*n_non_drum}     */
  PM_PLAN_TOK_HIST,     /* [4][131] token counts over slots 1..15:
                           0 drum pitch, 1 non-drum pitch, 2 drum dur, 3 non-drum dur */
  PM_PLAN_ROW_LIST,     /* [2*15N] (node, slot) rows = node*15 + slot of the drum nodes in [0, 15 n_drum),
                           of the non-drum nodes from 15N on: row maps of the un-embedding GEMMs */
  PM_PLAN_NODE_TREL,    /* [N] the track relation (0..3) that has in-edges at the node (see pm_segreduce_fwd) */
  PM_PLAN_TRK_LIST,     /* [4N] nodes grouped by that relation (group t at offset t*N), inside a group sorted by
                           class (receives onset edges, receives next edges) in the order (0,0) (1,0) (1,1) (0,1) */
  PM_PLAN_TRK_CNT,      /* [32 + 4 W] {4 group sizes, #nodes with in-edges of more than one track relation, 0,0,0,
                           then 5 class boundaries b0..b4 per group at [8 + 5t + k]: rows [b1,b3) of the list
                           receive onset edges, rows [b2,b4) receive next edges}; from [32] on the tile schedule of
                           the GCL products, (group, first row, rows, 0) for each of the W workgroups of a launch
                           (pm_gcl_tile_order gives the same from a host copy of the first 32 ints) */
  PM_PLAN_SCRATCH,      /* cursors + scan partials                                */
  PM_PLAN_NFIELDS
};
int pm_plan_layout(int32_t N, int32_t E, int32_t G, int64_t* offsets /* [PM_PLAN_NFIELDS+1] */);
int pm_plan_build(const int64_t* edge_index /* [2,E] row0=src,row1=dst (data.py:173) */,
                  const int32_t* edge_type /* [E] 0..5 */, const int32_t* edge_dist /* [E] 0..31 */,
                  const int64_t* bars /* [N] */, const int64_t* batch /* [N] */,
                  const uint8_t* is_drum /* [N] */, const int32_t* tokens /* [N,16,2] */,
                  int32_t n_bars, int32_t n_slots /* active token slots S, 1..15 (see below); 15 = all */,
                  int32_t N, int32_t E, int32_t G, int32_t* plan, pm_stream_t stream);
/* HOST function (no GPU): the schedule of the GCL products over the plan's (track group, 64-row tile) list — the
 * kernels keep one workgroup per CU and a tile costs 2-4 blocks of K, so workgroup b of XCD b % 8 takes the XCD's
 * share of the 4-block tiles first, then of the 3-block, then of the 2-block tiles (csrc/tile_order.h).
 * With 257..264 tiles (one or a few more than the 256 CUs) an XCD that has a tile too many runs one of its cheapest
 * tiles as two 32-row halves behind two others, which keeps the launch as long as its heaviest tile.
 * trk_cnt_host: a host copy of the plan's PM_PLAN_TRK_CNT field [32]; out [3 * cap]: (group, first row of the group's
 * list, rows = 64 | 32) of workgroup b or (-1, -1, 0); returns the number of workgroups a launch for N nodes has. */
int pm_gcl_tile_order(const int32_t* trk_cnt_host, int32_t use_classes, int32_t N, int32_t* out, int32_t cap);
/* ... and of the uniform 64-row tiles of the chord products (pm_rows_times_weight*): out [2 * cap]: (first row, rows =
 * 64 | 32) of workgroup b or (-1, 0); returns the number of workgroups a launch over M rows has. */
int pm_row_tile_order(int32_t M, int32_t* out, int32_t cap);
/* Active slots: slot s >= S holds the PAD token in EVERY node of the batch (S = longest chord + EOS, known
 * to the host that built the batch).  PAD rows carry no loss (ignore_index) and an identical embedding, so the
 * token-level tensors of the fused step are [N, S, .] instead of [N, 15, .]; results are unchanged. */
/* Reference-format inputs -> compact ids (the reference feeds one-hots, data.py:179-182,235-268). */
int pm_edge_attrs_to_ids(const float* edge_attrs /* [E,33] */, int32_t E, int32_t* edge_type,
                         int32_t* edge_dist, pm_stream_t stream);
int pm_tokens_from_onehot(const float* c_tensor /* [N,16,230] */, int32_t N, int32_t* tokens /* [N,16,2] */,
                          pm_stream_t stream);
/* The host-known facts of a batch that does not carry them (a foreign PyG batch: PmBatch.n_slots and flags bit 0):
 * out[0] = last token slot 1..15 holding a non-PAD token in some node (0: none), out[1] != 0: some node receives track
 * edges of more than one track relation (the compact GCL does not apply), out[2] != 0: a token id, edge type or node id is
 * out of range.  `seen` [N] and `out` [3] are caller-zeroed device ints; the caller reads `out` (its one host sync). */
int pm_batch_flags(const int32_t* tokens /* [N,16,2] */, const int64_t* edge_index /* [2,E] */, const int32_t* edge_type,
                   int32_t N, int32_t E, int32_t* seen, int32_t* out, pm_stream_t stream);

/* ------------------------------------------------------------------ device-side graph construction
 * `graph_from_tensor` + the PyG collate of the reference (data.py:24-204) for a whole batch of bars on
 * the device: same node numbering (active cells in (track, timestep) order), same edge order (per bar: track edges per
 * track forward-then-inverse, onset edges per timestep forward-then-inverse, next edges forward only; self loop for an
 * edgeless bar), cell [0,0] of an empty bar switched on in place.  Two calls, the caller sizes the outputs in between
 * (totals[0] = N, totals[1] = E; the only host read):
 *   pm_graph_count: per-bar node / edge counts and their exclusive scans node_ptr / edge_ptr [G+1];
 *   pm_graph_emit : edge_index [2,E] (int64, batch-global node ids), edge_type / edge_dist [E], bars / batch [N]
 *                   (bar index inside the sample, sample index), is_drum [N], node_cell [N] = (g*4 + track)*32 + timestep
 *                   (gathers per-cell payloads such as the token grid). */
int pm_graph_count(float* s_tensor /* [G,4,32] 0/1, empty bars fixed in place */, int32_t G, int32_t* bar_nodes /* [G] */,
                   int32_t* bar_edges /* [G] */, int32_t* node_ptr /* [G+1] */, int32_t* edge_ptr /* [G+1] */,
                   int32_t* totals /* [2] */, pm_stream_t stream);
int pm_graph_emit(float* s_tensor, int32_t G, int32_t n_bars, const int32_t* node_ptr, const int32_t* edge_ptr, int64_t N,
                  int64_t E, int64_t* edge_index, int32_t* edge_type, int32_t* edge_dist, int64_t* bars, int64_t* batch,
                  uint8_t* is_drum, int32_t* node_cell, pm_stream_t stream);

/* Generation helpers, no host synchronisation:
 *   pm_binary_from_logits: `Decoder._binary_from_logits` (model.py:609-623): sigmoid(s_logits) >= thresh, an empty bar
 *                          gets cell [0,0]; written as float 0/1 and / or bytes (either output may be NULL, not both).
 *   pm_mtp_from_logits   : `mtp_from_logits` (utils.py:59-79): mtp [G,4,32,15,230] = the node's logits on active cells
 *                          (nodes numbered in cell order), the hard silence elsewhere (row 0 one-hot pitch EOS = 129,
 *                          rows 1..14 one-hot pitch PAD = 130).  bar_nodes [G] and node_ptr [G+1] are workspaces;
 *                          node_ptr[G] returns the number of active cells, which the caller compares with N (the
 *                          reference raises on a mismatch; the kernel itself writes silence for nodes >= N). */
int pm_binary_from_logits(const float* s_logits /* [G,4,32] */, int32_t G, float thresh, float* s_f32 /* [G,4,32] */,
                          uint8_t* s_u8 /* [G,4,32] */, pm_stream_t stream);
int pm_mtp_from_logits(const float* c_logits /* [N,15,230] */, const float* s_tensor /* [G,4,32] 0/1 */, int32_t G,
                       int64_t N, int32_t* bar_nodes /* [G] */, int32_t* node_ptr /* [G+1] */,
                       float* mtp /* [G,4,32,15,230] */, pm_stream_t stream);

/* ------------------------------------------------------------------ message aggregation
 * `GCL.message` + PyG `propagate` + torch_scatter mean (model.py:110,123-135):
 *   A[n, r*d:(r+1)*d] = mean_{e: dst=n, type=r} keep_e * relu(x[src_e] * T[dist_e]) / (1-p)
 *   A[n, 6d:7d]       = x[n]                       (root operand of model.py:116)
 * T[dist] = edge_nn.weight[:, dist] + edge_nn.bias  (Linear on a one-hot, model.py:127).
 * keep_e is the counter-based dropout mask pm_dropout_keep(seed, layer, eid, channel). */
int pm_edge_table(const float* nn_weight /* [d,32] */, const float* nn_bias /* [d] */, int32_t d,
                  float* T /* [32,d] */, pm_stream_t stream);
int pm_edge_table_bwd(const float* dT /* [32,d] */, int32_t d, float* d_nn_weight /* += */,
                      float* d_nn_bias /* += */, pm_stream_t stream);
/* compact != 0: the aggregate is [N,4d] = [track block | onset | next | x].  A node only receives track edges of
 * ONE track relation (its own track; data.py:36-49,173-176), so three of the four track blocks of every row are
 * identically zero: the compact form stores the non-zero one (relation PM_PLAN_NODE_TREL[n]) and the GCL GEMM
 * contracts K = 4d instead of 7d (the track block against weight[node_trel], grouped by relation).  Valid only
 * when PM_PLAN_TRK_CNT[4] == 0 (true for every graph built by the reference's rules). */
int pm_segreduce_fwd(const float* x /* [N,d] */, const float* T /* [32,d] */, const int32_t* plan,
                     int32_t N, int32_t E, int32_t G, int32_t d, float dropout_p, uint32_t seed,
                     uint32_t layer_uid, int32_t compact, float* A /* [N,7d] or [N,4d] */, pm_stream_t stream);
/* Same aggregate written PRE-SPLIT for the planes mode of the GEMM: three bf16 planes (value = p1 + p2 + p3 exactly),
 * plane k at planes + k*plane_stride, same [N, 7d | 4d] element layout. */
int pm_segreduce_fwd_planes(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                            int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid, int32_t compact,
                            uint16_t* planes, int64_t plane_stride, pm_stream_t stream);
/* GCL.forward (model.py:55-121) of one layer as ONE kernel (gcl.hip): the compact aggregate of 64 rows of a track
 * group is built chunk by chunk in LDS by four producer waves (gather of x rows, GCL.message model.py:123-135,
 * scatter-mean) while four consumer waves contract the previous chunk with the stacked weight [W_t; W_4; W_5; root]
 * (bf16 planes, six products, B fragments straight from `w_frag` = pm_split_planes_frag kind 1 of the layer's [7d, d]
 * weight).  h[n] = A'[n] @ W + bias, bit-identical to pm_segreduce_fwd_planes followed by the grouped planes product;
 * `col_stats` as PmGemmDesc.col_stats; `planes` (optional) receives the A' planes the weight gradient of the backward
 * pass contracts (blocks of row tiles without onset / next receivers are not written when use_classes != 0: the
 * backward never reads them).  d in {128, 256, 512}; the batch must satisfy the compact-GCL premise (track_unique).
 * d = 512 (training.json's width) runs the ring pipeline of wide.hip: eight MFMA waves x 64 of the 512 output columns,
 * the aggregate streamed through the same 2-image LDS ring, the distance table in per-chunk slices. */
int pm_gcl_forward_fused(const float* x /* [N,d] */, const float* T /* [32,d] */, const int32_t* plan, int32_t N,
                         int32_t E, int32_t G, int32_t d, float dropout_p, uint32_t seed, uint32_t layer_uid,
                         const uint16_t* w_frag, const float* bias /* [d] or NULL */, int32_t use_classes,
                         float* h /* [N,d] */, double* col_stats /* [PM_BN_REPL][2][d] += or NULL */,
                         uint16_t* planes /* or NULL */, int64_t plane_stride, pm_stream_t stream);
/* Input gradient of that product, dA'[rows_t] = dh[rows_t] @ [W_t; W_4; W_5; root]^T (the autograd of model.py:104-119),
 * A-stationary (gcl.hip): a workgroup keeps the dh planes of 64 rows of a track group in LDS and walks all 4d output
 * columns; `w_frag_t` = pm_split_planes_frag kind 0 of the layer's [7d, d] weight.  Same result as the grouped planes
 * product with transB (blocks of row tiles without onset / next receivers are not written when use_classes != 0: the
 * segment-reduce backward never reads them).  d in {128, 256, 512}, compact graphs (d = 512: the dh planes stream through
 * the LDS ring of wide.hip once per live 512-column output block instead of staying resident: 192 KB would not fit). */
int pm_gcl_input_grad_fused(const uint16_t* dh_planes /* 3 planes [N,d] */, int64_t plane_stride, const int32_t* plan,
                            int32_t N, int32_t E, int32_t G, int32_t d, const uint16_t* w_frag_t, int32_t use_classes,
                            float* dA /* [N,4d] */, pm_stream_t stream);
/* The same with the BatchNorm backward in front of it (pm_bn_bwd_fused with sums_ready != 0: autograd of model.py:203-206)
 * run in the kernel's prologue instead of as a pass of its own: a workgroup forms dh for its 64 rows from the pre-norm rows
 * `h`, the norm's output gradient `du` and the column sums in `acc3` [PM_BN_REPL][3][d] (pm_segreduce_bwd_norm or
 * pm_bn_bwd_sums), multiplies it as above and WRITES the three dh planes (for pm_gcl_weight_grad_fused afterwards);
 * dgamma / dbeta / dbias_pre (+=, any may be NULL) as pm_bn_bwd_fused.  Same values as the two calls.  d in {128, 256}. */
typedef struct PmBnBwd {
  const float* h;        /* [N,d] input of the norm */
  const float* du;       /* [N,d] gradient of its output */
  const float* mean; const float* var; const float* gamma; const float* beta; /* [d] batch statistics, parameters */
  const double* acc3;    /* [PM_BN_REPL][3][d] column sums of du', du' * xhat, xhat */
  float* dgamma; float* dbeta; float* dbias_pre; /* [d] += or NULL */
  float eps; int32_t relu; /* relu != 0: the norm is followed by a ReLU (du' = du * [BN(h) > 0]) */
  int32_t add_residual;  /* != 0: the self block of dA' leaves as dA'[n, 3d:4d] + du[n] — the residual path of
                            x_i = x_{i-1} + relu(BN(h)), so that pm_segreduce_bwd(_norm) runs with dres = NULL */
  int32_t reserved;
} PmBnBwd;
int pm_gcl_input_grad_bn(const PmBnBwd* norm, uint16_t* dh_planes /* 3 planes [N,d], written */, int64_t plane_stride,
                         const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, const uint16_t* w_frag_t,
                         int32_t use_classes, float* dA /* [N,4d] */, pm_stream_t stream);
/* ---- the fp16 pair operand format ("h2") of the three GCL products at d in {128, 256}
 * An fp32 product on the matrix cores needs its operands split into 16-bit pieces.  The exact three-term bf16 split above
 * costs six MFMA products per fp32 product.  The h2 kernels use Ootomo & Yokota's error-corrected half-precision product
 * instead: v * s = hi + lo with hi, lo fp16 (22 significant bits; s a power of two chosen from the tensor's |max| so that
 * the five-bit exponent of fp16 is not a limit: exact to apply and to undo), and the product runs as the THREE MFMA
 * products hi*hi + hi*lo + lo*hi with fp32 accumulation: element error <= 2^-22 relative plus the dropped lo*lo term of the
 * same size — below the rounding an fp32 dot product of the same length accumulates — at half the matrix-core work, two
 * thirds of the operand bytes (csrc/common.h pm_split2h_pair; parity of the step against the fp64 oracle: tests/).
 * Planes keep the layout and strides of the three-plane format; plane 2 is unused.  d = 512 runs the same format through the
 * ring pipeline of wide.hip (pm_gcl_forward_fused_h2, pm_bn_bwd_fused_h2 + pm_gcl_input_grad_fused_h2, pm_gcl_weight_grad_fused_h2).
 *   absmax_in : device words [PM_ABSMAX_SLOTS] holding float bits whose maximum is max |x| of the kernel's fp32 input (pm_absmax, or a producer's
 *               absmax_out: pm_bn_apply_fused_absmax, PmNormSums.absmax_out); forward: of the layer input x, input gradient: of du
 *   absmax_aux: forward only: the same for the distance table T
 *   scale_out : device float the kernel WRITES: the power of two its activation planes (A' / dh) carry; the weight gradient
 *               undoes both
 *   w_scale   : the power of two the weight planes were built with (pm_split_planes_frag_h2) */
enum { PM_ABSMAX_SLOTS = 64 };   /* a tensor's |max| lives in this many words (the maximum of them counts; one wave reads them with
                                   one load): workgroups add theirs with ONE atomic each, spread over the slots — atomics on one
                                   address serialise at ~150 ns each (4096 of them cost a 10 us launch 37 us more) */
typedef struct PmH2 {
  const uint32_t* absmax_in; const uint32_t* absmax_aux; float* scale_out; float w_scale; int32_t reserved;
} PmH2;
/* out[PM_ABSMAX_SLOTS]: slot = max(slot, float bits of the |max| a workgroup saw) (atomic; the words must start at 0 or at
 * an earlier maximum) */
int pm_absmax(const float* x, int64_t n, uint32_t* out, pm_stream_t stream);
int pm_split_planes_frag_h2(const float* W, int32_t rows, int32_t cols, int32_t kind, int32_t n_mats, int64_t src_stride,
                            int64_t dst_stride, float w_scale, uint16_t* out, pm_stream_t stream);
int pm_gcl_forward_fused_h2(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                            float dropout_p, uint32_t seed, uint32_t layer_uid, const uint16_t* w_frag, const float* bias,
                            int32_t use_classes, float* h, double* col_stats, uint16_t* planes, int64_t plane_stride,
                            const PmH2* h2, pm_stream_t stream);
int pm_gcl_input_grad_bn_h2(const PmBnBwd* norm, uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan, int32_t N,
                            int32_t E, int32_t G, int32_t d, const uint16_t* w_frag_t, int32_t use_classes, float* dA,
                            const PmH2* h2, pm_stream_t stream);
/* d = 512 (no norm backward inside the input gradient there): the norm's backward pass writing dh planes in the pair format,
 * and the input gradient reading them */
int pm_bn_bwd_fused_h2(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var, float eps,
                       const float* gamma, const float* beta, int relu, float* dgamma, float* dbeta, float* dbias_pre,
                       double* acc3, uint16_t* dx_planes, int64_t plane_stride, int32_t sums_ready, const PmH2* h2,
                       pm_stream_t stream);
int pm_gcl_input_grad_fused_h2(const uint16_t* dh_planes, int64_t plane_stride, const int32_t* plan, int32_t N, int32_t E,
                               int32_t G, int32_t d, const uint16_t* w_frag_t, int32_t use_classes, float* dA,
                               const float* dh_scale, float w_scale, pm_stream_t stream);
int pm_gcl_weight_grad_fused_h2(const uint16_t* a_planes, int64_t a_plane_stride, const uint16_t* dh_planes,
                                int64_t dh_plane_stride, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                                int32_t use_classes, float* dW, const float* a_scale, const float* dh_scale,
                                pm_stream_t stream);
/* Weight gradient of that product, d[W_t; W_4; W_5; root] += A'[rows_t]^T dh[rows_t] (gcl.hip): 128x128 output tiles,
 * one workgroup per (tile, track group, K slice), operands streamed by loader waves through an LDS ring, K slices added
 * with float atomics; `dW` is the layer's [7d, d] gradient (+=).  Same result as the grouped planes product with transA
 * up to the order of the atomic adds.  d in {128, 256, 512}, compact graphs. */
int pm_gcl_weight_grad_fused(const uint16_t* a_planes /* 3 planes [N,4d] */, int64_t a_plane_stride,
                             const uint16_t* dh_planes /* 3 planes [N,d] */, int64_t dh_plane_stride, const int32_t* plan,
                             int32_t N, int32_t E, int32_t G, int32_t d, int32_t use_classes, float* dW /* [7d,d] += */,
                             pm_stream_t stream);
/* The weight products of GCL.forward (model.py:112-119) with the aggregate READ from A' planes (written by
 * pm_segreduce_fwd_planes) instead of built in the kernel: h[n] = A'[n] @ [W_t; W_4; W_5; root] + bias, `col_stats` as
 * above.  The path of dense graphs (BASELINE configs[4]: hundreds of edges per node — the fused kernel's producers keep
 * at most three edges per (node, relation) in flight).  d = 512 (wide.hip). */
int pm_gcl_forward_from_planes(const uint16_t* a_planes /* 3 planes [N,4d] */, int64_t plane_stride, const int32_t* plan,
                               int32_t N, int32_t E, int32_t G, int32_t d, const uint16_t* w_frag,
                               const float* bias /* [d] or NULL */, int32_t use_classes, float* h /* [N,d] */,
                               double* col_stats /* [PM_BN_REPL][2][d] += or NULL */, pm_stream_t stream);
/* ... on planes in the fp16 pair format: two fp16 planes of A' * (*a_scale), written by pm_bar_aggregate_fwd with a PmH2 (whose
 * scale_out is `a_scale`); `w_frag` from pm_split_planes_frag_h2(kind 1) with `w_scale`.  d = 512. */
int pm_gcl_forward_from_planes_h2(const uint16_t* a_planes, int64_t plane_stride, const int32_t* plan, int32_t N, int32_t E,
                                  int32_t G, int32_t d, const uint16_t* w_frag, const float* bias, int32_t use_classes,
                                  float* h, double* col_stats, const float* a_scale, float w_scale, pm_stream_t stream);
/* C[N, Nout] = X[N, K] @ W (+ bias) for a plain linear layer with a short inner dimension, K in {128, 256, 512}, Nout a
 * multiple of K (chord decoder forward model.py:555-559; chord encoder input gradient, autograd of model.py:384-390),
 * A-stationary (linear.hip k_rows_w): the 64 fp32 rows of a tile are split into bf16 planes once and kept in LDS for all
 * output columns.  `w_frag` = pm_split_planes_frag of the weight: kind 0 for W [Nout, K] (y = x W^T), kind 1 for
 * W [K, 32*w_tiles] (y = x W; only the first Nout columns are used).  K = 512: wide.hip (X re-split per 512-column
 * output block, streamed through the LDS ring). */
int pm_rows_times_weight(const float* X, int32_t ldx, int32_t N, int32_t K, const uint16_t* w_frag, int32_t kind,
                         int32_t w_tiles, int32_t Nout, const float* bias /* [Nout] or NULL */, float* C, int32_t ldc,
                         pm_stream_t stream);
/* The same for a long inner dimension and d output columns: C[N, Nout] = X[N, K] @ W, K a multiple of 128, Nout in
 * {128, 256, 512} (chord encoder forward model.py:384-390 without its bias; chord decoder input gradient): producer waves split
 * 64 x 128 fp32 chunks into bf16 planes in an LDS ring, MFMA waves contract them (linear.hip k_rows_wk).  `w_frag`: kind 0
 * for W [Nout, 16*w_pitch] (y = x W[:, :K]^T), kind 1 for W [K, Nout] (y = x W). */
int pm_rows_times_weight_longk(const float* X, int32_t ldx, int32_t N, int32_t K, const uint16_t* w_frag, int32_t kind,
                               int32_t w_pitch, int32_t Nout, float* C, int32_t ldc, pm_stream_t stream);
/* ... and the layer's weight gradient over the same rows: C[M, Nn] += A[:, :M]^T B[:, :Nn] (K rows each; fp32, leading
 * dimensions multiples of 4, M and Nn multiples of 128), colsum_a (optional) [M] += column sums of A — the bias gradient
 * when A is the layer's output gradient.  Six-product bf16 chain as the products above; K slices add with float atomics. */
int pm_rows_tn_weight_grad(const float* A, int32_t lda, int32_t M, const float* B, int32_t ldb, int32_t Nn, int32_t K,
                           float* C /* += */, int32_t ldc, float* colsum_a /* NULL or += */, pm_stream_t stream);
/* pm_segreduce_bwd_norm: as pm_segreduce_bwd, and additionally accumulates the three column sums that the backward of
 * the BatchNorm BELOW needs (dx is that norm's output gradient: x_i = x_{i-1} + relu(BN(h_{i-1})), model.py:203-206)
 * into acc3 [PM_BN_REPL][3][d] (caller-zeroed), so that pm_bn_bwd_fused can run with sums_ready = 1. */
typedef struct PmNormSums {
  const float* h;                  /* [N,d] input of that norm (pre-norm GCL output of the layer below) */
  const float* mean; const float* var; const float* gamma; const float* beta;   /* [d] */
  float eps; int32_t relu;
  double* acc3;
  uint32_t* absmax_out;            /* NULL, or device words [PM_ABSMAX_SLOTS]: atomic max of the float bits of |dx| (PmH2.absmax_in of the layer below) */
} PmNormSums;
int pm_segreduce_bwd_norm(const float* x, const float* T, const float* dA, const float* dres, const int32_t* plan,
                          int32_t N, int32_t E, int32_t G, int32_t d, float dropout_p, uint32_t seed,
                          uint32_t layer_uid, int32_t compact, float* dx, float* dT, const PmNormSums* next_norm,
                          pm_stream_t stream);
int pm_segreduce_bwd(const float* x, const float* T, const float* dA /* [N,7d] or [N,4d] */,
                     const float* dres /* [N,d] or NULL: added to dx (residual path, model.py:206) */,
                     const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, float dropout_p,
                     uint32_t seed, uint32_t layer_uid, int32_t compact, float* dx /* [N,d] */,
                     float* dT /* [32,d] += */, pm_stream_t stream);

/* Bar-resident aggregation (bar.hip): the route of DENSE graphs (BASELINE configs[4]: 127 in-edges per node).  Same maths
 * as pm_segreduce_fwd_planes / pm_segreduce_bwd(_norm) on the compact layout (GCL.message model.py:123-135, propagate /
 * scatter-mean model.py:110, their autograd), but one workgroup owns one bar (edges never leave their bar, data.py:24-121; a bar
 * has at most 4 x 32 nodes — a larger "bar" traps) and a chunk of channels, reads the bar's rows from HBM ONCE into LDS and
 * gathers from there.  Forward: A' [N, 4d] as operand planes — three bf16 planes (`h2` NULL; bit-identical to
 * pm_segreduce_fwd_planes(compact = 1)) or the two fp16 planes of the pair format (`h2`: absmax_in = |max| words of x, absmax_aux
 * = of T, *scale_out receives the power of two the planes carry; w_scale unused); d a multiple of 128.  Backward: dx, dT += and
 * (next_norm non-NULL) the column sums / |dx|max of the norm below, as pm_segreduce_bwd_norm; d a multiple of 64. */
int pm_bar_aggregate_fwd(const float* x, const float* T, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                         float dropout_p, uint32_t seed, uint32_t layer_uid, uint16_t* planes, int64_t plane_stride,
                         const PmH2* h2 /* or NULL */, pm_stream_t stream);
int pm_bar_aggregate_bwd(const float* x, const float* T, const float* dA /* [N,4d] */, const float* dres /* [N,d] or NULL */,
                         const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, float dropout_p, uint32_t seed,
                         uint32_t layer_uid, float* dx /* [N,d] */, float* dT /* [32,d] += */,
                         const PmNormSums* next_norm /* or NULL */, pm_stream_t stream);

/* ------------------------------------------------------------------ dense contraction (fp32 MFMA)
 * C[M,N] (=|+=) op(A)[M,K] * op(B)[K,N] (+ bias[N]) (ReLU), v_mfma_f32_32x32x2_f32.
 * Replaces `addmm`/`mm` of every nn.Linear and of `h @ weight[i]`, `x_r @ root`
 * (model.py:112,116) — with A = [h_0|..|h_5|x] the 7 GEMMs of a GCL are one call.
 * transA: A is stored [K,M]; transB: B is stored [N,K] (nn.Linear weight).
 * rowmap (optional): the gathered dimension (M when !transA, else K) is indirect:
 *   physical row = rowmap[r / rows_per_entry] * rows_per_entry + r % rows_per_entry
 *   and applies to A and C (!transA) or to A and B (transA); `dyn_entries` (device int,
 *   optional) overrides the entry count so no host sync is needed for data-dependent sizes. */
enum { PM_GEMM_RELU = 1, PM_GEMM_ACCUM = 2,
       PM_GEMM_RELU_ADD = 16 /* C = C + relu(op(A) op(B) + bias): the residual tail x + relu(.) of an eval-mode GCL layer
                                whose BatchNorm is folded into the weights (pm_bn_fold_weights); no split-K */,
       PM_GEMM_ZEROED = 8 /* C is known to hold zeros (a pre-cleared arena region): the split-K path of the small head
                             products skips its own clear */,
       PM_GEMM_PARTITION = 4 /* grouped + row lists + device counts: the groups' lists partition at most M (K when
                                transA) rows IN TOTAL; the launch then only enumerates live row panels */ };
/* tile configuration pm_gemm_f32 picks for a shape (host only).  fp32 MFMA: 0 = 64x64x16, 1 = 128x128x16,
 * 2 = 64x64x32, 3 = 128x128x32.  Split mode (fp32 operands split exactly into three bf16 terms, six partial products
 * on v_mfma_f32_32x32x16_bf16, fp32 accumulation; error below an fp32 FMA chain): 4 = 128x128x16, 5 = 128x64x16,
 * 6 = 64x64x32, 7 = 128x128x32; needs 16-byte aligned operands, otherwise the fp32 rule applies. */
int pm_gemm_config(int32_t transA, int32_t M, int32_t N, int32_t K);
int pm_gemm_force_config(int32_t cfg); /* -1 = automatic (default); 0..7 pins a configuration (A/B timing) */
int pm_gemm_f32(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda,
                const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias, int flags,
                int split_k, const int32_t* rowmap, int32_t rows_per_entry, const int32_t* dyn_entries,
                pm_stream_t stream);
/* n_groups independent GEMMs of the same shape bound in ONE launch (blockIdx.y = group): group g uses
 * A + g*a_stride, B + g*b_stride, C + g*c_stride, bias + g*bias_stride (elements), rowmap + g*map_stride and
 * dyn_entries + g*dyn_stride.  With a row map and a device-side count per group this is the per-relation
 * contraction of the compact GCL: h[rows_t] += A'[rows_t, :d] @ weight[t] for the four track relations t. */
int pm_gemm_f32_grouped(int transA, int transB, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda,
                        const float* B, int32_t ldb, float* C, int32_t ldc, const float* bias, int flags,
                        int split_k, const int32_t* rowmap, int32_t rows_per_entry, const int32_t* dyn_entries,
                        int32_t n_groups, int64_t a_group_stride, int64_t b_group_stride, int64_t c_group_stride,
                        int64_t bias_group_stride, int32_t map_group_stride, int32_t dyn_group_stride,
                        pm_stream_t stream);
/* Descriptor form of the grouped GEMM, plus "stacked" operands: when b_split_rows > 0 the STORED rows of B below
 * b_split_rows belong to the group (B + g*b_group_stride) and the rows at and above it are shared by all groups and
 * read from B + b_shared_off + row*ldb.  c_split_rows / c_shared_off do the same for the rows of C when transA (the
 * shared rows are then accumulated atomically by all groups).  This binds the whole compact GCL contraction
 *   h[rows_t] = A'[rows_t, 0:4d] @ [weight[t]; weight[4]; weight[5]; root]        (model.py:112,116)
 * its input gradient and its weight gradient in one launch each. */
/* fp32 -> three bf16 planes with x = p1 + p2 + p3 EXACTLY (8 significand bits each); n % 4 == 0. */
int pm_split_planes(const float* src, int64_t n, uint16_t* planes, int64_t plane_stride /* elements, >= n */,
                    pm_stream_t stream);
typedef struct PmGemmDesc {
  int32_t transA, transB, M, N, K;
  const float* A; int32_t lda;
  const float* B; int32_t ldb;
  float* C; int32_t ldc;
  const float* bias;
  int32_t flags, split_k;
  const int32_t* rowmap; int32_t rows_per_entry;
  const int32_t* dyn_entries;
  int32_t n_groups;
  int64_t a_group_stride, b_group_stride, c_group_stride, bias_group_stride;
  int32_t map_group_stride, dyn_group_stride;
  int32_t b_split_rows; int64_t b_shared_off;
  int32_t c_split_rows; int64_t c_shared_off;
  double* col_stats; /* optional [PM_BN_REPL][2][N] fp64, += : column sums of the stored C values and of their
                        squares (the statistics pass of the BatchNorm that follows, fused into the epilogue; the
                        replica is picked from the row-panel index); !transA, no ACCUM, split_k == 1 */
  int32_t operand_planes; /* != 0: A and B are given PRE-SPLIT as three bf16 planes each (pm_split_planes and the
                        plane outputs of pm_segreduce_fwd / pm_bn_bwd_fused): A / B point at plane 0, the planes are
                        a_plane_stride / b_plane_stride ELEMENTS apart, lda / ldb and the group strides count bf16
                        elements (multiples of 8); the six-product split GEMM then runs without conversion work */
  int64_t a_plane_stride, b_plane_stride;
  const int32_t* class_ptr; /* optional, with row lists: per group 5 boundaries b0..b4 of the list (PM_PLAN_TRK_CNT + 8):
                        rows [b1,b3) have a non-zero second block, rows [b2,b4) a non-zero third block of the
                        [4 x class_block] blocked dimension (K when !transA && !transB, N when transB, M when transA);
                        the all-zero blocks are skipped tile by tile (C tiles that only hold such blocks are left
                        untouched when transB).  Results are unchanged where they are defined. */
  int32_t class_block;
  const void* b_frag;       /* optional, planes mode, !transA: B (a weight matrix, possibly stacked via b_split_rows)
                               additionally as FRAGMENT-MAJOR planes (pm_split_planes_frag: kind 0 when transB, kind 1
                               otherwise, over the whole stored matrix with `ldb` columns).  The kernel then takes its B
                               operands straight from memory into registers (no LDS image of B); used when the shape
                               is aligned with the fragment grid (N % 128 == 0, K % 32 == 0), ignored otherwise. */
  float* a_colsum;          /* optional, transA (weight gradient dW = dy^T x of a linear layer, A = dy stored [K, M]):
                               a_colsum[m] += sum_k A[k, m] — the layer's bias gradient, computed by the same launch
                               (no row map, one group, fp32 operands) */
} PmGemmDesc;
/* Fragment-major bf16 planes of weight matrices W [rows, cols] (fp32; rows, cols multiples of 32) for PmGemmDesc.b_frag:
 * per (32-wide n tile, 16-wide k-step) three contiguous 1 KiB blocks (planes) in MFMA operand order.  kind 0: n = row,
 * k = column (the product uses W transposed: transB); kind 1: k = row, n = column.  out: rows*cols*3 bf16 per matrix. */
int pm_split_planes_frag(const float* W, int32_t rows, int32_t cols, int32_t kind, int32_t n_mats, int64_t src_stride,
                         int64_t dst_stride, uint16_t* out, pm_stream_t stream);
int pm_gemm_f32_desc(const PmGemmDesc* desc, pm_stream_t stream);

/* ------------------------------------------------------------------ batch normalisation
 * nn.BatchNorm1d / BatchNorm2d / PyG BatchNorm (model.py:203,222,228,282,359-375,338,475,638).
 * Tensors are viewed as [O, C, I] (row-major [M,C]: O=M, I=1; NCHW: O=G, I=H*W). */
int pm_bn_stats(const float* x, int32_t O, int32_t C, int32_t I, float* mean /* [C] */,
                float* var /* [C] biased */, float* running_mean /* NULL or [C], updated */,
                float* running_var, float momentum, double* scratch /* [PM_BN_SCRATCH(C)] */, pm_stream_t stream);
int pm_bn_apply(const float* x, int32_t O, int32_t C, int32_t I, const float* mean, const float* var,
                float eps, const float* gamma, const float* beta, const float* residual /* or NULL */,
                int relu, float* y, pm_stream_t stream);
/* dx = BN'(du), du = dy * [relu ? bn(x) > 0 : 1]; dgamma/dbeta accumulate; dbias_pre (optional) += column sums of dx,
 * i.e. the gradient of a bias added right in front of this BatchNorm (GCL.bias, model.py:119 + :203). */
int pm_bn_bwd(const float* x, const float* dy, int32_t O, int32_t C, int32_t I, const float* mean,
              const float* var, float eps, const float* gamma, const float* beta, int relu,
              float* dgamma /* += */, float* dbeta /* += */, float* dbias_pre /* NULL or += */, float* dx,
              double* scratch /* [PM_BN_SCRATCH(C)] */, pm_stream_t stream);
#define PM_BN_SCRATCH(C) (256 * 3 * (C) + 4 * (C))
/* The same two operations in ONE launch each for small batches (row-major [O, C], O <= PM_BN_SMALL_MAX_ROWS: the norms of
 * the heads over B rows, model.py:475,638): statistics (+ running statistics) and apply; backward sums and dx.  No scratch. */
#define PM_BN_SMALL_MAX_ROWS 2048
int pm_bn_small_fwd(const float* x, int32_t O, int32_t C, float eps, const float* gamma, const float* beta,
                    const float* residual /* or NULL */, int relu, float* y, float* mean /* [C] out */,
                    float* var /* [C] biased, out */, float* running_mean /* NULL or [C], updated */, float* running_var,
                    float momentum, pm_stream_t stream);
int pm_bn_small_bwd(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var, float eps,
                    const float* gamma, const float* beta, int relu, float* dgamma /* += */, float* dbeta /* += */,
                    float* dbias_pre /* NULL or += */, float* dx, pm_stream_t stream);
/* Split forms for synchronised BatchNorm under data parallelism (the reference is single-device, so its
 * BatchNorm statistics span what is here the GLOBAL batch): the column sums of one rank come out as `sums` [3][C] fp64 —
 * statistics mode (dy == NULL): {sum x, sum x^2, 0}; backward mode: {sum du, sum du*xhat, sum xhat} —, the host adds them
 * over the ranks (one all-reduce) and hands the totals back with the global row count O_global * I. */
int pm_bn_partial_sums(const float* x, const float* dy /* or NULL */, int32_t O, int32_t C, int32_t I, const float* mean,
                       const float* var, float eps, const float* gamma, const float* beta, int relu,
                       double* sums /* [3][C] out */, double* scratch /* [PM_BN_SCRATCH(C)] */, pm_stream_t stream);
int pm_bn_stats_from_sums(const double* sums /* [3][C] */, double count, int32_t C, float* mean, float* var,
                          float* running_mean /* or NULL */, float* running_var, float momentum, pm_stream_t stream);
int pm_bn_bwd_from_sums(const float* x, const float* dy, int32_t O, int32_t C, int32_t I, const float* mean,
                        const float* var, float eps, const float* gamma, const float* beta, int relu,
                        const double* sums_local /* [3][C] this rank */, const double* sums_global /* [3][C] all ranks */,
                        double count_global, float* dgamma /* += */, float* dbeta /* += */, float* dbias_pre /* NULL or += */,
                        float* dx, double* scratch /* [2C] */, pm_stream_t stream);

/* ------------------------------------------------------------------ element-wise helpers */
/* Fused forms for row-major [O, C] (I = 1, C % 4 == 0, 16-byte aligned), used by the native step for the GCL norms:
 * `sums` [PM_BN_REPL][2][C] fp64 = column sums of x and x*x, accumulated by the epilogue of the GEMM that produced x
 * (PmGemmDesc.col_stats; PM_BN_REPL replicas spread the atomics, consumers add them up); mean / var / running
 * statistics are written as a side effect (saved for the backward).  pm_bn_bwd_fused accumulates its three column
 * sums into the caller-ZEROED `acc3` [PM_BN_REPL][3][C] fp64 with atomics and needs no finalize launch. */
enum { PM_BN_REPL = 8 };
int pm_bn_apply_fused(const float* x, int32_t O, int32_t C, const double* sums, float eps, const float* gamma,
                      const float* beta, const float* residual /* or NULL */, int relu, float* y,
                      float* mean /* [C] out */, float* var /* [C] out */, float* running_mean /* or NULL */,
                      float* running_var, float momentum, pm_stream_t stream);
/* ... which also leaves max |y| (float bits, atomic max) in absmax_out[PM_ABSMAX_SLOTS]: PmH2.absmax_in of the GCL layer that reads y */
int pm_bn_apply_fused_absmax(const float* x, int32_t O, int32_t C, const double* sums, float eps, const float* gamma,
                             const float* beta, const float* residual, int relu, float* y, float* mean, float* var,
                             float* running_mean, float* running_var, float momentum, uint32_t* absmax_out,
                             pm_stream_t stream);
int pm_bn_bwd_fused(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var,
                    float eps, const float* gamma, const float* beta, int relu, float* dgamma, float* dbeta,
                    float* dbias_pre /* or NULL */, float* dx /* fp32 output, or NULL with dx_planes */, double* acc3,
                    uint16_t* dx_planes /* or NULL: dx as three bf16 planes (PmGemmDesc.operand_planes) */,
                    int64_t plane_stride /* elements */,
                    int32_t sums_ready /* != 0: acc3 already holds the sums (pm_segreduce_bwd_norm) */,
                    pm_stream_t stream);
/* The column sums of pm_bn_bwd_fused alone: acc3 [PM_BN_REPL][3][C] (caller-zeroed) += sums of du', du' * xhat, xhat. */
int pm_bn_bwd_sums(const float* x, const float* dy, int32_t O, int32_t C, const float* mean, const float* var, float eps,
                   const float* gamma, const float* beta, int relu, double* acc3, pm_stream_t stream);
/* Eval-mode BatchNorm folded into the linear map in front of it (model.py:203 under `vae.eval()`,
 * generate.py:112): with s = gamma / sqrt(running_var + eps), t = beta - running_mean * s,
 *   W_out[k, n] = W[k, n] * s[n]      (W [rows, cols] row-major, the GCL operand [weight; root] [7d, d])
 *   b_out[n]    = bias[n] * s[n] + t[n]
 * so BN(A @ W + b) = A @ W_out + b_out and the normalisation pass over [N, d] disappears from the forward. */
int pm_bn_fold_weights(const float* W, int32_t rows, int32_t cols, const float* bias, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, float eps, float* W_out, float* b_out,
                       pm_stream_t stream);
/* `num_batches_tracked` of all BatchNorm modules after a training forward: counters[i] += inc[i] + [group_cnt[0] > 0] *
 * sel[0][i] + [group_cnt[1] > 0] * sel[1][i] (the embedding norms only count when their node group — drums / non-drums,
 * plan field GROUP_CNT — is non-empty, model.py:362,375). */
int pm_bn_counters_update(int64_t* counters /* [n] */, const int64_t* inc /* [n] */, const int64_t* sel /* [2,n] */,
                          const int32_t* group_cnt /* [2] */, int32_t n, pm_stream_t stream);
int pm_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, pm_stream_t stream);
/* dh = dy * [h > 0], as fp32 (dh, may be NULL) and / or as three bf16 operand planes (planes, may be NULL; plane_stride
 * elements apart): the backward of the GCL layer tail x' = x + relu(h) of a model built with batch_norm = False
 * (model.py:202-206), where pm_bn_bwd_fused stands otherwise.  n % 4 == 0. */
int pm_relu_bwd_planes(const float* dy, const float* h, int64_t n, float* dh, uint16_t* planes, int64_t plane_stride,
                       pm_stream_t stream);
/* y = relu(x) + res (res may be NULL): layer tail of a model built with batch_norm = False (model.py:203-206,219-230). */
int pm_relu_residual_fwd(const float* x, const float* res, int64_t n, float* y, pm_stream_t stream);
/* Element dropout of the cfg.dropout layers (model.py:160,199,244-247,267-270,389-390,473,479,558-559,640) on a
 * row-major [rows, cols] tensor: y = x * keep(seed, site, row, col) / (1 - p) with the counter hash of the message
 * dropout (pm_dropout_hash(seed, site, row, col)); y may alias x; the backward is the same call on the gradient. */
int pm_dropout_rows(const float* x, int64_t rows, int32_t cols, float p, uint32_t seed, uint32_t site, float* y,
                    pm_stream_t stream);
int pm_add(const float* a, const float* b, int64_t n, float* out, pm_stream_t stream);
int pm_colsum_acc(const float* x, int32_t M, int32_t C, int32_t ld, float* out /* [C] += */, pm_stream_t stream);
/* same over an indirect row set (rows = rowmap[r / rpe] * rpe + r % rpe, entry count read on device) */
int pm_colsum_rows_acc(const float* x, int32_t C, int32_t ld, const int32_t* rowmap, int32_t rows_per_entry,
                       const int32_t* dyn_entries, int32_t max_entries, float* out /* [C] += */, pm_stream_t stream);
int pm_reparam_fwd(const float* mu, const float* log_var, const float* eps, int64_t n, float* z, pm_stream_t stream);
int pm_reparam_bwd(const float* dz, const float* log_var, const float* eps, int64_t n, float* dmu /* += */,
                   float* dlog_var /* += */, pm_stream_t stream);

/* ------------------------------------------------------------------ token embedding front
 * ContentEncoder embeddings (model.py:355-377): Linear(131|99 -> d/2) on one-hot tokens is
 * a row lookup, and BatchNorm1d over those rows is a count-weighted statistic of the
 * table, so the stage is: normalise 4 small tables, then gather. */
int pm_embed_tables(const float* w_pitch_drum /* [d/2,131] */, const float* b_pitch_drum,
                    const float* w_pitch_nd, const float* b_pitch_nd, const float* w_dur /* [d/2,99] */,
                    const float* b_dur, const float* bn_drum_g, const float* bn_drum_b,
                    const float* bn_nd_g, const float* bn_nd_b, const float* bn_dur_g, const float* bn_dur_b,
                    float* rm_drum, float* rv_drum, float* rm_nd, float* rv_nd, float* rm_dur, float* rv_dur,
                    const int32_t* tok_hist /* [4][131] */, int32_t d, int training, float eps, float momentum,
                    float* tables /* [4][131][d/2] normalised */, float* stats /* [4][2][d/2] mean,var */,
                    pm_stream_t stream);
int pm_embed_gather(const float* tables, const int32_t* tokens /* [N,16,2] */, const uint8_t* is_drum,
                    int32_t N, int32_t d, int32_t n_slots, float* X /* [N,S,d] */, pm_stream_t stream);
int pm_embed_bwd_scatter(const float* dX /* [N,S,d] */, const int32_t* tokens, const int32_t* plan, int32_t N,
                         int32_t E, int32_t G, int32_t d, int32_t n_slots,
                         float* S /* [4][131][d/2], zeroed by the call */, pm_stream_t stream);
/* PAD tail of the chord encoder when n_slots < 15 (model.py:381-390): forward adds the constant contribution of the
 * all-PAD slots (+ the Linear bias) per node group and applies the ReLU in place; backward adds the tail's exact
 * gradients to chord_encoder.weight[:, S*d:] and to the PAD rows of the token sums. */
int pm_chord_pad_fwd(const float* tables, const float* chord_w /* [d,15d] */, const float* chord_b,
                     const uint8_t* is_drum, int32_t N, int32_t d, int32_t n_slots, float* cvec /* [2][d] scratch */,
                     float* y /* [N,d] in place */, pm_stream_t stream);
int pm_chord_pad_bwd(const float* dy /* [N,d] */, const uint8_t* is_drum, int32_t N, int32_t d, int32_t n_slots,
                     const float* tables, const float* chord_w, float* gsum /* [2][d] scratch */,
                     float* d_chord_w /* += */, float* S /* token sums, += */, pm_stream_t stream);
/* The chord encoder as table algebra (chord.hip; model.py:344-390): its input is a lookup of the four embedding tables, so the
 * Linear(15 d -> d) distributes over it.  Forward: PT [2 groups][S][2 kinds][131][d] = table rows times the slot's weight
 * block (pm_chord_tables_fwd), x0[n] = relu(cvec[group] + the 2 S looked-up rows of PT) (pm_chord_sum_fwd; cvec [2][d] =
 * bias + the all-PAD tail slots, pm_chord_pad_vec).  Backward, dy = gradient of the pre-activation (ReLU mask applied):
 * Gt (layout of PT; the CALLER clears it) += per (group, slot, kind, token) sums of dy rows (pm_chord_sum_bwd: one-hot^T x dy on
 * the matrix cores), from which pm_chord_tables_bwd_w forms d_chord_w[:, :S*d] (+=) and d_chord_b (+= column sums of dy; may be
 * NULL) and pm_chord_tables_bwd_x the token sums S [4][131][d/2] (+= with float atomics: the caller clears S) that
 * pm_embed_tables_bwd takes; the tail slots follow through pm_chord_pad_bwd as before (any order: it only adds).  X [N, S, d]
 * and its gradient are never formed. */
int pm_chord_pad_vec(const float* tables, const float* chord_w, const float* chord_b, int32_t d, int32_t n_slots,
                     float* cvec /* [2][d] */, pm_stream_t stream);
int pm_chord_tables_fwd(const float* tables /* [4][131][d/2] */, const float* chord_w /* [d,15d] */, int32_t d, int32_t n_slots,
                        float* PT, const float* chord_b /* with cvec */, float* cvec /* [2][d] or NULL: as pm_chord_pad_vec, same launch */,
                        pm_stream_t stream);
int pm_chord_sum_fwd(const float* PT, const float* cvec, const int32_t* tokens /* [N,16,2] */, const uint8_t* is_drum, int32_t N,
                     int32_t d, int32_t n_slots, float* x0 /* [N,d] */, pm_stream_t stream);
int pm_chord_sum_bwd(const float* dy /* [N,d] */, const int32_t* tokens, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                     int32_t d /* multiple of 32, <= 512 */, int32_t n_slots, float* Gt, pm_stream_t stream);
int pm_chord_tables_bwd_w(const float* Gt, const float* tables, int32_t d, int32_t n_slots, float* d_chord_w /* += */,
                          float* d_chord_b /* += or NULL */, pm_stream_t stream);
int pm_chord_tables_bwd_x(const float* Gt, const float* chord_w, int32_t d, int32_t n_slots, float* S /* += */, pm_stream_t stream);
int pm_embed_tables_bwd(const float* S, const float* w_pitch_drum, const float* b_pitch_drum,
                        const float* w_pitch_nd, const float* b_pitch_nd, const float* w_dur, const float* b_dur,
                        const float* bn_drum_g, const float* bn_nd_g, const float* bn_dur_g, const float* stats,
                        const int32_t* tok_hist, int32_t d, float eps, float* dw_pitch_drum, float* db_pitch_drum,
                        float* dw_pitch_nd, float* db_pitch_nd, float* dw_dur, float* db_dur, float* dg_drum,
                        float* dbeta_drum, float* dg_nd, float* dbeta_nd, float* dg_dur, float* dbeta_dur,
                        pm_stream_t stream);

/* Synchronised form (data parallel, BatchNorm statistics over the global batch): call once with sums_out [4][2][d/2] fp64
 * (only the local sums {sum_v S[v], sum_v S[v] xhat[v]} per table and channel are written), add them and the token
 * histograms over the ranks, then call again with sums_global / hist_global (sums_out NULL).  All three NULL = the plain
 * form above. */
int pm_embed_tables_bwd_sync(const float* S, const float* w_pitch_drum, const float* b_pitch_drum, const float* w_pitch_nd,
                             const float* b_pitch_nd, const float* w_dur, const float* b_dur, const float* bn_drum_g,
                             const float* bn_nd_g, const float* bn_dur_g, const float* stats, const int32_t* tok_hist,
                             int32_t d, float eps, float* dw_pitch_drum, float* db_pitch_drum, float* dw_pitch_nd,
                             float* db_pitch_nd, float* dw_dur, float* db_dur, float* dg_drum, float* dbeta_drum,
                             float* dg_nd, float* dbeta_nd, float* dg_dur, float* dbeta_dur, double* sums_out,
                             const double* sums_global, const int32_t* hist_global, pm_stream_t stream);

/* ------------------------------------------------------------------ bar pooling / broadcast
 * PyG GlobalAttention over the nodes of each bar (model.py:335-340,408-409) and the
 * bar -> node broadcast `repeat_interleave(out, counts)` (model.py:543-545). */
int pm_gate_fwd(const float* x /* [N,d] */, const float* w /* [d] */, const float* b /* [1] */, int32_t N,
                int32_t d, float* g /* [N] */, pm_stream_t stream);
int pm_attnpool_fwd(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                    const float* bn_g, const float* bn_b, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                    int32_t d, float* alpha /* [N] softmax weights */, float* out /* [G,d] */, pm_stream_t stream);
int pm_attnpool_bwd(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                    const float* bn_g, const float* alpha, const float* dout /* [G,d] */, const float* gate_w,
                    const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, float* dx /* [N,d] */,
                    float* d_gate_w /* [d] += */, float* d_gate_b /* [1] += */, float* d_bn_g /* [1] += */,
                    float* d_bn_b /* [1] += */, float* scratch /* [3*N + 8] */,
                    const float* x_gate /* NULL, or the gate MLP's own input when it differs from x (dropout in front of
                                           the gate Linear, model.py:160): d_gate_w is then taken against it ... */,
                    float* dx_gate /* ... and the gradient w.r.t. it is written here instead of being added to dx */,
                    pm_stream_t stream);
/* The same in two phases for synchronised BatchNorm of the gate (BatchNorm1d(1) over ALL nodes, model.py:338): phase 1
 * leaves the two local sums {sum dgn, sum dgn * ghat} as fp64 at the head of `scratch` (the host adds them over the
 * ranks), phase 2 takes the global sums and node count. */
int pm_attnpool_bwd_sums(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                         const float* alpha, const float* dout, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                         int32_t d, float* scratch /* [3*N + 8] */, pm_stream_t stream);
int pm_attnpool_bwd_from_sums(const float* x, const float* g, const float* g_mean, const float* g_var, float eps,
                              const float* bn_g, const float* alpha, const float* dout, const float* gate_w,
                              const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, float* dx, float* d_gate_w,
                              float* d_gate_b, float* d_bn_g, float* d_bn_b, float* scratch, const float* x_gate,
                              float* dx_gate, const double* sums_global /* [2] */, double count_global,
                              pm_stream_t stream);
int pm_bar_broadcast_fwd(const float* bars /* [G,d] */, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                         int32_t d, float* x /* [N,d] */, pm_stream_t stream);
int pm_bar_broadcast_bwd(const float* dx /* [N,d] */, const int32_t* plan, int32_t N, int32_t E, int32_t G,
                         int32_t d, float* dbars /* [G,d] */, pm_stream_t stream);

/* ------------------------------------------------------------------ structure CNN (4x32 bar grids)
 * CNNEncoder / CNNDecoder convolutions (model.py:219-230,279-285): 3x3, padding 1, NCHW;
 * `up4` fuses nn.Upsample(scale_factor=(1,4)) in front of the convolution (model.py:280-281);
 * pool4 = nn.MaxPool2d((1,4)) (model.py:225). */
int pm_conv3x3_fwd(const float* x, const float* w, const float* b, int32_t G, int32_t Ci, int32_t Co, int32_t H,
                   int32_t W, int up4, float* y, pm_stream_t stream);
int pm_conv3x3_bwd_data(const float* dy, const float* w, int32_t G, int32_t Ci, int32_t Co, int32_t H, int32_t W,
                        int up4, float* dx, pm_stream_t stream);
int pm_conv3x3_bwd_weight(const float* x, const float* dy, int32_t G, int32_t Ci, int32_t Co, int32_t H, int32_t W,
                          int up4, float* dw /* += */, float* db /* += */, pm_stream_t stream);
int pm_maxpool4_fwd(const float* x, int64_t n_out, float* y, pm_stream_t stream);
int pm_maxpool4_bwd(const float* x, const float* dy, int64_t n_out, float* dx, pm_stream_t stream);

/* ------------------------------------------------------------------ loss (training.py:298-347)
 * Fused softmax cross-entropy of the pitch (131, ignore 130) and duration (99, ignore 98)
 * logits with their gradient, KL divergence, and the structure BCE.  `out` receives
 * {pitch, dur, structure, kld} (float64[4], zeroed by the call). */
/* db_* (optional, all or none, need d_logits): += column sums of d_logits over the drum rows / non-drum rows
 * (pitch block) and all rows (duration block) = gradients of the three un-embedding biases (model.py:561-567). */
int pm_content_ce(const float* c_logits /* [N,15,230] */, const int32_t* tokens /* [N,16,2] */,
                  const int32_t* tok_hist, const uint8_t* is_drum /* [N] or NULL */, int32_t N,
                  int32_t n_slots /* c_logits is [N,S,230] */, float grad_scale,
                  float* d_logits /* or NULL */, float* db_pitch_drum /* [131] or NULL */,
                  float* db_pitch_nd /* [131] */, float* db_dur /* [99] */, double* out, pm_stream_t stream);
/* Same with per-term gradient weights read on the device: dev_scale [2] = {pitch, duration} multiplies d_logits (not the
 * reported loss values).  Data parallel: with dev_scale = n_valid_local * world / n_valid_global the mean of the ranks'
 * gradients is the gradient of the token mean over the GLOBAL batch (training.py:316-323). */
int pm_content_ce_scaled(const float* c_logits, const int32_t* tokens, const int32_t* tok_hist, const uint8_t* is_drum,
                         int32_t N, int32_t n_slots, float grad_scale, const float* dev_scale /* [2] device, or NULL */,
                         float* d_logits, float* db_pitch_drum, float* db_pitch_nd, float* db_dur, double* out,
                         pm_stream_t stream);
/* Fused un-embedding + cross-entropy: the three Linear(d/2 -> 131 | 131 | 99) products of
 * ContentDecoder.forward (model.py:561-567; pitch per drum / non-drum row list of the plan, duration on all rows) and the
 * two CrossEntropyLoss(ignore_index = PAD) terms of `_losses` (training.py:316-323) in one launch: the logits of a
 * 64-row tile stay in the MFMA accumulators, only d_logits [N,S,230] is written (and `logits` when not NULL).
 * H [N,S,d] is the chord decoder's output; out[0] / out[1] receive the pitch / duration loss (zeroed by the call);
 * db_* (all or none) += the bias gradients; dev_scale as in pm_content_ce_scaled.
 * `w_planes`: scratch of pm_unembed_scratch_bytes(d) bytes (bf16 planes of the three weights + accumulator replicas);
 * with it and d/2 in {64, 128, 256} the products run on the bf16 matrix pipe (six products of exact three-way splits,
 * k_unembed_ce_planes); NULL or another width: the fp32-MFMA kernel of round 2. */
int64_t pm_unembed_scratch_bytes(int32_t d);
int pm_unembed_ce(const float* H, const float* w_pitch_drum /* [131,d/2] */, const float* b_pitch_drum,
                  const float* w_pitch_nd, const float* b_pitch_nd, const float* w_dur /* [99,d/2] */, const float* b_dur,
                  const int32_t* tokens, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots,
                  float grad_scale, const float* dev_scale, float* logits /* or NULL */, float* d_logits,
                  float* db_pitch_drum, float* db_pitch_nd, float* db_dur, double* out, uint16_t* w_planes /* or NULL */,
                  pm_stream_t stream);
/* Input gradient of the three un-embeddings (autograd of model.py:561-567) in one launch: dH[rows, :d/2] = d_logits[rows,
 * :131] @ W_pitch (drum / non-drum row lists of the plan), dH[:, d/2:] = d_logits[:, 131:] @ W_dur; d_logits [N,S,230] as
 * pm_unembed_ce leaves it, dH [N,S,d].  Six-product bf16 chain (fp32-exact products); d/2 in {64, 128, 256}.
 * `w_planes`: scratch of pm_unembed_dh_scratch_bytes(d) bytes; a call with prepare != 0 fills it from the weights (needed
 * once per parameter update), a call with prepare == 0 runs the product.  PM_E_UNSUPPORTED when N * n_slots * max(d, 230) * 4
 * >= 2^31 (32-bit byte offsets into d_logits and dH): the caller then runs the three products as GEMMs. */
int64_t pm_unembed_dh_scratch_bytes(int32_t d);
int pm_unembed_dh(const float* d_logits, const float* w_pitch_drum /* [131,d/2] */, const float* w_pitch_nd,
                  const float* w_dur /* [99,d/2] */, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                  int32_t n_slots, float* dH, uint16_t* w_planes, int32_t prepare, pm_stream_t stream);
/* Row lists of the decoder head WITHOUT the rows whose targets are PAD (CrossEntropyLoss(ignore_index), training.py:101-102,316-323:
 * no loss, zero gradient; 30 % of the active-slot rows at the bench's batches): row_lists [3][N*n_slots] — job 0 pitch of the drum
 * nodes' rows, 1 pitch of the other rows, 2 duration of all rows, each ascending; a row stays when its pitch OR its duration target
 * is not PAD —, pad_lists (or NULL) the rows left out in the same layout, row_counts [pm_unembed_row_counts_len(N, n_slots)]
 * (device: [0..2] the lengths of row_lists, [4..6] of pad_lists, the rest scratch), and zeros in the halves of `dH_zero`
 * [N*n_slots, d] (or NULL) that belong to the rows left out.  pm_unembed_ce_rows / pm_unembed_dh_rows are pm_unembed_ce /
 * pm_unembed_dh over such lists (row_counts: three lengths): logits, d_logits and dH of the other rows are not touched (the
 * weight-gradient products take the same lists as row maps). */
int64_t pm_unembed_row_counts_len(int32_t N, int32_t n_slots);
int pm_unembed_row_lists(const int32_t* tokens, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots,
                         int32_t* row_lists, int32_t* pad_lists, int32_t* row_counts, float* dH_zero, pm_stream_t stream);
int pm_unembed_ce_rows(const float* H, const float* w_pitch_drum, const float* b_pitch_drum, const float* w_pitch_nd,
                       const float* b_pitch_nd, const float* w_dur, const float* b_dur, const int32_t* tokens, const int32_t* plan,
                       int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots, float grad_scale, const float* dev_scale,
                       float* logits, float* d_logits, float* db_pitch_drum, float* db_pitch_nd, float* db_dur, double* out,
                       uint16_t* w_planes, const int32_t* row_lists, const int32_t* row_counts, pm_stream_t stream);
int pm_unembed_dh_rows(const float* d_logits, const float* w_pitch_drum, const float* w_pitch_nd, const float* w_dur,
                       const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d, int32_t n_slots, float* dH,
                       uint16_t* w_planes, const int32_t* row_lists, const int32_t* row_counts, pm_stream_t stream);
/* Weight gradients of the three un-embeddings in ONE launch (autograd of model.py:561-567): dw_pitch_drum / dw_pitch_nd [131, d/2],
 * dw_dur [99, d/2] += d_logits[rows, block]^T H[rows, half] over the plan's drum / non-drum rows and all rows, or over the lists of
 * pm_unembed_row_lists (row_lists / row_counts; NULL: the plan's).  d_logits [N*n_slots, 230], H [N*n_slots, d] (the un-embeddings'
 * input, fp32, 16-byte aligned); d/2 a multiple of 128; bf16 six-product chain, every row read once per 128 columns of d/2, output
 * blocks added with float atomics (not reproducible run to run: deterministic callers use pm_gemm_f32 with the same row maps). */
int pm_unembed_dw(const float* d_logits, const float* H, const int32_t* plan, int32_t N, int32_t E, int32_t G, int32_t d,
                  int32_t n_slots, float* dw_pitch_drum, float* dw_pitch_nd, float* dw_dur, const int32_t* row_lists,
                  const int32_t* row_counts, pm_stream_t stream);
/* Bias gradients (+=) of the three un-embeddings (model.py:561-567) from a given d(loss)/d(logits) [N, S, 230]: column sums
 * of the pitch columns over the drum nodes' / the other nodes' rows and of the duration columns over all rows. */
int pm_unembed_bias_grads(const float* d_logits, const uint8_t* is_drum /* [N] */, int32_t N, int32_t S,
                          float* db_pitch_drums /* [131] */, float* db_pitch_non_drums /* [131] */, float* db_dur /* [99] */,
                          pm_stream_t stream);
int pm_kld(const float* mu, const float* log_var, int32_t B, int32_t d, float beta, float* dmu /* or NULL, += */,
           float* dlog_var, double* out, pm_stream_t stream);
int pm_bce_logits(const float* logits, const float* target, int64_t n, float grad_scale, float* dlogits /* or NULL */,
                  double* out, pm_stream_t stream);

/* ------------------------------------------------------------------ evaluation metrics
 * `_accuracies` (training.py:349-497) as integer counts on the device (the reference takes 9 `.item()` syncs per batch):
 * counts[0..7] = {pitch correct, pitch not-PAD, pitch correct on drum nodes, pitch not-PAD on drum nodes,
 *                 duration correct, duration not-PAD, note (pitch and duration) correct, 0}; slots 1..15 of every node. */
int pm_content_accuracy(const float* c_logits /* [N,15,230] */, const int32_t* tokens /* [N,16,2] */,
                        const uint8_t* is_drum /* [N] */, int32_t N, int64_t* counts /* [8] */, pm_stream_t stream);
/* counts[0..3] = {prediction == target, true positives, predicted positives, target positives},
 * prediction = sigmoid(logit) >= 0.5 (training.py:470-497). */
int pm_structure_metrics(const float* s_logits, const float* s_target, int64_t n, int64_t* counts /* [4] */,
                         pm_stream_t stream);

/* ------------------------------------------------------------------ optimiser (train.py:181, training.py:160-166)
 * torch.optim.Adam (no weight decay, no amsgrad) over one flat fp32 buffer. */
int pm_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                 float beta1, float beta2, float eps, int32_t step, float grad_scale, pm_stream_t stream);

/* Gradient accumulation over `iters_to_accumulate` micro-batches (training.py:149,158: backward of tot_loss / k, optimizer
 * step every k-th batch): accum = (first ? 0 : accum) + scale * grads, scale = 1 / k. */
int pm_grad_accumulate(const float* grads, float* accum, int64_t n, float scale, int32_t first, pm_stream_t stream);

/* ------------------------------------------------------------------ data-parallel exchange
 * The reference is single-device (train.py:120-122).  Sum-all-reduce of the flat fp32 gradient buffer over RCCL / xGMI
 * for hosts that do not go through torch.distributed: rank 0 makes a 128-byte id (pm_comm_unique_id), the host hands it
 * to every rank, each rank creates its communicator on its current device (pm_comm_init) and then calls pm_allreduce —
 * in place, on `stream`, ordered after the kernels that wrote `buf`.  RCCL is loaded at run time; without it these four
 * return PM_E_UNSUPPORTED and nothing else is affected. */
int pm_comm_unique_id(uint8_t* id128 /* [128] host, out */);
int pm_comm_init(const uint8_t* id128 /* [128] host */, int32_t rank, int32_t world, void** comm /* out */);
int pm_comm_destroy(void* comm);
int pm_allreduce(float* buf, int64_t count, void* comm, pm_stream_t stream);

/* Counter-based dropout mask shared by device code and the oracle (host-callable). */
uint32_t pm_dropout_hash(uint32_t seed, uint32_t layer_uid, uint32_t eid, uint32_t channel);

/* ------------------------------------------------------------------ launch-duration profiler (bench.py roofline)
 * HIP events around GEMM / segment-reduce launches while enabled; pm_prof_end sums the durations per class.  Every event
 * costs ~4 us of GPU idle time, so pm_prof_configure selects WHICH launches are bracketed: the classes of `class_mask`
 * (bit c = class c), every `stride`-th launch of each (default: all classes, every launch).
 * classes 0..32 = GEMM tile configuration (0..10) * 3 + {0 NN, 1 NT, 2 TN}; 33 = segment-reduce forward; 34 = backward.
 * `work` = algorithmic flops (GEMM) or algorithmic HBM bytes (segment-reduce) of the launches. */
enum { PM_PROF_NCLASS_PUBLIC = 35 };
int pm_prof_configure(int64_t class_mask, int32_t stride);
int pm_prof_begin(int32_t max_events);
int pm_prof_end(double* ms /* [35] host */, double* work /* [35] host */, int64_t* count /* [35] host */);

/* ------------------------------------------------------------------ native training step
 * The whole of `PolyphemusTrainer.train`'s inner iteration (training.py:137-166) issued from C++:
 * ~340 kernel launches per step with no interpreter between them.  The model is described by
 * offsets (in floats) into the caller's flat parameter / buffer / gradient arrays, named after the
 * reference's modules; activations live in a caller-provided workspace arena.
 * Four calls so that the data-parallel gradient buckets can be exchanged as soon as they are final:
 *   pm_vae_step_forward  : plan -> encoder -> reparam -> decoder -> losses (+ d loss / d logits)
 *   pm_vae_step_backward_decoder : decoder gradients are final afterwards
 *   pm_vae_step_backward_encoder : encoder head, attention pooling, graph encoder: the gradients of every encoder
 *                                  parameter from `c_encoder.graph_encoder` to the end of the encoder are final
 *   pm_vae_step_backward_encoder_tail : chord encoder, embeddings, structure encoder: all gradients final
 * `state` is a caller-owned HOST blob of pm_vae_step_state_bytes() that carries the saved-activation
 * pointers between the calls.  Training mode only (batch statistics, running stats updated). */
#define PM_MAX_LAYERS 16
typedef struct PmLin { int64_t w, b; } PmLin;              /* offsets into params (and grads)            */
typedef struct PmBn { int64_t w, b, rm, rv; } PmBn;        /* w,b: params; rm,rv: running stats in buffers */
typedef struct PmGcn {
  int64_t nn_w, nn_b;                                       /* shared edge network Linear(32 -> d)          */
  int64_t weight[PM_MAX_LAYERS];                            /* [6,d,d] immediately followed by root [d,d]   */
  int64_t bias[PM_MAX_LAYERS];
  PmBn norm[PM_MAX_LAYERS];
} PmGcn;
typedef struct PmVaeLayout {
  int32_t d, n_bars, n_layers;
  int32_t flags;                                             /* bit 0: the model was built with batch_norm = False: the norms
                                                                of the two GCN stacks and of the two CNNs do not exist
                                                                (model.py:176-188,218-238,278-292; their PmBn entries are unused) */
  /* encoder (model.py:420-483) */
  PmLin enc_conv0; PmBn enc_bn1; PmLin enc_conv4; PmBn enc_bn5; PmLin enc_lin1, enc_lin4, enc_s_bars;
  PmLin enc_pitch_nd, enc_pitch_d, enc_dur; PmBn enc_bn_nd, enc_bn_d, enc_bn_dur; PmLin enc_chord;
  PmGcn enc_gcn; PmLin enc_gate; PmBn enc_gate_bn; PmLin enc_c_bars;
  PmLin enc_merge; PmBn enc_bn_merge; PmLin enc_mu, enc_lv;
  /* decoder (model.py:486-655) */
  PmLin dec_lin; PmBn dec_bn; PmLin dec_s_bars, dec_s_lin1, dec_s_lin4, dec_conv1; PmBn dec_bn2; PmLin dec_conv4;
  PmLin dec_c_bars; PmGcn dec_gcn; PmLin dec_chord, dec_pitch_d, dec_pitch_nd, dec_dur;
  float dropout;                                             /* cfg.dropout: p of the element dropout layers (model.py:160,199,
                                                                244-247,267-270,389-390,473,479,558-559,640); 0 = none */
  int32_t reserved;
} PmVaeLayout;
typedef struct PmBatch {                                    /* device pointers of one collated batch          */
  const int64_t* edge_index; const int32_t* edge_type; const int32_t* edge_dist;
  const int64_t* bars; const int64_t* batch; const uint8_t* is_drum; const int32_t* tokens; const float* s_tensor;
  int32_t N, E, G, B;
  int32_t n_slots;                                          /* active token slots S (1..15), see pm_plan_build */
  int32_t flags;                                            /* bit 0: every node receives edges of at most one track
                                                               relation (host-verified) -> compact GCL, K = 4d;
                                                               bit 1: GCL GEMM operands as pre-split bf16 planes;
                                                               bit 2: also store the content logits (pm_vae_step_outputs)
                                                               — the fused un-embedding + CE otherwise writes d_logits only;
                                                               bit 3: the CALLER computes the loss (pm_vae_step_set_output_grads
                                                               follows): the forward stores the logits and skips its own
                                                               cross-entropy / KLD / BCE and their gradients */
  const float* ce_scale;                                    /* NULL, or device [2]: weights of the pitch / duration CE
                                                               gradients (pm_content_ce_scaled; data-parallel token mean) */
} PmBatch;
/* The step's A/B switches (environment variables PM_GCL_FUSED, PM_GCL_NO_DW, PM_NO_ROWS_W, PM_GCL_NO_CLASSES, PM_FUSED_CE,
 * PM_SIDE_STREAM, PM_SIDE_DELAY_US, PM_DAGG_BN, PM_DAGG_RES, PM_PLAN_SIDE, PM_CHORD_TABLES, PM_H2, PM_BAR_ROUTE, PM_PAD_SKIP, PM_UNEMBED_DW, PM_SENC_FIRST, PM_GCL_OFFSET_LIMIT,
 * PM_DEBUG: the complete list, csrc/vae_step.hip read_cfg) are read once, when the library is loaded; this re-reads them (host
 * only) so that one process can run one batch through two kernel sets.  Switches of earlier rounds that are no longer read:
 * PM_NO_ROWS_TN, PM_NO_UNEMBED_DH, PM_GCL_NO_BFRAG, PM_DENSE_DEG, PM_LATE_WGRADS, PM_DW_SIDE, PM_FUSED_HEADS. */
int pm_vae_step_reload_switches(void);
int64_t pm_vae_layout_bytes(void);
int64_t pm_vae_step_state_bytes(void);
int64_t pm_vae_step_workspace_bytes(const PmVaeLayout* lay, int32_t N, int32_t E, int32_t G, int32_t B,
                                    int32_t n_slots);
/* Streams: the pm_vae_step_* calls are ordered on `stream` for the caller.  Internally the structure encoder / decoder
 * (model.py:211-299,434-445,500-505) and the parameter-only preparation (weight planes, distance tables) are issued on one
 * further non-blocking stream per device, created by the library on first use, between an event recorded on `stream` and an
 * event `stream` waits for before it reads their results (the structure encoder's backward: forked by
 * pm_vae_step_backward_encoder, joined by pm_vae_step_backward_encoder_tail).  No host synchronisation; capturable.
 * PM_SIDE_STREAM=0 keeps every launch on `stream`.  pm_vae_step_forward builds the plan itself (pm_plan_build). */
int pm_vae_step_forward(const PmVaeLayout* lay, const float* params, float* buffers, float* grads,
                        const PmBatch* batch, int32_t* plan, const float* eps /* [B,d] */, float msg_dropout,
                        uint32_t seed_enc, uint32_t seed_dec, float beta, int structure_loss_on_logits,
                        void* workspace, int64_t workspace_bytes, void* state, double* losses /* [4] dev */,
                        pm_stream_t stream);
/* Introspection of the last forward (host only): info = {compact GCL, bf16-planes GEMM operands, active slots S,
 * fragment-major weight planes built (B-direct GEMM mode), N, E, G, B, then the EFFECTIVE switches of the library —
 * fused un-embedding + cross-entropy (PM_FUSED_CE), second-stream site mask (PM_SIDE_STREAM), deterministic mode,
 * fused GCL kernels (PM_GCL_FUSED), norm backward inside the GCL input gradient (PM_DAGG_BN), chord encoder as table algebra (PM_CHORD_TABLES) —, GCL stacks in the fp16 pair format (bit 0 encoder, 1 decoder),
 * decoder head over the row lists without PAD targets (PM_PAD_SKIP)}.  The parity tests use it to assert that the golden-pinned step IS
 * the measured variant. */
int pm_vae_step_info(const void* state, int32_t* info /* [16] host */);
/* The model outputs of `VAE.forward` (model.py:665-678) as the last forward computed them, copied out of the arena
 * (async on `stream`; any pointer may be NULL): s_logits [G,4,32], c_logits [N,S,230] (active slots only), mu and
 * log_var [B,d].  Valid until the next pm_vae_step_forward on this state. */
int pm_vae_step_outputs(const void* state, float* s_logits, float* c_logits, float* mu, float* log_var,
                        pm_stream_t stream);
/* Introspection for parity tools (host only, no GPU work): where the last forward keeps an activation inside the workspace
 * the caller passed to pm_vae_step_forward — the tensors the backward takes its ReLU decisions from.  stack: 0 = the encoder's
 * GCN, 1 = the decoder's; *byte_offset is relative to the workspace pointer, *numel in floats. */
enum { PM_SAVED_GCN_H = 0,      /* [N,d] pre-norm output of GCL layer `layer` */
       PM_SAVED_GCN_X,          /* [N,d] layer input x_layer (layer = n_layers: the stack's output) */
       PM_SAVED_GCN_XIN,        /* [N,d] the same behind the cfg.dropout layer (= X when the model has no dropout) */
       PM_SAVED_GCN_MEAN, PM_SAVED_GCN_VAR,   /* [d] batch statistics of the layer's norm */
       PM_SAVED_GCN_T,          /* [32,d] distance table of the stack's edge network */
       PM_SAVED_X0,             /* [N,d] chord encoder output (after its ReLU) */
       PM_SAVED_MERGE_PRE, PM_SAVED_MERGE_MEAN, PM_SAVED_MERGE_VAR,   /* encoder.bn_linear_merge: input [B,d], statistics [d] */
       PM_SAVED_DEC_PRE, PM_SAVED_DEC_MEAN, PM_SAVED_DEC_VAR,         /* decoder.batch_norm: input [B,2d], statistics [2d] */
       PM_SAVED_ENC_CNN_LIN1 }; /* [G,d] CNNEncoder.lin[1] output (after its ReLU) */
int pm_vae_step_saved(const void* state, int32_t what, int32_t stack, int32_t layer, int64_t* byte_offset, int64_t* numel);
/* out[r, c] = 1 where the backward of a norm followed by a ReLU lets the gradient through: (x - mean) * rsqrt(var + eps) *
 * gamma + beta > 0, evaluated by the expression the backward kernels use (csrc/common.h pm_bn_bwd_elem). */
int pm_bn_relu_decisions(const float* x /* [rows,C] */, const float* mean, const float* var, const float* gamma, const float* beta,
                         float eps, int64_t rows, int32_t C, uint8_t* out /* [rows,C] */, pm_stream_t stream);
/* The drop-in module's path — `model(graph)` called by the unchanged training.py:137-166, which computes `_losses` itself
 * and calls `backward()`: between pm_vae_step_forward (PmBatch.flags bit 2: keep the logits) / pm_vae_step_outputs and the
 * four backward calls the caller hands in the gradients of the four outputs (any may be NULL = zero; d_c_logits covers the
 * step's S slots, [N, S, 230]; d_s_logits != NULL switches the structure decoder's backward on).  They replace the
 * gradients of the step's own loss kernels. */
int pm_vae_step_set_output_grads(void* state, const float* d_s_logits /* [G,4,32] */, const float* d_c_logits /* [N,S,230] */,
                                 const float* d_mu /* [B,d] */, const float* d_log_var /* [B,d] */, pm_stream_t stream);
/* The outputs of the last forward and the buffers their gradients are read from, as byte offsets into the workspace given to
 * pm_vae_step_forward (host only): [0..3] = s_logits [G,4,32], c_logits [N,S,230], mu [B,d], log_var [B,d]; [4..7] = the
 * gradient buffers in the same order.  A caller that writes a gradient into its buffer passes that pointer to
 * pm_vae_step_set_output_grads (no copy); any other d_c_logits pointer is used where it lies (no copy either: it must stay
 * alive, 16-byte aligned, until the four backward calls have run). */
int pm_vae_step_output_views(const void* state, int64_t* byte_offset /* [8] */, int64_t* numel /* [8] */);
/* ORDERING CONTRACT of the decoder's gradients: pm_vae_step_backward_decoder returns with the weight gradients of the decoder's
 * head (un-embeddings, chord decoder, the two head products) still running on the library's second stream, NOT yet joined to
 * `stream`.  They are final for the caller only behind pm_vae_step_join_decoder_grads, pm_vae_step_backward_encoder_heads or
 * pm_vae_step_backward_encoder (each makes `stream` wait for them).  A host that reads, synchronises on, or all-reduces the
 * decoder's gradients directly after this call MUST call pm_vae_step_join_decoder_grads first. */
int pm_vae_step_backward_decoder(void* state, pm_stream_t stream);
/* Data parallel: the weight gradients of the decoder's head run on the library's second stream beside the head chain and
 * are joined inside pm_vae_step_backward_encoder; a caller that hands the decoder's gradient bucket to an all-reduce
 * between the two calls makes `stream` wait for them with this call first (no-op when nothing is open). */
int pm_vae_step_join_decoder_grads(void* state, pm_stream_t stream);
/* The head chain of the encoder backward as a call of its own (optional: pm_vae_step_backward_encoder runs it when it has
 * not been called).  It returns with `stream` waiting for the decoder's weight gradients WITHOUT a stall (they have ~0.5 ms
 * of head chain beside them): a data-parallel caller launches the decoder's gradient bucket behind it, in front of the
 * encoder's GCN stack. */
int pm_vae_step_backward_encoder_heads(void* state, pm_stream_t stream);
int pm_vae_step_backward_encoder(void* state, pm_stream_t stream);
int pm_vae_step_backward_encoder_tail(void* state, pm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* POLYPHEMUS_HIP_H */
