"""ctypes binding of libpolyphemus_hip.so (the C ABI in include/polyphemus_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, the
product path raises.  torch is used only for device memory and the stream handle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_LIB_PATH: development only — load a differently built library (kernel A/B variants of tools/build_variants.py)
LIB_PATH = os.environ.get("PM_LIB_PATH") or os.path.join(_HERE, "libpolyphemus_hip.so")

PROF_NCLASS = 40        # prof.h: 33 GEMM classes, segreduce_fwd, segreduce_bwd, gcl_fwd, gcl_dagg, gcl_dw, rows_w, rows_tn
PLAN_FIELDS = ["rowptr", "csr_src", "csr_dist", "csr_eid", "colptr", "csc_dst", "csc_reldist", "csc_eid",
               "csc_invcnt", "node_bar", "bar_ptr", "group_list", "group_cnt", "tok_hist", "row_list", "node_trel", "trk_list", "trk_cnt", "scratch"]

# argument codes: p = device pointer, i = int32, l = int64, f = float, u = uint32, s = stream
_SIGS = {
    "pm_plan_layout": "iiip",
    "pm_gcl_tile_order": "piipi",
    "pm_row_tile_order": "ipi",
    "pm_plan_build": "pppppppiiiiips",
    "pm_edge_attrs_to_ids": "pipps",
    "pm_tokens_from_onehot": "pips",
    "pm_batch_flags": "pppiipps",
    "pm_edge_table": "ppips",
    "pm_edge_table_bwd": "pipps",
    "pm_graph_count": "pippppps",
    "pm_graph_emit": "piippllppppppps",
    "pm_binary_from_logits": "pifpps",
    "pm_mtp_from_logits": "ppilppps",
    "pm_segreduce_fwd": "pppiiiifuuips",
    "pm_segreduce_fwd_planes": "pppiiiifuuipls",
    "pm_gcl_forward_fused": "pppiiiifuuppipppls",
    "pm_gcl_input_grad_fused": "plpiiiipips",
    "pm_gcl_input_grad_bn": "pplpiiiipips",
    "pm_chord_pad_vec": "pppiips",
    "pm_chord_tables_fwd": "ppiippps",
    "pm_chord_sum_fwd": "ppppiiips",
    "pm_chord_sum_bwd": "pppiiiiips",
    "pm_chord_tables_bwd_w": "ppiipps",
    "pm_chord_tables_bwd_x": "ppiips",
    "pm_bn_bwd_sums": "ppiippfppips",
    "pm_gcl_weight_grad_fused": "plplpiiiiips",
    "pm_gcl_forward_from_planes": "plpiiiippipps",
    "pm_rows_times_weight": "piiipiiippis",
    "pm_rows_times_weight_longk": "piiipiiipis",
    "pm_rows_tn_weight_grad": "piipiiipips",
    "pm_segreduce_bwd": "pppppiiiifuuipps",
    "pm_segreduce_bwd_norm": "pppppiiiifuuippps",
    "pm_gemm_f32": "iiiiipipipipiipips",
    "pm_gemm_f32_grouped": "iiiiipipipipiipipilllliis",
    "pm_gemm_f32_desc": "ps",
    "pm_gemm_config": "iiii",
    "pm_gemm_force_config": "i",
    "pm_bn_stats": "piiippppfps",
    "pm_bn_small_fwd": "piifpppipppppfs",
    "pm_bn_small_bwd": "ppiippfppipppps",
    "pm_bn_apply": "piiippfpppips",
    "pm_bn_partial_sums": "ppiiippfppipps",
    "pm_bn_stats_from_sums": "pDippppfs",
    "pm_bn_bwd_from_sums": "ppiiippfppippDppppps",
    "pm_bn_bwd": "ppiiippfppippppps",
    "pm_bn_apply_fused": "piipfpppipppppfs",
    "pm_bn_bwd_fused": "ppiippfppipppppplis",
    "pm_split_planes": "plpls",
    "pm_split_planes_frag": "piiiillps",
    "pm_relu_bwd": "pplps",
    "pm_add": "pplps",
    "pm_bn_counters_update": "ppppis",
    "pm_grad_accumulate": "pplfis",
    "pm_colsum_acc": "piiips",
    "pm_colsum_rows_acc": "piipipips",
    "pm_reparam_fwd": "ppplps",
    "pm_reparam_bwd": "ppplpps",
    "pm_embed_tables": "pppppppppppppppppppiiffpps",
    "pm_embed_gather": "pppiiips",
    "pm_chord_pad_fwd": "ppppiiipps",
    "pm_chord_pad_bwd": "ppiiippppps",
    "pm_embed_bwd_scatter": "pppiiiiips",
    "pm_embed_tables_bwd": "ppppppppppppifpppppppppppps",
    "pm_embed_tables_bwd_sync": "ppppppppppppifppppppppppppppps",
    "pm_gate_fwd": "pppiips",
    "pm_attnpool_fwd": "ppppfpppiiiipps",
    "pm_attnpool_bwd": "ppppfpppppiiiipppppppps",
    "pm_attnpool_bwd_sums": "ppppfpppiiiips",
    "pm_attnpool_bwd_from_sums": "ppppfpppppiiiipppppppppDs",
    "pm_relu_residual_fwd": "pplps",
    "pm_bn_fold_weights": "piipppppfpps",
    "pm_dropout_rows": "plifuups",
    "pm_bar_broadcast_fwd": "ppiiiips",
    "pm_bar_broadcast_bwd": "ppiiiips",
    "pm_conv3x3_fwd": "pppiiiiiips",
    "pm_conv3x3_bwd_data": "ppiiiiiips",
    "pm_conv3x3_bwd_weight": "ppiiiiiipps",
    "pm_maxpool4_fwd": "plps",
    "pm_maxpool4_bwd": "pplps",
    "pm_content_ce": "ppppiifppppps",
    "pm_content_ce_scaled": "ppppiifpppppps",
    "pm_unembed_ce": "pppppppppiiiiifpppppppps",
    "pm_unembed_dh": "pppppiiiiippis",
    "pm_unembed_dh_scratch_bytes": "i",
    "pm_unembed_row_counts_len": "ii",
    "pm_unembed_dw": "pppiiiiippppps",
    "pm_unembed_row_lists": "ppiiiiipppps",
    "pm_unembed_ce_rows": "pppppppppiiiiifpppppppppps",
    "pm_unembed_dh_rows": "pppppiiiiipppps",
    "pm_unembed_scratch_bytes": "i",
    "pm_kld": "ppiifppps",
    "pm_unembed_bias_grads": "ppiippps",
    "pm_bce_logits": "pplfpps",
    "pm_content_accuracy": "pppips",
    "pm_structure_metrics": "pplps",
    "pm_adam_step": "pppplffffifs",
    "pm_comm_unique_id": "p",
    "pm_comm_init": "piip",
    "pm_comm_destroy": "p",
    "pm_allreduce": "plps",
    "pm_prof_configure": "li",
    "pm_prof_begin": "i",
    "pm_prof_end": "ppp",
    "pm_vae_step_workspace_bytes": "piiiii",
    "pm_vae_step_forward": "pppppppfuufiplpps",
    "pm_vae_step_info": "pp",
    "pm_vae_step_outputs": "ppppps".replace(" ", ""),
    "pm_vae_step_set_output_grads": "ppppps",
    "pm_vae_step_saved": "piiipp",
    "pm_vae_step_output_views": "ppp",
    "pm_bn_relu_decisions": "pppppflips",
    "pm_vae_step_backward_decoder": "ps",
    "pm_vae_step_backward_encoder": "ps",
    "pm_vae_step_backward_encoder_tail": "ps",
    "pm_vae_step_join_decoder_grads": "ps",
    "pm_vae_step_backward_encoder_heads": "ps",
    "pm_vae_step_reload_switches": "",
    "pm_relu_bwd_planes": "pplppls",
    "pm_absmax": "plps",
    "pm_split_planes_frag_h2": "piiiillfps",
    "pm_gcl_forward_fused_h2": "pppiiiifuuppippplps",
    "pm_gcl_input_grad_bn_h2": "pplpiiiipipps",
    "pm_gcl_weight_grad_fused_h2": "plplpiiiiippps",
    "pm_bn_bwd_fused_h2": "ppiippfppippppplips",
    "pm_gcl_input_grad_fused_h2": "plpiiiipippfs",
    "pm_bn_apply_fused_absmax": "piipfpppipppppfps",
    "pm_bar_aggregate_fwd": "pppiiiifuuplps",
    "pm_bar_aggregate_bwd": "pppppiiiifuuppps",
    "pm_gcl_forward_from_planes_h2": "plpiiiippipppfs",
    "pm_set_deterministic": "i",
    "pm_get_deterministic": "",
    "pm_deterministic_faults": "",
    "pm_h2_clamp_events": "i",
}
_CT = {"p": C.c_void_p, "i": C.c_int32, "l": C.c_int64, "f": C.c_float, "u": C.c_uint32, "s": C.c_void_p, "D": C.c_double}
_RET64 = {"pm_vae_step_workspace_bytes", "pm_vae_layout_bytes", "pm_vae_step_state_bytes", "pm_unembed_scratch_bytes",
          "pm_unembed_dh_scratch_bytes", "pm_unembed_row_counts_len"}
ABI_VERSION = 8          # PM_ABI_VERSION of include/polyphemus_hip.h this table was written against
EXPORTED = sorted(list(_SIGS) + ["pm_abi_version", "pm_build_info", "pm_dropout_hash", "pm_vae_layout_bytes",
                                 "pm_vae_step_state_bytes"])

_lib: Optional[C.CDLL] = None


class HipExtensionError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the HIP extension (once).  Raises HipExtensionError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                f"{LIB_PATH} is missing: build it with `python -m polyphemus_amd.build` "
                "(__graft_entry__.build()).  There is no CPU fallback for the HIP path.")
        L = C.CDLL(LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = [_CT[c] for c in sig]
            fn.restype = C.c_int64 if name in _RET64 else C.c_int
        L.pm_vae_layout_bytes.restype = C.c_int64
        L.pm_vae_step_state_bytes.restype = C.c_int64
        L.pm_abi_version.restype = C.c_int
        if L.pm_abi_version() != ABI_VERSION:      # struct layouts and argument lists below are those of ONE header version
            raise HipExtensionError(f"{LIB_PATH} has ABI version {L.pm_abi_version()}, this binding expects {ABI_VERSION}: "
                                    "rebuild it (`python -m polyphemus_amd.build --force`)")
        L.pm_build_info.restype = C.c_char_p
        L.pm_dropout_hash.argtypes = [C.c_uint32] * 4
        L.pm_dropout_hash.restype = C.c_uint32
        _lib = L
    return _lib


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_ERR = {-1: "PM_E_INVALID (bad argument)", -2: "PM_E_LAUNCH (kernel launch failed)", -3: "PM_E_UNSUPPORTED"}


def call(name: str, *args) -> None:
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise HipExtensionError(f"{name} failed: {_ERR.get(rc, rc)}")


def plan_layout(N: int, E: int, G: int):
    """Field offsets (int32 elements) of the plan buffer; host-only call."""
    off = (C.c_int64 * (len(PLAN_FIELDS) + 1))()
    rc = lib().pm_plan_layout(N, E, G, C.cast(off, C.c_void_p))
    if rc != 0:
        raise HipExtensionError(f"pm_plan_layout({N},{E},{G}) failed: {_ERR.get(rc, rc)}")
    return list(off)


def gcl_tile_order(trk_cnt, use_classes: bool, N: int):
    """Host-only: [(track group, first row, rows)] in workgroup order of the GCL products (csrc/tile_order.h); rows is 64,
    or 32 for half a tile; (-1, -1, 0) for workgroups that exit.  `trk_cnt`: the 32 ints of the plan's trk_cnt field."""
    tc = (C.c_int32 * 32)(*[int(v) for v in list(trk_cnt)[:32]])
    grid = lib().pm_gcl_tile_order(C.cast(tc, C.c_void_p), int(bool(use_classes)), int(N), None, 0)
    if grid < 0:
        raise HipExtensionError(f"pm_gcl_tile_order failed: {_ERR.get(grid, grid)}")
    out = (C.c_int32 * (3 * grid))()
    lib().pm_gcl_tile_order(C.cast(tc, C.c_void_p), int(bool(use_classes)), int(N), C.cast(out, C.c_void_p), grid)
    return [(out[3 * b], out[3 * b + 1], out[3 * b + 2]) for b in range(grid)]


def set_deterministic(on: bool) -> None:
    """Deterministic mode of the library (include/polyphemus_hip.h, pm_set_deterministic): every float atomic of the step
    is ordered, two runs on the same inputs are bit-identical.  Slower; for parity tests and debugging."""
    call("pm_set_deterministic", int(bool(on)))


def is_deterministic() -> bool:
    return bool(lib().pm_get_deterministic())


def deterministic_faults() -> int:
    """Gates that could not be set up + waves that timed out waiting for their turn since the library was loaded; 0 = every
    gated launch was ordered (synchronises the device)."""
    return int(lib().pm_deterministic_faults())


def h2_clamp_events(reset: bool = False) -> int:
    """Threads of the fp16-pair split kernels whose scaled value saturated at +-65504 (a gradient was clipped) since the library
    was loaded / the last reset; 0 = every operand fitted its scale (synchronises the device)."""
    return int(lib().pm_h2_clamp_events(1 if reset else 0))


class deterministic:
    """`with deterministic(True): ...` — the mode for the block, the previous mode restored afterwards (a process started with
    PM_DETERMINISTIC=1 stays in it)."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = is_deterministic()
        set_deterministic(self.on)
        return self

    def __exit__(self, *exc):
        set_deterministic(self.prev)
        return False
