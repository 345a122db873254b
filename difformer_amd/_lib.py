"""ctypes binding of libdifformer_hip.so (C ABI: include/difformer_hip.h).

There is deliberately no fallback: if the HIP library is missing or does not
export the expected ABI the import of any operator fails loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFFORMER_HIP_LIB: another build of the SAME ABI (A/B kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("DIFFORMER_HIP_LIB") or os.path.join(_HERE, "lib", "libdifformer_hip.so")
ABI_VERSION = 2

c_i64, c_int, c_f32, c_vp, c_sz = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/difformer_hip.h one to one
SIGNATURES = {
    "dif_version": (c_int, []),
    "dif_set_exact_fp32": (c_int, [c_int]),
    "dif_last_error": (ctypes.c_char_p, []),
    "dif_simple_reduced_len": (c_sz, [c_int, c_int, c_int]),
    "dif_simple_workspace_bytes": (c_sz, [c_i64, c_int, c_int, c_int]),
    "dif_simple_reduce_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_int,
                                      c_vp, c_vp, c_sz, c_vp]),
    "dif_simple_apply_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp]),
    "dif_project_reduce_workspace_bytes": (c_sz, [c_i64, c_int, c_int]),
    "dif_project_reduce_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                                       c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "dif_simple_bwd_workspace_bytes": (c_sz, [c_i64, c_int, c_int, c_int]),
    "dif_simple_bwd_prep_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_int,
                                        c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dif_rowgemm_f32": (c_int, [c_vp, c_i64, c_vp, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp, c_f32, c_vp, c_i64,
                                c_vp, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp]),
    "dif_sigmoid_workspace_bytes": (c_sz, [c_i64, c_i64, c_int, c_int, c_int]),
    "dif_sigmoid_attn_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int,
                                     c_vp, c_i64, c_vp, c_sz, c_vp]),
    "dif_sigmoid_attn_fwd_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int,
                                         c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "dif_sigmoid_bwd_workspace_bytes": (c_sz, [c_i64, c_i64, c_int, c_int, c_int]),
    "dif_sigmoid_attn_bwd_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64,
                                         c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_sz, c_vp]),
    "dif_batched_simple_workspace_bytes": (c_sz, []),
    "dif_batched_simple_attn_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_int,
                                            c_vp, c_i64, c_vp, c_sz, c_vp]),
    "dif_batched_simple_attn_fwd_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_int,
                                                c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "dif_batched_simple_raw_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_int,
                                           c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp]),
    "dif_batched_sigmoid_attn_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                             c_int, c_vp, c_i64, c_vp]),
    "dif_batched_sigmoid_attn_fwd_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                                 c_int, c_vp, c_i64, c_vp, c_vp]),
    "dif_batched_sigmoid_bwd_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dif_batched_sigmoid_attn_bwd_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp,
                                                 c_int, c_int, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                                 c_vp, c_sz, c_vp]),
    "dif_csr_workspace_bytes": (c_sz, [c_i64, c_i64, c_int]),
    "dif_csr_build": (c_int, [c_vp, c_i64, c_i64, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz,
                              c_vp]),
    "dif_gcn_spmm_part_scratch_bytes": (c_sz, [c_i64, c_i64, c_int]),
    "dif_gcn_spmm_part_f32": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_int,
                                      c_vp, c_i64, c_f32, c_f32, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_i64, c_f32,
                                      c_vp, c_vp, c_f32, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp, c_i64, c_vp]),
    "dif_subgraph_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "dif_subgraph": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dif_subgraph_batches_workspace_bytes": (c_sz, [c_i64, c_i64, c_int]),
    "dif_subgraph_batches_group": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dif_subgraph_batches_emit": (c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dif_graph_prepare_workspace_bytes": (c_sz, [c_i64, c_i64, c_int]),
    "dif_graph_prepare": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dif_subgraph_batches_csr_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "dif_subgraph_batches_csr": (c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_sz, c_vp, c_vp, c_vp, c_vp,
                                         c_sz, c_vp]),
    "dif_gcn_spmm_f32": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_int,
                                 c_vp, c_i64, c_f32, c_f32, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dif_gcn_edge_weight_grad_f32": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_f32, c_vp, c_vp]),
    "dif_gram_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dif_gram_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dif_simple_coeffs_len": (c_sz, [c_int, c_int]),
    "dif_simple_coeffs_f32": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp]),
    "dif_closed_form_attn_bwd_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp,
                                             c_i64, c_vp, c_vp, c_vp]),
    "dif_closed_form_attn_bwd_groups": (c_int, [c_i64]),
    "dif_simple_coeffs_bwd_len": (c_sz, [c_int, c_int]),
    "dif_simple_coeffs_bwd_f32": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp,
                                          c_vp]),
    "dif_simple_layer_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_vp,
                                     c_i64, c_int, c_f32, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                                     c_vp, c_sz, c_vp]),
    "dif_sliced_plan": (c_int, [c_i64, c_i64, c_int, c_vp]),
    "dif_sliced_measure": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp,
                                   c_vp, c_vp, c_vp, c_vp]),
    "dif_sliced_emit": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64,
                                c_vp, c_vp]),
    "dif_sliced_prescale_f32": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "dif_sliced_spmm_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp,
                                    c_i64, c_f32, c_f32, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dif_sliced_spmm_workspace_bytes": (c_i64, [c_i64, c_i64, c_int]),
    "dif_wide_partials": (c_i64, [c_int]),
    "dif_wide_gram_f64": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "dif_wide_scale_f64": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "dif_row_order_workspace_bytes": (c_sz, [c_i64]),
    "dif_row_order": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dif_simple_layer_head_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_vp, c_i64,
                                          c_int, c_f32, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_i64, c_vp]),
    "dif_gram_bg_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dif_gram_bg_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dif_simple_coeffs_bg_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_f32, c_vp, c_vp, c_vp]),
    "dif_gram_sym_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_sz, c_vp]),
    "dif_layer_tail_mix_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_f32, c_vp, c_i64, c_f32, c_vp, c_vp, c_i64, c_int, c_vp, c_i64,
                                       c_vp, c_i64, c_f32, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp]),
    "dif_layer_tail_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_f32,
                                   c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp]),
    "dif_layer_tail_bwd_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dif_layer_tail_bwd_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_f32, c_vp, c_vp, c_f32,
                                       c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "dif_linear_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp]),
    "dif_input_gram_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp, c_vp, c_vp,
                                   c_vp, c_vp, c_sz, c_vp]),
    "dif_simple_layer_wide_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_int, c_vp, c_f32, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32,
                                          c_vp, c_i64, c_int, c_f32, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp]),
    "dif_xwide_packed_bytes": (c_i64, [c_int, c_int]),
    "dif_xwide_pack_f32": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    "dif_simple_layer_xwide_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_f32, c_vp, c_i64, c_vp, c_vp,
                                           c_f32, c_vp, c_i64, c_int, c_f32, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp]),
    "dif_gram_coeffs_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp,
                                    c_sz, c_vp]),
    "dif_linear_xwide_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp]),
    "dif_gram128_workspace_bytes": (c_sz, [c_i64, c_int]),
    "dif_gram128_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_sz, c_vp]),
    "dif_linear_packed_bytes": (c_i64, [c_int]),
    "dif_linear_pack_f32": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "dif_linear_packed_f32": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp]),
    "dif_gcn_spmm_tail_f32": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_int,
                                      c_vp, c_i64, c_f32, c_f32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_f32, c_vp, c_vp, c_f32,
                                      c_int, c_vp, c_i64, c_vp]),
}
# bfloat16 storage variants share the argument lists of their float32 twins
for _n in ("dif_linear", "dif_project_reduce", "dif_simple_reduce", "dif_simple_apply", "dif_layer_tail", "dif_sigmoid_attn"):
    SIGNATURES[_n + "_bf16"] = SIGNATURES[_n + "_f32"]
SIGNATURES["dif_gcn_spmm_part_bf16"] = SIGNATURES["dif_gcn_spmm_part_f32"]
SIGNATURES["dif_simple_layer_head_bf16"] = SIGNATURES["dif_simple_layer_head_f32"]
SIGNATURES["dif_simple_layer_gather_f32"] = (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32,
                                                     c_vp, c_i64, c_int, c_f32, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp, c_vp,
                                                     c_int, c_vp, c_i64, c_vp])
SIGNATURES["dif_simple_layer_gather_bf16"] = SIGNATURES["dif_simple_layer_gather_f32"]
SIGNATURES["dif_gram_bf16"] = (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_sz, c_vp])
SIGNATURES["dif_simple_layer_bf16"] = (c_int, [c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_vp,
                                               c_i64, c_int, c_f32, c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp])
SIGNATURES["dif_gcn_spmm_tail_bf16"] = (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64,
                                                c_int, c_vp, c_i64, c_f32, c_f32, c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_i64, c_f32,
                                                c_vp, c_vp, c_f32, c_int, c_vp, c_i64, c_vp])



class TinyCfg(ctypes.Structure):
    """dif_tiny_cfg of include/difformer_hip.h."""
    _fields_ = [(k, ctypes.c_int32) for k in ("n", "in_channels", "hidden", "out_channels", "num_layers", "kernel", "use_bn",
                                               "use_residual", "use_weight", "use_graph", "use_source", "training")] + \
               [(k, ctypes.c_float) for k in ("alpha", "attn_scale", "gcn_scale", "dropout", "eps")] + \
               [("launch_plan", ctypes.c_int32), ("nnz", ctypes.c_int64)]


SIGNATURES["dif_wide_coeffs_f64"] = (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp])
SIGNATURES["dif_gram_sym_workspace_bytes"] = (c_sz, [c_i64, c_int])
SIGNATURES["dif_tiny_tape_floats"] = (c_sz, [c_int, c_int, c_int])
SIGNATURES["dif_tiny_scratch_floats"] = (c_sz, [c_int, c_int, c_int])
SIGNATURES["dif_tiny_graph_workspace_bytes"] = (c_sz, [c_i64, c_i64])
SIGNATURES["dif_tiny_graph_build"] = (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp])
SIGNATURES["dif_tiny_forward_f32"] = (c_int, [ctypes.POINTER(TinyCfg), c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp])
SIGNATURES["dif_tiny_backward_f32"] = (c_int, [ctypes.POINTER(TinyCfg), c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                               c_vp, c_vp, c_vp])

_lib = None


class DifformerHipError(RuntimeError):
    """A C-ABI call returned non-zero (argument rejected or HIP runtime failure)."""

    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


def _single_hip_runtime():
    """Two HIP runtimes in one process (torch's bundled one + a system one) do not share
    streams or allocations; refuse to run in that state."""
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    if len(paths) > 1:
        raise ImportError(f"difformer_amd: more than one HIP runtime mapped: {sorted(paths)}; "
                          "import torch before difformer_amd so that both share torch's libamdhip64")


def load():
    """Load the shared library once; raise ImportError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"difformer_amd: HIP extension not built ({LIB_PATH} missing). Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C difformer_amd/csrc`. "
            "There is no CPU / eager fallback.")
    import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so the SONAME is shared)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"difformer_amd: {LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    ver = lib.dif_version()
    if ver != ABI_VERSION:
        raise ImportError(f"difformer_amd: ABI version mismatch (library {ver}, host {ABI_VERSION}); rebuild")
    _single_hip_runtime()
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dif_last_error()
        raise DifformerHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}", rc)
