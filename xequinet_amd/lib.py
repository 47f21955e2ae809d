"""ctypes binding of libxeq_hip.so (the C ABI declared in include/xeq.h).

There is NO CPU fallback: if the HIP library is missing the import of any op
raises, and every op rejects non-HIP tensors.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int32, c_int64, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XEQ_LIB_PATH") or os.path.join(_HERE, "libxeq_hip.so")   # XEQ_LIB_PATH: development builds

XEQ_F32, XEQ_F64 = 0, 1
COPY_MANY_MAX = 16       # XEQ_COPY_MANY_MAX of include/xeq.h
SB_ACCUM_VEC = 16        # XEQ_SB_ACCUM_VEC of include/xeq.h: xeq_message_bwd_sb adds dL/dvec to the buffer it is handed
XHAT_HIGHER_L_ZERO = 2   # XEQ_XHAT_HIGHER_L_ZERO of include/xeq.h: hint bit on the xhat_layout argument of the wq message kernels
SB_Y0_ZERO = 8           # XEQ_SB_Y0_ZERO / XEQ_SB_Q_ACCUMULATE: the training-pass forms of the sb message kernels (ops.DiffMessage)
SB_Q_ACCUMULATE = 16
SB_NO_GY = 32
WQ_MIRROR_WALK = 4       # XEQ_WQ_MIRROR_WALK: the reverse wq kernel walks the forward plan of a symmetric list
WQ_PACKED_WEIGHTS = 64   # XEQ_WQ_PACKED_WEIGHTS: w_rbf of the wq message kernels is the packed copy (xeq_message_wq_pack_weights)
WQ_MAX_PART_SETS = 8     # XEQ_WQ_MAX_PART_SETS: sets of per-block partials one xeq_message_wq_edge_grad_sum launch adds up
RBF_KINDS = {"bessel": 0, "gaussian": 1, "expbern": 2, "expnorm": 3}
CUTOFF_KINDS = {"cosine": 0, "polynomial": 1}

_P = c_void_p
_I3 = ctypes.POINTER(c_int32)

# name -> argtypes, exactly the prototypes of include/xeq.h
_PROTOS = {
    "xeq_launch_count": [],
    "xeq_launch_names": [c_int64, c_char_p, c_int64],
    "xeq_csr_rowptr": [_P, c_int64, c_int64, _P, _P],
    "xeq_csr_by_key_workspace": [c_int64, c_int64],
    "xeq_csr_by_key": [_P, c_int64, c_int64, _P, c_int64, _P, _P, _P],
    "xeq_csr_by_key_bounded": [_P, c_int64, c_int64, _P, _P, c_int64, _P, _P, _P],
    "xeq_exclusive_scan_i32": [_P, c_int64, _P, _P],
    "xeq_exclusive_scan_i32_workspace": [c_int64],
    "xeq_exclusive_scan_i32_ws": [_P, c_int64, _P, _P, c_int64, _P],
    "xeq_reverse_edge_map": [_P, c_int64, c_int64, _P, _P, _P],
    "xeq_reverse_edge_map_pbc": [c_int, _P, _P, c_int64, c_int64, _P, _P, _P],
    "xeq_radius_graph_count": [c_int, _P, _P, c_int64, c_int64, c_double, _P, _P],
    "xeq_radius_graph_fill": [c_int, _P, _P, c_int64, c_int64, c_double, _P, c_int64, _P, _P],
    "xeq_radius_graph_bin_ids": [c_int, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P],
    "xeq_radius_graph_count_cl": [c_int, _P, _P, c_int64, c_int64, c_double, _P, _P, _P, _P, _P, _P, _P, _P],
    "xeq_radius_graph_fill_cl": [c_int, _P, _P, c_int64, c_int64, c_double, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P],
    "xeq_pbc_image_counts": [c_int, _P, c_int64, _I3, c_double, _I3],
    "xeq_pbc_tables_host": [c_int, _P, c_int64, _I3, c_double, _P, c_int64],
    "xeq_pbc_wrap": [c_int, _P, _P, c_int64, c_int64, _P, _P, _I3, _P, _P, _P],
    "xeq_radius_graph_pbc_count": [c_int, _P, _P, c_int64, c_int64, _P, c_int64, c_double, _P, _P],
    "xeq_radius_graph_pbc_fill": [c_int, _P, _P, c_int64, c_int64, _P, _P, _P, c_int64, c_double, _P, c_int64, _P, _P, _P],
    "xeq_radius_graph_pbc_count_pruned": [c_int, _P, _P, c_int64, c_int64, _P, c_int64, c_double, _P, _P, _I3, _P, _P],
    "xeq_radius_graph_pbc_fill_pruned": [c_int, _P, _P, c_int64, c_int64, _P, _P, _P, c_int64, c_double, _P, _P, _I3, _P, c_int64,
                                         _P, _P, _P],
    "xeq_radius_graph_pbc_bin_ids": [c_int, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P],
    "xeq_radius_graph_pbc_count_cl": [c_int, _P, _P, c_int64, c_int64, _P, c_int64, c_double, _P, _P, _I3, _P, _P, _P, _P, _P, _P],
    "xeq_radius_graph_pbc_fill_cl": [c_int, _P, _P, c_int64, c_int64, _P, _P, _P, c_int64, c_double, _P, _P, _I3, _P, _P, _P, _P,
                                     _P, c_int64, _P, _P, _P, _P],
    "xeq_edge_vectors_fwd": [c_int, _P, _P, c_int64, _P, _P, _P, _P, _P, _P],
    "xeq_edge_vectors_bwd": [c_int, _P, c_int64, _P, _P, _P, _P, _P, _P],
    "xeq_sph_harm_fwd": [c_int, _P, c_int64, _I3, c_int, _P, _P],
    "xeq_sph_harm_bwd": [c_int, _P, _P, c_int64, _I3, c_int, _P, _P],
    "xeq_radial_fwd": [c_int, _P, c_int64, c_int, c_int, c_int, c_double, _P, _P, _P, _P, _P],
    "xeq_elementwise_tp_fwd": [c_int, _P, _P, c_int64, c_int64, _I3, _P, _P],
    "xeq_channel_dot_fwd": [c_int, _P, _P, c_int64, _I3, _P, _P],
    "xeq_eqln_fwd": [c_int, _P, _P, _P, c_int64, _I3, c_double, _P, _P],
    "xeq_eqln_bwd": [c_int, _P, _P, _P, c_int64, _I3, c_double, _P, _P],
    "xeq_segment_sum": [c_int, _P, _P, c_int64, c_int64, _P, _P],
    "xeq_linear_supported": [c_int, c_int, c_int],
    "xeq_mfma_order_probe": [_P, _P, _P, _P],
    "xeq_linear_fwd": [_P, c_int64, c_int64, c_int, _P, _P, c_int, c_int, c_int, _P, _P, c_int64, _P],
    "xeq_head_dot": [_P, c_int64, c_int, _P, _P, _P, _P],
    "xeq_head_bwd_hidden": [_P, c_int64, c_int, _P, _P, _P, _P],
    "xeq_head_supported": [c_int, c_int, c_int],
    "xeq_head_fwd": [_P, c_int64, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P, _P],
    "xeq_head_bwd": [_P, c_int64, c_int, _P, c_int64, _P, c_int64, _P, _P, _P],
    "xeq_first_block_front": [_P, c_int, c_int64, c_int64, _P, _P, _P, c_int, c_int, c_int64, _P, _P, _P, _P],
    "xeq_rowptr_from_degrees": [_P, c_int64, c_int64, _P, _P, _P, _P],
    "xeq_rowptr_from_degrees_max": [],
    "xeq_wgrad_chunks": [c_int64, c_int, c_int],
    "xeq_wgrad": [_P, c_int64, _P, c_int64, c_int64, c_int, c_int, c_int, c_int, _P, _P],
    "xeq_load_padded_batch": [c_int, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_double, c_double, _P, _P, _P, _P, _P],
    "xeq_load_padded_batch_z64": [c_int, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_double, c_double, _P, _P, _P, _P, _P],
    "xeq_load_padded_shard": [c_int, _P, _P, c_int, _P, _P, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_double, c_double, _P, _P, _P, _P, _P],
    "xeq_copy_many": [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), _P],
    "xeq_compare_many": [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), ctypes.c_int32, _P, _P],
    "xeq_tensor_product_path": [c_int, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_int, _P, _P, c_int64, c_double, _P, _P],
    "xeq_tensor_product": [c_int, _P, _P, c_int64, c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int32), _P, ctypes.POINTER(ctypes.c_int32),
                           c_int, _P, c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(c_double), _P, _P],
    "xeq_tensor_product_wgrad": [c_int, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int32), _P,
                                 ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), c_int,
                                 ctypes.POINTER(c_double), c_int, _P, _P],
    "xeq_tensor_product_wgrad_chunks": [c_int64],
    "xeq_scatter_add": [c_int, _P, _P, c_int64, c_int64, _P, c_int64, _P],
    "xeq_message_fwd": [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                        c_int, c_int, c_int, c_double, c_int, _I3, _P, _P, c_int, _P],
    "xeq_message_bwd": [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                        c_int, c_int, c_int, c_double, c_int, _I3, _P, _P, _P, c_int, _P],
    "xeq_message_param_grad_parts": [c_int64],
    "xeq_message_param_grad": [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_double, c_int,
                               _I3, c_int, c_int, _P, _P],
    "xeq_message_param_grad_mc_supported": [c_int, c_int, c_int, c_int, _I3],
    "xeq_param_basis_width": [c_int, c_int],
    "xeq_message_param_grad_mc_parts": [c_int64, c_int, _I3],
    "xeq_param_basis": [_P, c_int64, c_int, c_int, c_int, c_double, _P, _P, _P, _P],
    "xeq_message_param_grad_mc": [c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _I3, c_int, c_int, _P, c_int, _P, _P],
    "xeq_to_bt": [_P, c_int64, _I3, _P, _P],
    "xeq_norm_param_grad_chunks": [c_int64],
    "xeq_norm_param_grad": [_P, _P, _P, _P, c_int64, _P, c_int64, c_int, _I3, c_int, _P, _P],
    "xeq_message_auto_family": [c_int, c_int64, c_int64, c_int, c_int, _I3],
    "xeq_message_wq_edges_per_stream": [c_int64, c_int64],
    "xeq_edge_basis_width": [c_int],
    "xeq_edge_basis": [c_int, _P, c_int64, c_int, c_int, c_int, c_double, _P, _P, _P, _P, _P],
    "xeq_message_fwd_sb": [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _I3, _P, _P, c_int, _P],
    "xeq_message_bwd_sbq": [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _I3, _P, _P, _P, _P, c_int, _P],
    "xeq_message_q_wgrad": [c_int, _P, _P, c_int64, c_int, c_int, _I3, c_int, _P, _P],
    "xeq_message_q_wgrad_chunks": [c_int64],
    "xeq_train_norm": [c_int, c_int, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _I3, c_double, c_double, c_int, _P, _P, _P, _P],
    "xeq_train_uv": [c_int, c_int, c_int64, _P, _P, _P, _I3, c_double, _P, _P, _P],
    "xeq_train_out": [c_int, c_int, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _I3, _P, _P, _P, _P],
    "xeq_zero_rows_from": [_P, c_int64, c_int64, _P, _P],
    "xeq_message_fwd_sb_pair": [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _I3, _P, _P, c_int, _P],
    "xeq_message_bwd_sbq_pair": [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _I3, _P, _P, _P, _P, c_int, _P],
    "xeq_message_bwd_sb": [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _I3, _P, _P, _P,
                           c_int, _P],
    "xeq_message_sb_fits": [c_int64, c_int64, c_int, c_int, _I3],
    "xeq_message_wq_supported": [c_int, c_int, _I3],
    "xeq_message_wq_fits": [c_int64, c_int64, c_int, c_int, _I3],
    "xeq_message_wq_pcap": [c_int64, c_int64],
    "xeq_message_wq_waves": [],
    "xeq_message_wq_record_floats": [],
    "xeq_message_wq_record_floats_for": [c_int],
    "xeq_message_wq_plan_workspace": [c_int64],
    "xeq_message_wq_win_ints": [c_int],
    "xeq_message_wq_plan": [_P, _P, _P, _P, c_int64, c_int64, c_int, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P],
    "xeq_edge_basis_wq": [_P, c_int64, c_int64, _P, _P, c_int, c_int, c_int, c_double, _P, _P, _P, _P, _P],
    "xeq_message_fwd_wq": [c_int64, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _I3,
                           _P, _P, c_int, _P],
    "xeq_message_wq_parts_floats": [c_int64, c_int64, _I3],
    "xeq_message_bwd_wq": [c_int64, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int,
                           _I3, _P, _P, _P, c_int, _P],
    "xeq_message_wq_packed_weight_floats": [c_int, c_int, _I3],
    "xeq_message_wq_pack_weights": [_P, _P, c_int, c_int, _I3, _P, _P],
    "xeq_message_wq_edge_grad": [_P, c_int64, c_int64, _P, _P, _P, _I3, _P, _P, _P],
    "xeq_message_wq_edge_grad_sum": [_P, c_int64, c_int64, _P, _P, _P, _I3, c_int, ctypes.POINTER(c_void_p), _P, _P],
    "xeq_norm_fwd": [c_int, _P, _P, _P, _P, _P, _P, c_int64, c_int, _I3, c_int, _P, c_int64, _P, _P, _P],
    "xeq_norm_bwd": [c_int, _P, _P, _P, _P, _P, c_int64, c_int, _I3, c_int, _P, c_int64, _P, _P, _P, _P, _P, _P],
    "xeq_uv_reduce_fwd": [c_int, _P, c_int64, _I3, c_double, _P, c_int64, c_int, _P, _P],
    "xeq_uv_reduce_bwd": [c_int, _P, _P, _P, c_int64, c_int, c_int64, _I3, c_double, _P, _P, _P, _P],
    "xeq_update_out_fwd": [c_int, _P, _P, _P, _P, _P, c_int64, c_int, _I3, _P, _P, _P],
    "xeq_update_out_bwd": [c_int, _P, _P, _P, _P, _P, c_int64, c_int, _I3, _P, _P, _P, _P],
    "xeq_update_uv_supported": [c_int, c_int, _I3],
    "xeq_update_uv_fwd": [_P, _P, _P, _P, _P, _P, c_int64, c_int, _I3, c_int, _P, _P, _P, c_int, c_double, _P, c_int64, _P, _P, _P, _P],
    "xeq_update_uv_bwd": [_P, _P, _P, c_int64, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int, _I3, c_int, _P, _P, _P, c_double,
                          _P, _P, _P, _P],
    "xeq_mlp2_supported": [c_int, c_int, c_int, c_int],
    "xeq_node_tile_split": [c_int64, c_int, ctypes.POINTER(c_int64)],
    "xeq_mlp_packed_floats": [c_int, c_int],
    "xeq_mlp_pack": [_P, _P, c_int, c_int, c_int, _P, _P],
    "xeq_mlp2_fwd": [_P, c_int64, c_int64, c_int, _P, _P, c_int, _P, _P, c_int64, _P],
    "xeq_mlp2_bwd": [_P, c_int64, c_int64, c_int, _P, _P, _P, c_int, _P, c_int64, _P],
    "xeq_mlp2_and_linear": [c_int, _P, c_int64, c_int64, c_int, _P, _P, c_int, _P, _P, c_int64, _P, c_int64, c_int, _P, c_int, _P, c_int64, _P],
    "xeq_pack_epoch": [],
    "xeq_pack_epoch_bump": [],
    "xeq_rowptr_guard": [_P, c_int64, c_int64, _P, _P, _P],
    "xeq_node_block_supported": [c_int, c_int, _I3],
    "xeq_node_block_fwd_tiles": [c_int],
    "xeq_node_block_rows": [c_int64],
    "xeq_node_block_auto": [c_int64],
    "xeq_small_rows_limit": [],
    "xeq_node_block_set_waves": [c_int],
    "xeq_node_block_pack_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "xeq_node_block_fwd": [c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                           _P, _P, _P, _P, _P],
    "xeq_node_block_linear_test": [_P, c_int64, _P, c_int, c_int, _P, _P, _P],
    "xeq_node_block_bwd_tiles": [c_int, c_int],
    "xeq_node_block_pack_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P],
    "xeq_node_block_bwd": [c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_double, _P, _P, _P, _P, _P,
                           _P, _P, _P],
}
# entry points that return a size, not a status
_RET_I64 = {"xeq_launch_count", "xeq_launch_names", "xeq_message_wq_packed_weight_floats", "xeq_rowptr_from_degrees_max", "xeq_csr_by_key_workspace", "xeq_message_wq_pcap", "xeq_message_wq_plan_workspace", "xeq_message_wq_win_ints",
            "xeq_message_wq_parts_floats", "xeq_mlp_packed_floats", "xeq_exclusive_scan_i32_workspace", "xeq_node_block_fwd_tiles", "xeq_node_block_bwd_tiles", "xeq_node_block_rows", "xeq_pack_epoch", "xeq_tensor_product_wgrad_chunks", "xeq_small_rows_limit"}
EXPORTS = ["xeq_version", "xeq_last_error", *_PROTOS]

_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """dlopen the in-tree HIP library (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m xequinet_amd.csrc.build` "
            "(xequinet_amd has no CPU / pure-PyTorch fallback)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.xeq_version.restype = c_int
    lib.xeq_last_error.restype = c_char_p
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int64 if name in _RET_I64 else c_int
    _lib = lib
    return lib


def pack_epoch() -> int:
    """Epoch of the packed weight copies (include/xeq.h): part of every pack-cache key."""
    return int(load().xeq_pack_epoch())


def bump_pack_epoch() -> None:
    """Parameters changed behind autograd's back (a replayed captured optimizer step): every pack cache misses from here on."""
    load().xeq_pack_epoch_bump()


def launch_count() -> int:
    """Kernel launches libxeq_hip.so has enqueued in this process (include/xeq.h: xeq_launch_count)."""
    return int(load().xeq_launch_count())


def launch_names(first: int) -> list:
    """Entry-point names of the launches numbered ``first`` .. launch_count() - 1, in order (include/xeq.h: xeq_launch_names)."""
    L = load()
    need = int(L.xeq_launch_names(int(first), None, 0))
    if need < 0:
        raise ValueError("launch_names: the range is out of the library's ring of names")
    buf = ctypes.create_string_buffer(need)
    L.xeq_launch_names(int(first), buf, need)
    return [n for n in buf.value.decode().split("\n") if n]


def call(name: str, *args) -> None:
    lib = load()
    status = getattr(lib, name)(*args)
    if status != 0:
        raise RuntimeError(f"{name} failed ({status}): {lib.xeq_last_error().decode()}")


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return XEQ_F32
    if t.dtype == torch.float64:
        return XEQ_F64
    raise TypeError(f"xequinet_amd supports float32/float64 tensors, got {t.dtype}")


def require_hip(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "xequinet_amd ops run on MI355X (HIP) tensors only and have no CPU fallback; "
                f"got a tensor on {t.device}"
            )


def ptr3(ts):
    """Three device pointers as one argument (``const void* const [3]``); None for an absent list."""
    if ts is None:
        return None
    return (c_void_p * 3)(*[None if t is None else t.data_ptr() for t in ts])


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def mul3(mul) -> ctypes.Array:
    return (c_int32 * 3)(*[int(m) for m in mul])
