"""ctypes binding of libpuflow_hip.so (C ABI: include/puflow_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, we raise.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PF_LIB_PATH", os.path.join(_HERE, "libpuflow_hip.so"))   # override: tuning builds only
_lib = None

class PfEcTrain(ctypes.Structure):
    """include/puflow_hip.h: PfEcTrain (one EdgeConv unit of the training step)."""
    _fields_ = [("B", c_int), ("N", c_int), ("K", c_int), ("C", c_int), ("growth", c_int), ("nconv", c_int), ("odim", c_int),
                ("pooling", c_int), ("slope", c_float), ("eps", c_float), ("momentum", c_float),
                ("x", c_void_p), ("idx", c_void_p), ("W", c_void_p * 9), ("bias", c_void_p * 9),
                ("gamma", c_void_p * 8), ("beta", c_void_p * 8), ("run_mean", c_void_p * 8), ("run_var", c_void_p * 8),
                ("Wpq", c_void_p), ("bpq", c_void_p), ("PQ", c_void_p), ("Y", c_void_p), ("aff", c_void_p), ("out", c_void_p),
                ("arg", c_void_p), ("dout", c_void_p), ("dA", c_void_p), ("dPQ", c_void_p), ("coef", c_void_p),
                ("dWpq", c_void_p), ("dx", c_void_p), ("dW", c_void_p * 9), ("dbias", c_void_p * 9),
                ("dgamma", c_void_p * 8), ("dbeta", c_void_p * 8), ("ws", c_void_p), ("ws_floats", c_longlong), ("stat", c_void_p), ("csr_off", c_void_p), ("csr_edge", c_void_p),
                ("flags", c_int), ("sync", c_void_p), ("sync_cb", c_void_p), ("sync_user", c_void_p), ("sync_sums", c_void_p),
                ("ws_dw", c_void_p), ("ws_dw_floats", c_longlong), ("dx_add", c_void_p)]


class PfBnMlpTrain(ctypes.Structure):
    """include/puflow_hip.h: PfBnMlpTrain (a BatchNorm MLP of the interpolation module in the training step)."""
    _fields_ = [("rows", c_int), ("nl", c_int), ("kin0a", c_int), ("kin0b", c_int), ("width", c_int * 3),
                ("slope", c_float), ("eps", c_float), ("momentum", c_float), ("xa", c_void_p), ("xb", c_void_p),
                ("W", c_void_p * 3), ("b", c_void_p * 3), ("gamma", c_void_p * 2), ("beta", c_void_p * 2),
                ("run_mean", c_void_p * 2), ("run_var", c_void_p * 2), ("y", c_void_p * 3), ("aff", c_void_p * 2),
                ("dout", c_void_p), ("d", c_void_p * 2), ("coef", c_void_p * 2), ("dxa", c_void_p), ("dxb", c_void_p),
                ("dW", c_void_p * 3), ("db", c_void_p * 3), ("dgamma", c_void_p * 2), ("dbeta", c_void_p * 2),
                ("ws", c_void_p), ("ws_floats", c_longlong), ("stat", c_void_p),
                ("sync_cb", c_void_p), ("sync_user", c_void_p), ("sync_sums", c_void_p), ("flags", c_int)]


class PfMlpTrain(ctypes.Structure):
    """include/puflow_hip.h: PfMlpTrain (a 2/3-layer point-wise MLP of the training step)."""
    _fields_ = [("rows", c_int), ("nl", c_int), ("td", c_int), ("ldy", c_int), ("cc", c_int), ("cdiv", c_int),
                ("width", c_int * 3), ("slope", c_float * 2), ("y", c_void_p), ("c", c_void_p),
                ("W", c_void_p * 3), ("b", c_void_p * 3), ("h", c_void_p * 2), ("out", c_void_p), ("dout", c_void_p),
                ("dz", c_void_p * 2), ("dy", c_void_p), ("dc", c_void_p), ("dW", c_void_p * 3), ("db", c_void_p * 3),
                ("ws", c_void_p), ("ws_floats", c_longlong), ("chunk", c_int), ("flags", c_int)]


_P8 = c_void_p * 8


class PfFlowChain(ctypes.Structure):
    """include/puflow_hip.h: PfFlowChain (all flow blocks of one direction of the training step)."""
    _fields_ = [("nb", c_int), ("rows", c_int), ("R", c_int), ("inv", c_int), ("td", c_int * 8), ("cc", c_int * 8), ("n_ld", c_float),
                ("x", c_void_p), ("c", _P8), ("s", _P8), ("t", _P8), ("logs", _P8), ("bias", _P8), ("W", _P8),
                ("w0", _P8), ("w2", _P8), ("b2", _P8), ("w4", _P8), ("b4", _P8),
                ("pin", c_void_p), ("mid", c_void_p), ("o", c_void_p), ("h1", c_void_p), ("h2", c_void_p), ("out", c_void_p),
                ("ssum", c_void_p), ("ld", c_void_p), ("logp", c_void_p), ("Bsz", c_int), ("part", c_void_p), ("counter", c_void_p), ("img", c_void_p),
                ("dout", c_void_p), ("dssum", c_void_p), ("dld", c_void_p), ("dlogp", c_void_p), ("dx", c_void_p),
                ("dc", _P8), ("ds", _P8), ("dt", _P8), ("dz1", c_void_p), ("dz2", c_void_p), ("dob", c_void_p),
                ("dlogs", _P8), ("dbias", _P8), ("dW", _P8), ("dw0", _P8), ("dw2", _P8), ("db2", _P8), ("dw4", _P8), ("db4", _P8),
                ("ws", c_void_p), ("ws_floats", c_longlong), ("dev_descs", c_void_p), ("dz1s", c_void_p), ("img_ready", c_int)]


# name -> (restype, argtypes); must list every symbol declared in include/puflow_hip.h
SIGNATURES = {
    "pf_version": (c_int, []),
    "pf_error_string": (c_char_p, [c_int]),
    "pf_knn": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pf_nn1": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pf_edgeconv": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pf_edgeconv_tuned": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p]),
    "pf_edgeconv_pq": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_longlong), c_void_p, c_int, c_int, c_int,
                               c_void_p]),
    "pf_post": (c_int, [c_int, c_void_p, c_void_p, POINTER(c_longlong), c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                        c_void_p]),
    "pf_pq_gemm": (c_int, [c_int, c_void_p, c_void_p, POINTER(c_longlong), c_void_p, c_int, c_void_p]),
    "pf_cond_all": (c_int, [POINTER(c_void_p), c_void_p, POINTER(c_longlong), POINTER(c_void_p), c_void_p, c_void_p, c_int, c_void_p]),
    "pf_cond": (c_int, [c_int, c_void_p, c_void_p, POINTER(c_longlong), c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "pf_flow_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "pf_flow_inv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pf_flow_fwd_logp_ws_floats": (c_longlong, [c_int, c_int]),
    "pf_flow_fwd_logp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "pf_logp": (c_int, [c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_interp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_longlong), c_void_p, c_int, c_int, c_int,
                          c_void_p]),
    "pf_interp_weights": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_longlong), c_void_p, c_int, c_int, c_void_p]),
    "pf_flow_inv_interp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pf_chamfer_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "pf_chamfer_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                               c_int, c_int, c_void_p]),
    "pf_chamfer_bwd_det": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_int, c_int, c_void_p]),
    "pf_emd_forward": (c_int, [c_void_p] * 11 + [c_float, c_int, c_int, c_int, c_void_p]),
    "pf_emd_forward_ex": (c_int, [c_void_p] * 11 + [c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_emd_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pf_gemm_ws_floats": (c_longlong, [c_int, c_int, c_int]),
    "pf_gemm": (c_int, [c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_void_p,
                        c_int, c_int, c_int, c_void_p, c_longlong, c_void_p]),
    "pf_gemm_reduce": (c_int, [c_void_p, c_void_p, c_int, c_int, c_longlong, c_int, c_void_p]),
    "pf_gemm_ex": (c_int, [c_int, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_longlong, c_void_p, c_longlong, c_void_p,
                        c_int, c_int, c_int, c_void_p, c_longlong, c_void_p]),
    "pf_bn_chunks": (c_int, [c_longlong]),
    "pf_bn_lrelu_fwd": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_bn_lrelu_bwd": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_bn_colstat": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_bn_apply_stats": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_float,
                                  c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_bn_bwd_sums": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "pf_bn_bwd_apply": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "pf_colsum": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p]),
    "pf_act_fwd": (c_int, [c_void_p, c_float, c_longlong, c_void_p, c_void_p]),
    "pf_act_bwd": (c_int, [c_void_p, c_void_p, c_float, c_longlong, c_void_p, c_void_p]),
    "pf_edge_feature_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_edge_feature_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_maxpool_k_fwd": (c_int, [c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pf_maxpool_k_bwd": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p]),
    "pf_scatter_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_scatter_rows_det": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p]),
    "pf_group_sum": (c_int, [c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p]),
    "pf_softmax_wsum_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p]),
    "pf_softmax_wsum_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_longlong, c_void_p, c_void_p,
                                    c_void_p]),
    "pf_actnorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p]),
    "pf_actnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_couple_inject_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p]),
    "pf_couple_inject_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "pf_inject_inv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p]),
    "pf_inject_inv_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_couple_add": (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p]),
    "pf_slice_tail": (c_int, [c_void_p, c_int, c_longlong, c_void_p, c_void_p]),
    "pf_batch_sum_fwd": (c_int, [c_void_p, c_int, c_longlong, c_int, c_void_p, c_void_p]),
    "pf_batch_sum_bwd": (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_int, c_void_p, c_void_p]),
    "pf_dist_feature": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pf_fps": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pf_fps_grouped": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pf_fps_scratch_layout": (c_int, [c_int, POINTER(c_longlong), POINTER(c_longlong)]),
    "pf_fps_exchange_probe": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "pf_flow_params_fwd": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "pf_flow_params_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "pf_flow_affine_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p]),
    "pf_flow_affine_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_couple_inject2_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "pf_couple_inject2_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "pf_inject_inv2_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p]),
    "pf_inject_inv2_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_clip_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_float,
                             c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_clip_adam_ptrs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_float,
                             c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_mlp_train_ws_floats": (c_longlong, [c_void_p]),
    "pf_mlp_train_fwd": (c_int, [c_void_p, c_void_p]),
    "pf_mlp_train_bwd": (c_int, [c_void_p, c_void_p]),
    "pf_mlp_train_fwd_batch": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "pf_mlp_train_bwd_batch": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "pf_mlp_train_dw_batch": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "pf_flowchain_ws_floats": (c_longlong, [c_void_p]),
    "pf_flowchain_part_floats": (c_longlong, [c_void_p]),
    "pf_flowchain_img_floats": (c_longlong, [c_void_p]),
    "pf_flowchain_fwd": (c_int, [c_void_p, c_void_p]),
    "pf_flowchain_bwd": (c_int, [c_void_p, c_void_p]),
    "pf_interp_wsum_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_longlong, c_void_p, c_void_p, c_void_p]),
    "pf_interp_wsum_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_void_p, c_void_p,
                                   c_void_p]),
    "pf_interp_wsum_bwd_det": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "pf_emd_init": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p]),
    "pf_pugan_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p]),
    "pf_pugan_loss_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_pugan_grad": (c_int, [c_void_p] * 7 + [c_int] * 3 + [c_float] * 3 + [c_void_p] * 3),
    "pf_bnmlp_train_ws_floats": (c_longlong, [c_void_p]),
    "pf_bnmlp_train_fwd": (c_int, [c_void_p, c_void_p]),
    "pf_bnmlp_train_bwd": (c_int, [c_void_p, c_void_p]),
    "pf_train_set_dw_stream": (c_int, [c_void_p]),
    "pf_fold_wu_fwd": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p] * 5),
    "pf_fold_wu_bwd": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p] * 11),
    "pf_knn_csr": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_knn_csr_pair": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_knn_csr_sort": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "pf_ec_train_ws_floats": (c_longlong, [c_void_p]),
    "pf_ec_train_fwd": (c_int, [c_void_p, c_void_p]),
    "pf_ec_train_fold_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "pf_ec_train_bwd": (c_int, [c_void_p, c_void_p]),
    "pf_cnf_init": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_double,
                            c_void_p, c_double, c_int, c_float, c_float, c_int, c_int, c_void_p, c_void_p]),
    "pf_cnf_context": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p]),
    "pf_cnf_steps": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                             c_float, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "pf_knn_large": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pf_normalize_pc": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pf_format_xyz_bound": (c_longlong, [c_longlong, c_int]),
    "pf_format_xyz": (c_longlong, [c_void_p, c_longlong, c_int, c_void_p, c_longlong]),
    "pf_parse_xyz": (c_longlong, [c_void_p, c_longlong, c_void_p, c_longlong, POINTER(c_int)]),
    "pf_cnf_rhs": (c_int, [c_void_p, c_void_p, POINTER(c_float), c_int, c_float, c_float, c_float, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pf_cnf_step": (c_int, [c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_float, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pf_sum_n": (c_int, [POINTER(c_void_p), c_int, c_void_p, c_longlong, c_void_p]),
    "pf_copy_n": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_longlong), c_int, c_void_p]),
    "pf_lincomb": (c_int, [POINTER(c_void_p), POINTER(c_float), c_int, c_void_p, c_longlong, c_void_p]),
    "pf_scaled_sumsq": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_float), c_int, c_float, c_float,
                                c_float, c_longlong, c_void_p, c_void_p, c_void_p]),
}


class PuflowHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PuflowHipError(
            f"{LIB_PATH} not found: build it with `python -m puflow_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().pf_error_string(rc).decode()
        raise PuflowHipError(f"{what}: {msg} (code {rc})")


def offsets(vals):
    return (c_longlong * len(vals))(*[int(v) for v in vals])
