"""ctypes binding of libsdfr_hip.so (the C ABI declared in include/sdfr.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails the product raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDFR_LIB") or os.path.join(_HERE, "lib", "libsdfr_hip.so")     # SDFR_LIB: A/B builds (tools/ab_build.sh)

ABI_VERSION = 400          # include/sdfr.h SDFR_VERSION
_lib = None

# name -> (restype, argtypes); mirrors include/sdfr.h one to one
_PROTOS = {
    "sdfr_version": (c_int, []),
    "sdfr_build_flags": (c_int, []),
    "sdfr_last_error": (c_char_p, []),
    "sdfr_debug_set_trace": (c_int, [c_void_p]),
    "sdfr_mlp_forward_counted": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    "sdfr_trace_setup": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sdfr_trace_setup2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_cone": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_int,
                                c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "sdfr_trace_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_int64, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_march": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_void_p, c_int,
                                 c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_pool": (c_int, []),
    "sdfr_trace_hits": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_composite": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_backward_ws_floats": (c_int64, [c_int, c_int, c_int]),
    "sdfr_trace_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_setup_r": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_cone_r": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_float, c_float, c_int, c_int, c_int,
                                  c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "sdfr_trace_march_r": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_int,
                                   c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_hits_r": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_composite_r": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_points_r": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_refine_backward_r": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_points": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_trace_refine_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_ws_words_r": (c_int64, [c_int, c_int, c_int]),
    "sdfr_surfels_forward_r": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_forward_r": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                     c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_backward_r": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_float,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "sdfr_loss_2d_r": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "sdfr_prefilter_audit_select": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sdfr_prefilter_audit_check": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p]),
    "sdfr_scale_net": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_decoder_create": (c_int, [POINTER(c_void_p), c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                    POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int]),
    "sdfr_decoder_destroy": (c_int, [c_void_p]),
    "sdfr_decoder_macs": (c_int64, [c_void_p]),
    "sdfr_mlp_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sdfr_mlp_forward_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sdfr_mlp_forward_f16_counted": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_mlp_forward_split": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sdfr_mlp_forward_split_counted": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sdfr_decoder_mask_words": (c_int64, [c_void_p, c_int64]),
    "sdfr_mlp_jacobian": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int, c_void_p]),
    "sdfr_band_select": (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_band_select_margin": (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_prefilter_guard": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_prefilter_plan": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p]),
    "sdfr_mlp_forward_f16_skip": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "sdfr_band_select_skip": (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_prefilter_guard2": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_candidate_rows": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_mlp_forward_f16_ragged": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sdfr_mlp_forward_skip": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "sdfr_mlp_forward_ragged": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sdfr_candidate_band_map": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_surface_project": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_surface_project_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "sdfr_sdf_input_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_project_dcm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_surfels_forward": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                     c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_project_dcm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_forward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_backward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_ws_words": (c_int64, [c_int, c_int, c_int, c_int]),
    "sdfr_splat_weights": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                   c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_weights_backward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                            c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_forward_clamp": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_int, c_void_p, c_int, c_int, c_float, c_float, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_weights_clamp": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                         c_int, c_float, c_float, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_weights_backward_clamp": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                                  c_int, c_int, c_float, c_float, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_void_p]),
    "sdfr_params_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_surface_latent_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_params_backward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "sdfr_pose_latent_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    "sdfr_gather_rows3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sdfr_scatter_values": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "sdfr_gather_rows": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sdfr_scatter_add_rows3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sdfr_loss_3d": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_loss_2d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p]),
    "sdfr_solver_step": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                 c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    # r06: fused entry points
    "sdfr_params_plan": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sdfr_band_select_ex": (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_void_p]),
    "sdfr_mlp_forward_candidates": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                            c_void_p]),
    "sdfr_candidate_band": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "sdfr_losses_fused": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_splat_backward_x": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                      c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sdfr_pose_latent_solver": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                        c_float, c_void_p, c_void_p, c_void_p]),
}

EXPORTS = tuple(_PROTOS)


class SdfrError(RuntimeError):
    pass


class Extents(ctypes.Structure):
    """sdfr_extents of include/sdfr.h (read on the host at the call; `wh` is a device pointer)"""
    _fields_ = [("wh", c_void_p), ("pix_stride", c_int), ("cone_cap", c_int)]


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise SdfrError("libsdfr_hip.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or sdflabel_amd/csrc/build.sh -- there is no CPU fallback" % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        # the version first: a stale library lacks newer symbols, and the message should say "rebuild", not AttributeError
        try:
            h.sdfr_version.restype = c_int
            h.sdfr_version.argtypes = []
            ver = int(h.sdfr_version())
        except AttributeError:
            ver = -1
        if ver != ABI_VERSION:
            raise SdfrError("%s reports ABI version %d, this binding is written against %d (include/sdfr.h SDFR_VERSION): rebuild the library"
                            % (LIB_PATH, ver, ABI_VERSION))
        for name, (res, args) in _PROTOS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.sdfr_version() != ABI_VERSION:
            raise SdfrError("%s reports ABI version %d, this binding is written against %d (include/sdfr.h SDFR_VERSION): rebuild the library"
                            % (LIB_PATH, h.sdfr_version(), ABI_VERSION))
        flags = h.sdfr_build_flags()
        if flags and os.environ.get("SDFR_ALLOW_AB") != "1":
            raise SdfrError("%s is an experiment build (sdfr_build_flags() = %d%s): the product refuses it; set SDFR_ALLOW_AB=1 for A/B timing runs"
                            % (LIB_PATH, flags, ", results wrong by construction" if flags & 2 else ""))
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise SdfrError("%s failed (%d): %s" % (what, rc, lib().sdfr_last_error().decode()))


def ptr(t):
    """device (or host) pointer of a torch tensor / None (a plain int: ctypes converts it for the c_void_p parameters)"""
    return None if t is None else t.data_ptr()


def stream_ptr():
    """the current stream of the CURRENT device: call it inside `with guard(tensor):` so that device is the tensors' one"""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


TRAFFIC_SOURCES = {          # the csrc files that define the kernels a profiles/traffic_*.json file was measured on (bench.py, tools/summarize_profile.py)
    "traffic_mlp_forward.json": ("mlp_kernel.h", "mlp.hip", "mlp_fwd32.hip"),
    "traffic_splat.json": ("splat.hip", "splat_bbox.h"),
    "traffic_sphere_step.json": ("trace.hip",),
}


def source_sha16(names):
    """first 16 hex digits of the SHA-256 over the named csrc files: stored in profiles/traffic_*.json when the PMC passes are summarised, compared
    by bench.py before it quotes the stored figure (a figure measured on other kernel sources is reported as traffic: null)"""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def stored_traffic(root, name):
    """profiles/<name> as a dict if it exists AND was measured on the kernel sources of this tree, else None (old files without a hash: None)"""
    import json
    path = os.path.join(root, "profiles", name)
    if not os.path.isfile(path):
        return None
    d = json.load(open(path))
    return d if d.get("source_sha16") == source_sha16(TRAFFIC_SOURCES[name]) else None


def guard(t):
    """Context manager making the device of tensor (or torch.device) `t` current, as every kernel launch needs: the launch stream
    (stream_ptr) and any allocation inside the library then belong to the device that holds the data, whatever the caller's current
    device is (the reference works on any device the tensors live on).  Nothing to do when that device is current already."""
    import torch
    dev = t.device if hasattr(t, "device") else torch.device(t)
    if dev.type == "cuda" and (dev.index is None or dev.index == torch._C._cuda_getDevice()):
        return _NO_GUARD
    return torch.cuda.device(dev)


def traced(name):
    """decorator: run the function inside torch.profiler.record_function('sdfr::<name>') while a profiler is active (bench.py's dropin_api
    section uses the ranges to tell the library's launches from the caller's own torch ops); a plain call otherwise"""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            import torch
            if not torch.autograd._profiler_enabled():
                return fn(*a, **k)
            with torch.profiler.record_function("sdfr::" + name):
                return fn(*a, **k)
        return wrapped
    return deco


def splat_ws(B, cap, W, H, device):
    """workspace of sdfr_splat_forward / sdfr_surfels_forward: screen boxes + per-tile surfel lists (int32)"""
    import torch
    return torch.empty((int(lib().sdfr_splat_ws_words(int(B), max(int(cap), 1), int(W), int(H))),), dtype=torch.int32, device=device)


def require_gpu_float(*tensors):
    """Boundary check: GPU tensors, float32 or float16 (half tensors -- the reference's default `precision` -- are widened to float32
    at the boundary and the results narrowed back; all kernels compute in float32 except the optional float16 decoder MFMAs)."""
    import torch
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise SdfrError("sdflabel_amd runs on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)
        if t.dtype not in (torch.float32, torch.float16):
            raise SdfrError("sdflabel_amd accepts float32 or float16 tensors (got %s)" % t.dtype)


require_gpu_f32 = require_gpu_float
