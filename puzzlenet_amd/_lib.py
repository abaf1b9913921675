"""ctypes binding of libpzn.so — the C ABI declared in include/pzn.h.

The library is built ahead of time by ``python -m puzzlenet_amd.build`` (hipcc,
gfx950) and lives next to this file.  There is NO fallback: if the shared
object is missing, or a call returns a non-zero status, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpzn.so")

_c_f = ctypes.c_void_p      # device pointers travel as raw addresses
_c_i = ctypes.c_int
_c_sz = ctypes.c_size_t
_c_fl = ctypes.c_float
_c_ll = ctypes.c_longlong
_PP = ctypes.POINTER(ctypes.c_void_p)   # array of device pointers (one per problem)

# name -> (restype, argtypes); mirrors include/pzn.h one to one.
SIGNATURES = {
    "pzn_version": (_c_i, []),
    "pzn_strerror": (ctypes.c_char_p, [_c_i]),
    "pzn_device_check": (_c_i, []),
    "pzn_ktimer_enable": (_c_i, [_c_i]),
    "pzn_ktimer_collect": (_c_i, []),
    "pzn_ktimer_row": (_c_i, [_c_i, ctypes.c_char_p, _c_i, ctypes.POINTER(_c_i), ctypes.POINTER(ctypes.c_double)]),
    "pzn_square_distance_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_fps_f32": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_fps_background_f32": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_i, _c_f]),
    "pzn_knn_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_ball_query_f32": (_c_i, [_c_fl, _c_i, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_gather_fwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_gather_bwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_group_fwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_group_bwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f]),
    "pzn_emd_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "pzn_emd_walk_counter_offset": (_c_sz, [_c_i, _c_i, _c_i]),
    "pzn_emd_walk_counter_count": (_c_i, []),
    "pzn_emd_approxmatch_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_emd_matchcost_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_emd_matchcost_grad_f32": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_emd_workspace_bytes_f64": (_c_sz, [_c_i, _c_i, _c_i]),
    "pzn_emd_approxmatch_f64": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_emd_matchcost_f64": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_emd_matchcost_grad_f64": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_emd_fused_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f]),
    "pzn_emd_fused_small_multi_f32": (_c_i, [_c_i] + [_c_f] * 8 + [_c_f]),
    "pzn_chamfer_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "pzn_chamfer_fwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    "pzn_linear_fwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_linear_maxpool_fwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_linear_dgrad_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_linear_wgrad_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_i, _c_f]),
    "pzn_linear_maxpts_workspace_bytes": (_c_sz, [_c_i, _c_i]),
    "pzn_linear_maxpts_dgrad_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_linear_maxpts_wgrad_f32": (_c_i, [_c_f, _c_f, ctypes.POINTER(ctypes.c_void_p), _c_i, _c_i, _c_i, _c_i, _c_i, _c_f,
                                    _c_f, _c_f]),
    "pzn_linear_slice_fwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_linear_slice_dgrad_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_linear_slice_wgrad_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_i, _c_f, _c_f]),
    "pzn_linear_maxpool_dgrad_f32": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_linear_maxpool_wgrad_f32": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_i, _c_f]),
    "pzn_gemm_set_precision": (_c_i, [_c_i]),
    "pzn_gemm_get_precision": (_c_i, []),
    "pzn_attn_set_precision": (_c_i, [_c_i]),
    "pzn_attn_get_precision": (_c_i, []),
    "pzn_attn_fused_supported": (_c_i, [_c_i, _c_i, _c_i]),
    "pzn_attn_fused_weight_bytes": (_c_sz, []),
    "pzn_attn_fused_qk_image_bytes": (_c_sz, [_c_i]),
    "pzn_attn_fused_v_image_bytes": (_c_sz, [_c_i]),
    "pzn_attn_fused_prep_weights": (_c_i, [_c_f] * 6),
    "pzn_attn_fused_prep_weights_n": (_c_i, [_c_i] + [_PP] * 5 + [_c_f]),
    "pzn_attn_fused_proj": (_c_i, [_c_i] + [_PP] * 5 + [_c_i] + [_PP] * 3 + [_c_f]),
    "pzn_attn_fused_fwd": (_c_i, [_c_i] + [_PP] * 6 + [_c_i] + [_PP] * 5 + [_c_i, _c_fl, _c_f]),
    "pzn_attn_fused_bwd_q": (_c_i, [_c_i, _PP, _c_i, _PP, _c_i] + [_PP] * 5 + [_c_i] + [_PP] * 6 + [_c_f]),
    "pzn_attn_fused_bwd_k": (_c_i, [_c_i] + [_PP] * 9 + [_c_i] + [_PP] * 3 + [_c_f]),
    "pzn_attn_chain_saved_bytes": (_c_sz, [_c_i]),
    "pzn_attn_chain_scratch_bytes": (_c_sz, [_c_i]),
    "pzn_attn_chain_fwd_f32": (_c_i, [_c_f, _PP, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f]),
    "pzn_attn_chain_bwd_f32": (_c_i, [_c_f, _PP, _c_f, _c_f, _c_f, _c_i, _PP, _c_i, _c_f, _c_f, _c_f]),
    "pzn_attn_fused_wgrads": (_c_i, [_c_f] * 6 + [_c_i] * 3 + [_c_f] * 8 + [_c_i, _c_f]),
    "pzn_sharedmlp_max_fwd_f32": (_c_i, [_c_f] * 5 + [_c_i] * 4 + [_c_f] * 4),
    "pzn_sharedmlp_max_bwd_f32": (_c_i, [_c_f] * 7 + [_c_i] * 4 + [_c_f] * 6 + [_c_i, _c_f]),
    "pzn_knn_group_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f]),
    "pzn_attn_block_fwd_f32": (_c_i, [_c_f] * 9 + [_c_i] * 4 + [_c_f] * 7 + [_c_f]),
    "pzn_attn_block_bwd_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i, _c_i]),
    "pzn_attn_block_bwd_f32": (_c_i, [_c_f] * 13 + [_c_i] * 4 + [_c_f] * 10 + [_c_i, _c_f]),
    "pzn_maxpool_points_fwd_f32": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_maxpool_points_bwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_se3_exp_fwd_f32": (_c_i, [_c_f, _c_i, _c_f, _c_f]),
    "pzn_se3_exp_bwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_f, _c_f]),
    "pzn_se3_transform_fwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_f, _c_f]),
    "pzn_se3_transform_bwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_comp_fwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_f, _c_f]),
    "pzn_comp_bwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_f, _c_f]),
    "pzn_boundary_ce_fwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_boundary_ce_bwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_topk_rows_f32": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_avg4_f32": (_c_i, [_c_f, _c_f, _c_f, _c_f, ctypes.c_size_t, _c_f, _c_f]),
    "pzn_colmean_workspace_bytes": (_c_sz, [_c_i, _c_i]),
    "pzn_colmean_argmax_f32": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f]),
    "pzn_adam_step_f32": (_c_i, [_c_f, _c_f, _c_f, _c_f, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, _c_i, _c_f]),
    "pzn_bn_points_relu_fwd_f32": (_c_i, [_c_f] * 5 + [_c_i, _c_fl, _c_fl, _c_i, _c_i, _c_i] + [_c_f] * 4),
    "pzn_stem_fwd_f32": (_c_i, [_c_f] * 7 + [_c_fl, _c_fl] + [_c_f] * 6 + [_c_fl, _c_fl, _c_i, _c_i, _c_i] + [_c_f] * 6),
    "pzn_stem_bwd_workspace_bytes": (_c_sz, [_c_i]),
    "pzn_stem_bwd_f32": (_c_i, [_c_f] * 15 + [_c_i, _c_i, _c_i] + [_c_f] * 10),
    "pzn_bn_points_relu_bwd_f32": (_c_i, [_c_f] * 6 + [_c_i] * 4 + [_c_f] * 4),
    "pzn_sa_prep_f32": (_c_i, [_c_f] * 4 + [_c_i] * 5 + [_c_f] * 3),
    "pzn_sa_level_fwd_f32": (_c_i, [_c_f] * 5 + [_c_i] * 5 + [_c_f] * 3),
    "pzn_sa_level_fwd_ws_f32": (_c_i, [_c_f] * 5 + [_c_i] * 5 + [_c_f] * 4),
    "pzn_sa_level_fwd_workspace_bytes": (_c_sz, [_c_i, _c_i]),
    "pzn_sa_level_prep_weights_f32": (_c_i, [_c_f, _c_i, _c_i, _c_f, _c_f]),
    "pzn_sa_level_fwd_packed_f32": (_c_i, [_c_f] * 4 + [_c_i] * 5 + [_c_f] * 4),
    "pzn_outproj_maxpts_workspace_bytes": (_c_sz, [_c_i] * 4),
    "pzn_cloud_bias_relu_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f]),
    "pzn_point_mlp3_supported": (_c_i, [_c_i] * 4),
    "pzn_point_mlp3_bwd_workspace_bytes": (_c_sz, [_c_ll, _c_i, _c_i, _c_i, _c_i]),
    "pzn_point_mlp3_bwd_f32": (_c_i, [_c_f] * 4 + [_c_ll, _c_i, _c_f, _c_i, _c_i, _c_f, _c_f, _c_i, _c_i] + [_c_f] * 7 + [_c_i, _c_f, _c_f]),
    "pzn_point_mlp3_fwd_f32": (_c_i, [_c_f, _c_ll, _c_i, _c_f, _c_i, _c_f, _c_i, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f]),
    "pzn_cloud_gated_colsum_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_outproj_maxpts_fwd_f32": (_c_i, [_PP, _c_i, _c_f, _c_f] + [_c_i] * 4 + [_c_f] * 5),
    "pzn_sa_level_chain_saved_bytes": (_c_sz, [_c_i] * 6),
    "pzn_sa_level_chain_scratch_bytes": (_c_sz, [_c_i] * 5),
    "pzn_sa_level_chain_fwd_f32": (_c_i, [_c_f] * 8 + [_c_i] * 6 + [_c_f] * 4),
    "pzn_sa_level_chain_bwd_f32": (_c_i, [_c_f] * 10 + [_c_i] * 6 + [_c_f] * 5 + [_c_i, _c_f, _c_f]),
    "pzn_sa_level_bwd_pt_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i]),
    "pzn_sa_level_bwd_pt_f32": (_c_i, [_c_f] * 12 + [_c_i] * 6 + [_c_f] * 5 + [_c_i, _c_f, _c_f]),
    "pzn_knn_inverse_lists": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f]),
    "pzn_attn_fwd_f32": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "pzn_attn_bwd_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i, _c_i]),
    "pzn_attn_bwd_f32": (_c_i, [_c_f] * 6 + [_c_i] * 4 + [_c_f] * 5),
    "pzn_cut_compact_f32": (_c_i, [_c_f] * 4 + [_c_i] * 5 + [_c_f] * 6),
    "pzn_pick_mask_f32": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "pzn_chamfer_bwd_f32": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f]),
}

_lib = None


class PznError(RuntimeError):
    pass


class PznUnsupported(PznError):
    """status PZN_EUNSUPPORTED (-3): the entry point does not take this shape / alignment; composed entry points
    document the alternative path."""


def load():
    """Load libpzn.so (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PznError(
                f"{LIB_PATH} is missing: build it with `python -m puzzlenet_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU or eager fallback.")
        # torch ships its own libamdhip64.so.7; device pointers and streams handed to the
        # C ABI belong to THAT runtime instance, so it must be the one libpzn.so binds to.
        # Same SONAME => whichever copy is loaded first wins; load torch's first.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status, what):
    if status != 0:
        msg = load().pzn_strerror(status).decode()
        raise (PznUnsupported if status == -3 else PznError)(f"{what} failed: {msg} (status {status})")


_FN = {}


def call(name, *args):
    """Invoke an int-status entry point and raise on error."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    status = fn(*args)
    if status != 0:
        check(status, name)
