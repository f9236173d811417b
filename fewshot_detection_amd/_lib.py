"""ctypes binding of libfsdet_hip.so (include/fsdet.h).  Loaded lazily; fails loudly."""
import ctypes as C
import os

# torch ships its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It MUST be in
# the process before libfsdet_hip.so is opened so that both resolve to ONE runtime (one set of
# contexts/streams); loading ours first would bind it to /opt/rocm's copy and every launch on a
# torch pointer/stream then fails with hipErrorNoDevice.
import torch  # noqa: F401  (side effect: loads torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfsdet_hip.so")

_p = C.c_void_p
_i = C.c_int
_ll = C.c_longlong
_f = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol include/fsdet.h declares (tests check)
PROTOTYPES = {
    "fsd_region_loss_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "fsd_region_loss_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _i, _i, _i, _p,
                                     _f, _f, _f, _f, _f, _ll, _i, _i, _i, _p, _p]),
    "fsd_region_build_targets": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _f, _f, _f, _ll, _i, _p]),
    "fsd_region_decode": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _f, _i, _i, _i, _p]),
    "fsd_region_nms": (_i, [_p, _p, _i, _i, _f, _p, _p, _p]),
    "fsd_packed_weight_elems": (_sz, [_i, _i, _i]),
    "fsd_pack_conv_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "fsd_conv_row_tiles": (_i, [_i, _i, _i, _i, _i, _i]),
    "fsd_conv2d_fwd": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fsd_conv3x3_c4_partial_rows": (_i, [_i, _i, _i]),
    "fsd_conv3x3_c4_fwd": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _i, _i, _i, _i, _i, _p]),
    "fsd_wino_packed_weight_elems": (_sz, [_i, _i, _i]),
    "fsd_wino_pack_weight": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "fsd_wino_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "fsd_wino_partial_rows": (_i, [_i, _i, _i, _i]),
    "fsd_wino_v_elems": (_sz, [_i, _i, _i, _i, _i]),
    "fsd_wino_conv3x3_fwd": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _p, _sz, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fsd_wino_fwd_plan": (_i, [_i, _i, _i, _i, _i, _i, _p]),
    "fsd_wino_wgrad_plan": (_i, [_i, _i, _i, _i, _i, _i, _p]),
    "fsd_conv3x3_wgrad_c4_bnfused_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "fsd_conv3x3_wgrad_c4_bnfused": (_i, [_p, _ll, _p, _ll, _p, _p, _p, _p, _ll, _p, _p, _sz, _i, _i, _i, _i, _i, _p]),
    "fsd_conv2d_fwd_act": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "fsd_conv2d_fwd_ex": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _f, _p]),
    "fsd_wino_conv3x3_fwd_ex": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _p, _sz, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _f, _p]),
    "fsd_conv2d_wgrad_ex": (_i, [_p, _ll, _p, _ll, _p, _p, _sz, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p]),
    "fsd_wino_conv3x3_fwd_act": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _p, _sz, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "fsd_conv2d_fwd_act_h": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "fsd_first_layer_bwd_rows": (_i, [_i, _i, _i]),
    "fsd_first_layer_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "fsd_first_layer_bwd_accum": (_i, [_p, _ll, _p, _ll, _p, _p, _p, _p, _f, _p, _ll, _p, _sz, _p, _i, _i, _i, _i, _i, _p]),
    "fsd_first_layer_bwd_accum_h": (_i, [_p, _ll, _p, _ll, _p, _p, _p, _p, _f, _p, _ll, _p, _sz, _p, _i, _i, _i, _i, _i, _p]),
    "fsd_first_layer_bwd_fold": (_i, [_p, _sz, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fsd_wino_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "fsd_wino_conv3x3_wgrad": (_i, [_p, _ll, _p, _ll, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _i, _i, _p]),
    "fsd_wino_dy_bn_transform": (_i, [_p, _ll, _p, _ll, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fsd_wino_dy_bn_transform_g": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _p, _f, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fsd_bn_bwd_apply_g": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _p, _f, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fsd_bn_bwd_apply_g_h": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _p, _f, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fsd_wino_grad_transforms": (_i, [_p, _ll, _p, _ll, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fsd_packed_weight_elems_bf16": (_sz, [_i, _i, _i]),
    "fsd_pack_conv_weight_bf16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "fsd_pack_conv_weight_bf16_pair": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fsd_bn_finalize_workspace_bytes": (_sz, [_i]),
    "fsd_bn_finalize": (_i, [_p, _i, _ll, _i, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p]),
    "fsd_bn_act_pool_fwd": (_i, [_p, _ll, _p, _p, _f, _i, _p, _ll, _i, _i, _i, _i, _p]),
    "fsd_transpose_batched": (_i, [_p, _ll, _ll, _p, _ll, _ll, _i, _i, _i, _p]),
    "fsd_nchw_to_nhwc4": (_i, [_p, _p, _i, _i, _ll, _p]),
    "fsd_fill": (_i, [_p, _f, _ll, _p]),
    "fsd_reorg_fwd": (_i, [_p, _ll, _p, _ll, _i, _i, _i, _i, _i, _p]),
    "fsd_global_maxpool_fwd": (_i, [_p, _ll, _p, _p, _i, _i, _i, _i, _p]),
    "fsd_upload_words": (_i, [_p, _p, _ll, _p]),
    "fsd_global_avgpool_fwd": (_i, [_p, _i, _ll, _p, _i, _i, _i, _i, _p]),
    "fsd_global_avgpool_bwd": (_i, [_p, _p, _i, _ll, _i, _i, _i, _i, _p]),
    "fsd_dynamic_conv_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fsd_fold_reweight_head": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fsd_conv2d_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "fsd_conv2d_wgrad": (_i, [_p, _ll, _p, _ll, _p, _p, _sz, _i, _i, _i, _i, _i, _i, _p]),
    "fsd_act_bwd_rows": (_i, [_ll]),
    "fsd_bn_act_pool_bwd_rows": (_i, [_i, _i, _i, _i]),
    "fsd_reduce_workspace_bytes": (_sz, [_i]),
    "fsd_bn_act_pool_bwd": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _p, _p, _p, _f, _i, _p, _p, _i, _i, _i, _i, _p]),
    "fsd_bn_bwd_finalize": (_i, [_p, _i, _ll, _i, _p, _p, _p, _p, _p, _p]),
    "fsd_bn_bwd_apply": (_i, [_p, _p, _ll, _p, _p, _p, _ll, _i, _p]),
    "fsd_colsum_partials": (_i, [_p, _ll, _p, _ll, _i, _p]),
    "fsd_reorg_bwd": (_i, [_p, _ll, _p, _ll, _i, _i, _i, _i, _i, _p]),
    "fsd_global_maxpool_bwd": (_i, [_p, _p, _p, _ll, _i, _i, _i, _i, _p]),
    "fsd_add_inplace": (_i, [_p, _ll, _p, _ll, _ll, _i, _p]),
    "fsd_head_unfold_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fsd_sgd_step": (_i, [_p, _p, _p, _f, _f, _f, _i, _ll, _p]),
    "fsd_sgd_multi_blocks": (_ll, [_ll, _i, _i, _i]),
    "fsd_sgd_step_multi": (_i, [_p, _p, _p, _p, _i, _ll, _ll, _f, _f, _f, _i, _p]),
    "fsd_conv_row_tiles_h": (_i, [_ll]),
    "fsd_conv2d_h_partial_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "fsd_conv2d_h_partial_rows_at": (_i, [_i, _i, _i, _i, _i, _i, _p, _ll, _p, _ll]),
    "fsd_conv2d_h_plan": (_i, [_ll, _i, _i, _i, _i, _i]),
    "fsd_conv2d_fwd_h": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fsd_conv2d_wgrad_h_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "fsd_conv2d_wgrad_h_plan": (_i, [_ll, _i, _i, _i]),
    "fsd_conv2d_wgrad_h": (_i, [_p, _ll, _p, _ll, _p, _p, _sz, _i, _i, _i, _i, _i, _i, _p]),
    "fsd_conv3x3_c4_fwd_h": (_i, [_p, _ll, _p, _p, _p, _ll, _p, _i, _i, _i, _i, _i, _p]),
    "fsd_conv3x3_wgrad_c4_bnfused_h": (_i, [_p, _ll, _p, _ll, _p, _p, _p, _p, _ll, _p, _p, _sz, _i, _i, _i, _i, _i, _p]),
    "fsd_bn_act_pool_fwd_h": (_i, [_p, _ll, _p, _p, _f, _i, _p, _ll, _i, _i, _i, _i, _p]),
    "fsd_transpose_batched_h": (_i, [_p, _i, _ll, _ll, _p, _i, _ll, _ll, _i, _i, _i, _p]),
    "fsd_reorg_fwd_h": (_i, [_p, _ll, _p, _ll, _i, _i, _i, _i, _i, _p]),
    "fsd_global_maxpool_fwd_h": (_i, [_p, _ll, _p, _p, _i, _i, _i, _i, _p]),
    "fsd_bn_act_pool_bwd_h": (_i, [_p, _ll, _p, _ll, _p, _ll, _p, _p, _p, _p, _f, _i, _p, _p, _i, _i, _i, _i, _p]),
    "fsd_bn_bwd_apply_h": (_i, [_p, _p, _ll, _p, _p, _p, _ll, _i, _p]),
    "fsd_colsum_partials_h": (_i, [_p, _ll, _p, _ll, _i, _p]),
    "fsd_reorg_bwd_h": (_i, [_p, _ll, _p, _ll, _i, _i, _i, _i, _i, _p]),
    "fsd_global_maxpool_bwd_h": (_i, [_p, _p, _p, _ll, _i, _i, _i, _i, _p]),
    "fsd_add_inplace_h": (_i, [_p, _ll, _p, _ll, _ll, _i, _p]),
    "fsd_augment_batch": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fsd_profile_enable": (None, [_i]),
    "fsd_profile_num_classes": (_i, []),
    "fsd_profile_collect": (_i, [_p, _p, _p, _i]),
    "fsd_launch_count": (_ll, [_i]),
    "fsd_clock_probe": (_i, [_p, _i, _p, _p]),
    "fsd_f32_gemm_mode": (_i, [_i]),
    "fsd_version": (C.c_char_p, []),
}

_lib = None


class FsdetLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises FsdetLibraryError (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise FsdetLibraryError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C fewshot_detection_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        try:
            hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
            if os.path.isfile(hip_rt):
                C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise FsdetLibraryError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise FsdetLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


_ERR = {-1: "FSD_ERR_ARG (bad argument)", -2: "FSD_ERR_UNSUPPORTED", -3: "FSD_ERR_WORKSPACE (too small)"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, "hipError_t %d" % rc)))
