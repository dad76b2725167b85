"""ctypes binding of liblighthand_hip.so (the C ABI declared in include/lighthand_hip.h).

The library is the only compute backend of this package: if it cannot be loaded the
import of any compute module fails loudly -- there is no CPU / eager fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LH_LIB_PATH") or os.path.join(_HERE, "liblighthand_hip.so")   # override: kernel experiments only

LH_F32, LH_BF16, LH_F16 = 0, 1, 2


class LightHandError(RuntimeError):
    pass


class IgemmDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("hi", C.c_int), ("wi", C.c_int), ("in_pix_stride", C.c_int),
                ("k_run", C.c_int), ("ho", C.c_int), ("wo", C.c_int), ("sh", C.c_int), ("sw", C.c_int),
                ("cout", C.c_int), ("OH", C.c_int), ("OW", C.c_int), ("osh", C.c_int), ("osw", C.c_int),
                ("ooh", C.c_int), ("oow", C.c_int), ("out_pix_stride", C.c_int), ("ntaps", C.c_int),
                ("relu", C.c_int), ("dh", C.c_byte * 64), ("dw", C.c_byte * 64), ("cfg", C.c_int * 8)]


class BnFinalizeCall(C.Structure):     # lh_bn_finalize_multi / lh_fuse_desc.fin
    _fields_ = [("stats", C.c_void_p), ("rows", C.c_int), ("count", C.c_int), ("c", C.c_int), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p), ("momentum", C.c_float),
                ("eps", C.c_float), ("scale", C.c_void_p), ("shift", C.c_void_p), ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p)]


class FuseDesc(C.Structure):
    _fields_ = [("x", C.c_void_p * 4), ("scale", C.c_void_p * 4), ("shift", C.c_void_p * 4),
                ("log2up", C.c_int * 4), ("nterms", C.c_int), ("relu", C.c_int), ("relu_mask", C.c_void_p),
                ("fin", C.POINTER(BnFinalizeCall) * 4), ("l2_touch", C.c_void_p), ("l2_touch_bytes", C.c_size_t)]


class FuseBwdDesc(C.Structure):
    _fields_ = [("dout", C.c_void_p), ("out", C.c_void_p), ("x", C.c_void_p * 4), ("scale", C.c_void_p * 4),
                ("shift", C.c_void_p * 4), ("save_mean", C.c_void_p * 4), ("save_invstd", C.c_void_p * 4), ("dx", C.c_void_p * 4),
                ("dgamma", C.c_void_p * 4), ("dbeta", C.c_void_p * 4), ("log2up", C.c_int * 4),
                ("accumulate", C.c_int * 4), ("nterms", C.c_int), ("relu", C.c_int),
                ("relu_mask", C.c_void_p), ("strips_cap", C.c_int), ("pre_partial", C.c_void_p), ("pre_rows", C.c_int),
                ("l2_touch", C.c_void_p), ("l2_touch_bytes", C.c_size_t), ("pre_partial2", C.c_void_p)]


class BnBwdGate(C.Structure):          # lh_igemm_gated
    _fields_ = [("x", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("partial", C.c_void_p), ("mask", C.c_void_p), ("x2", C.c_void_p), ("mean2", C.c_void_p), ("invstd2", C.c_void_p), ("partial2", C.c_void_p)]


class IgemmCall(C.Structure):          # one entry of lh_igemm_multi = the arguments of lh_igemm
    _fields_ = [("d", C.POINTER(IgemmDesc)), ("in_", C.c_void_p), ("wpack", C.c_void_p), ("out", C.c_void_p), ("addend", C.c_void_p),
                ("addend_mask", C.c_void_p), ("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("stats", C.c_void_p)]


class WgradCall(C.Structure):          # lh_wgrad_fused_multi
    _fields_ = [("d", C.POINTER(IgemmDesc)), ("rows", C.c_int), ("x", C.c_void_p), ("dy", C.c_void_p), ("dy_pix_stride", C.c_int),
                ("n_out", C.c_int), ("n_in", C.c_int), ("workspace", C.c_void_p), ("grad", C.c_void_p), ("so", C.c_long), ("si", C.c_long),
                ("sr", C.c_long), ("ss", C.c_long), ("taps_rs", C.POINTER(C.c_int)), ("accumulate", C.c_int)]


class WgradTableInfo(C.Structure):     # lh_wgrad_table_build / lh_wgrad_table_run
    _fields_ = [("bo", C.c_int), ("bi", C.c_int), ("kps", C.c_int), ("depth", C.c_int), ("n_problems", C.c_int), ("n_fold", C.c_int),
                ("n_items", C.c_int), ("n_fold_items", C.c_int), ("fold_lds", C.c_int), ("target_stages", C.c_int), ("nsplit_max", C.c_int),
                ("run_parts", C.c_int), ("off_items", C.c_size_t), ("off_fold_args", C.c_size_t), ("off_fold_items", C.c_size_t),
                ("table_bytes", C.c_size_t), ("workspace_bytes", C.c_size_t)]


class FuseFwdCall(C.Structure):        # lh_fuse_fwd_multi
    _fields_ = [("d", C.POINTER(FuseDesc)), ("out", C.c_void_p), ("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("c", C.c_int)]


class FuseBwdCall(C.Structure):        # lh_fuse_bwd_multi
    _fields_ = [("d", C.POINTER(FuseBwdDesc)), ("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("c", C.c_int), ("workspace", C.c_void_p)]


class BottleneckDesc(C.Structure):   # lh_bottleneck_infer
    _fields_ = [("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("cin", C.c_int), ("mid", C.c_int), ("cout", C.c_int)]


class Head(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_row_bytes", C.c_size_t), ("bias", C.c_void_p), ("out", C.c_void_p), ("n_out", C.c_int)]


class PackItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("n_out", C.c_int), ("n_in", C.c_int), ("ntaps", C.c_int),
                ("pad_", C.c_int), ("so", C.c_long), ("si", C.c_long), ("sr", C.c_long), ("ss", C.c_long),
                ("r", C.c_byte * 64), ("s", C.c_byte * 64)]


class PackOut(C.Structure):
    _fields_ = [("out", C.c_void_p), ("row_is_d1", C.c_int), ("ntaps", C.c_int), ("kpad", C.c_int), ("pad_", C.c_int),
                ("taps", C.c_int * 16)]


class PackConv(C.Structure):
    _fields_ = [("w", C.c_void_p), ("d0", C.c_int), ("d1", C.c_int), ("rs", C.c_int), ("npacks", C.c_int),
                ("packs", PackOut * 5)]


_P, _I, _L, _F, _SZ = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/lighthand_hip.h declares
SIGNATURES = {
    "lh_version": (_I, []),
    "lh_last_error": (C.c_char_p, []),
    "lh_dtype_size": (_I, [_I]),
    "lh_image_to_nhwc4": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lh_image_u8_to_nhwc4": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _I, _P]),
    "lh_image_jitter_workspace_bytes": (_SZ, [_I]),
    "lh_image_u8_jitter_to_nhwc4": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P, _I, _P]),
    "lh_nhwc_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lh_nchw_f32_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "lh_pack_weight": (_I, [_P, _P, C.POINTER(_SZ), _I, _I, _L, _L, _L, _L, _I, C.POINTER(_I), _I, _P]),
    "lh_pack_chunk_elems": (_I, []),
    "lh_pack_weights_multi": (_I, [_P, _P, _P, _I, _I, _P]),
    "lh_pack_weights_tiled": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "lh_igemm": (_I, [C.POINTER(IgemmDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "lh_igemm_multi": (_I, [C.POINTER(IgemmCall), _I, _I, _P]),
    "lh_igemm_phases_rows": (_I, [C.POINTER(C.POINTER(IgemmDesc)), _I, _I]),
    "lh_igemm_phases": (_I, [C.POINTER(C.POINTER(IgemmDesc)), _I, _P, C.POINTER(C.c_void_p), _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "lh_igemm_phases_head": (_I, [C.POINTER(C.POINTER(IgemmDesc)), _I, _P, C.POINTER(C.c_void_p), _P, _P, C.POINTER(Head), _I, _P]),
    "lh_igemm_tile": (_I, [C.POINTER(IgemmDesc), _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "lh_igemm_config": (_I, [C.POINTER(IgemmDesc), _I, C.POINTER(_I)]),
    "lh_igemm_candidates": (_I, [C.POINTER(IgemmDesc), _I, C.POINTER(_I), _I]),
    "lh_wgrad_tile": (_I, [C.POINTER(IgemmDesc), _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "lh_wgrad_candidates": (_I, [C.POINTER(IgemmDesc), _I, _I, _I, C.POINTER(_I), _I]),
    "lh_igemm_stats_rows": (_I, [C.POINTER(IgemmDesc), _I]),
    "lh_igemm_gated_rows": (_I, [C.POINTER(IgemmDesc), _I, _I]),
    "lh_igemm_gated": (_I, [C.POINTER(IgemmDesc), _P, _P, _P, _P, _P, C.POINTER(BnBwdGate), _I, _P]),
    "lh_wgrad_slab_bytes": (_SZ, [C.POINTER(IgemmDesc), _I, _I, _I]),
    "lh_wgrad": (_I, [C.POINTER(IgemmDesc), _P, _P, _I, _I, _I, _P, _I, _P]),
    "lh_wgrad_rowfold": (_I, [C.POINTER(IgemmDesc), _I, _P, _P, _I, _I, _P, _I, _P]),
    "lh_wgrad_workspace_bytes": (_SZ, [C.POINTER(IgemmDesc), _I, _I, _I]),
    "lh_wgrad_fused": (_I, [C.POINTER(IgemmDesc), _I, _P, _P, _I, _I, _I, _P, _P, _L, _L, _L, _L, C.POINTER(_I), _I, _I, _P]),
    "lh_wgrad_fused_multi": (_I, [C.POINTER(WgradCall), _I, _I, _P]),
    "lh_wgrad_table_build": (_I, [C.POINTER(WgradCall), _I, _I, C.POINTER(_I), _I, _P, _P, _SZ, C.POINTER(WgradTableInfo)]),
    "lh_wgrad_table_run": (_I, [_P, C.POINTER(WgradTableInfo), _I, _P]),
    "lh_wgrad_reduce": (_I, [C.POINTER(IgemmDesc), _P, _P, _I, _I, _L, _L, _L, _L, C.POINTER(_I), _I, _I, _P]),
    "lh_bn_stats": (_I, [_P, _I, _I, _P, C.POINTER(_I), _I, _P]),
    "lh_bn_stats_rows": (_I, [_I, _I]),
    "lh_bn_stats_slab_bytes": (_SZ, [_I, _I]),
    "lh_bn_finalize": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "lh_bn_finalize_multi": (_I, [C.POINTER(BnFinalizeCall), _I, _P]),
    "lh_bn_eval_affine": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "lh_fuse_fwd": (_I, [C.POINTER(FuseDesc), _P, _I, _I, _I, _I, _I, _P]),
    "lh_fuse_fwd_multi": (_I, [C.POINTER(FuseFwdCall), _I, _I, _P]),
    "lh_fuse_bwd_multi": (_I, [C.POINTER(FuseBwdCall), _I, _I, _P]),
    "lh_fuse_bwd_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "lh_fuse_bwd": (_I, [C.POINTER(FuseBwdDesc), _I, _I, _I, _I, _P, _I, _P]),
    "lh_stem_pool": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "lh_bottleneck_infer": (_I, [C.POINTER(BottleneckDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "lh_stem_conv_rows": (_I, [_I, _I, _I]),
    "lh_stem_conv": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "lh_maxpool3x3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lh_maxpool3x3s2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lh_bn_relu_maxpool3x3s2_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "lh_maxpool3x3s2_bwd_gated_rows": (_I, [_I, _I, _I, _I, _I]),
    "lh_maxpool3x3s2_bwd_gated": (_I, [_P, _P, _P, C.POINTER(BnBwdGate), _I, _I, _I, _I, _I, _P]),
    "lh_gaussian_target": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "lh_gaussian_target_alt": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "lh_mse_workspace_bytes": (_SZ, [_L]),
    "lh_mse_heatmap": (_I, [_P, _P, _L, _P, _P, _P, _P, _P]),
    "lh_heatmap_argmax": (_I, [_P, _I, _I, _I, _F, _P, _P, _P, _P]),
    "lh_channel_sum_workspace_bytes": (C.c_size_t, [_I]),
    "lh_channel_sum_nchw": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "lh_channel_sum_nhwc": (_I, [_P, C.c_long, _I, _I, _P, _P, _I, _P]),
    "lh_copy_strided_f32": (_I, [_P, _P, _P, _P, _P, _P]),
    "lh_pck_curve": (_I, [_P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P]),
    "lh_heatmap_soft_argmax": (_I, [_P, _I, _I, _I, _F, _F, _P, _P]),
    "lh_heatmap_refine": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _P]),
    "lh_keypoint_metrics": (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "lh_comm_unique_id": (_I, [_P]),
    "lh_comm_init": (_I, [C.POINTER(_P), _I, _I, _P]),
    "lh_comm_allreduce_sum": (_I, [_P, _P, _SZ, _I, _P]),
    "lh_comm_reduce_scatter_sum": (_I, [_P, _P, _P, _SZ, _I, _P]),
    "lh_comm_allgather": (_I, [_P, _P, _P, _SZ, _I, _P]),
    "lh_comm_alltoall": (_I, [_P, _P, _P, _SZ, _I, _P]),
    "lh_comm_size": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "lh_comm_destroy": (_I, [_P]),
    "lh_cast_f32_bf16": (_I, [_P, _P, _L, _I, _P]),
    "lh_sum_chunks": (_I, [_P, _P, _I, _L, _I, _P]),
    "lh_adam_step": (_I, [_P, _P, _P, _P, _L, _P, _P, _P, _F, _P]),
    "lh_adam_tick": (_I, [_P, _P, _P, _P]),
    "lh_adam_apply": (_I, [_P, _P, _P, _P, _L, _P, _F, _P]),
}

_lib = None


def load():
    """Load the shared library (once) and set every prototype.  Raises LightHandError when
    the library is missing: build it with ``python -c 'import __graft_entry__ as g; g.build()'``
    or ``make -C lighthand_amd/csrc``."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must come first: the library has to bind to the HIP runtime torch itself loaded
    # (PyTorch owns device memory and streams); loading ours first would start a second runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise LightHandError(
            f"{LIB_PATH} not found: the HIP extension is the only backend of lighthand_amd "
            "(no CPU fallback). Build it with `make -C lighthand_amd/csrc`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().lh_last_error().decode("utf-8", "replace")
        raise LightHandError(f"{what or 'liblighthand_hip'} failed (status {rc}): {msg}")


def dtype_code(torch_dtype):
    import torch
    return {torch.float32: LH_F32, torch.bfloat16: LH_BF16, torch.float16: LH_F16}[torch_dtype]
