"""ctypes binding of libtreelearn_hip.so (the C ABI declared in include/treelearn_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, a RuntimeError is
raised.  PyTorch only provides device memory (`tensor.data_ptr()`) and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtreelearn_hip.so")

TL_F32, TL_BF16, TL_F16 = 0, 1, 2
TL_OK, TL_ERR_ARG, TL_ERR_UNSUPPORTED = 0, -1, -3
TL_EPI_NONE, TL_EPI_STATS, TL_EPI_BN_BWD = 0, 1, 2
# opt-in, developer build only: the window form of the 27-tap bf16 convs (csrc/tl_conv_win.hip) for levels of >= 65536 voxels; measured
# at parity with the register-gather kernels on the config-2 tile, so the release library does not carry it
WIN_KERNEL = os.environ.get("TL_CONV_WIN") == "1"
_c = ctypes
_vp, _i64, _i32, _f32 = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_float


class ConvArgs(_c.Structure):
    _fields_ = [
        ("in_", _vp), ("in_ld", _i64), ("weight", _vp), ("table", _vp),
        ("n_out", _i64), ("n_in", _i64), ("K", _i32), ("Cin", _i32), ("Cout", _i32), ("dtype", _i32),
        ("in_scale", _vp), ("in_shift", _vp), ("in_relu", _i32), ("out_relu", _i32),
        ("residual", _vp), ("res_ld", _i64), ("out_scale", _vp), ("out_shift", _vp),
        ("out", _vp), ("out_ld", _i64),
        ("out2", _vp), ("out2_ld", _i64), ("out2_scale", _vp), ("out2_shift", _vp), ("out2_relu", _i32),
        ("out3", _vp), ("out3_ld", _i64), ("out3_scale", _vp), ("out3_shift", _vp), ("out3_relu", _i32),
        ("weight_frag", _vp), ("table_one_hot", _i32), ("table_compact", _vp),
        # training-mode epilogue reductions (include/treelearn_hip.h: TL_EPI_STATS / TL_EPI_BN_BWD)
        ("in_all_ones", _i32), ("epi_mode", _i32), ("red_part", _vp), ("red_nparts", _c.POINTER(_i32)),
        ("bn_x", _vp), ("bn_x_ld", _i64), ("bn_mean", _vp), ("bn_rstd", _vp), ("bn_scale", _vp), ("bn_shift", _vp), ("bn_relu", _i32),
        # block-local form of a 27-tap rulebook (tl_blk_build)
        ("blk_unit", _vp), ("blk_counter", _vp), ("blk_halo", _vp), ("blk_lrb", _vp), ("blk_pmask", _vp), ("table_scatter", _vp),
        ("weight_x3", _vp),
    ]


class TileBox(_c.Structure):               # tl_tile_box
    _fields_ = [("outer", _c.c_float * 4), ("inner", _c.c_double * 4), ("center", _c.c_double * 2), ("half_inner", _c.c_float)]


class Level(_c.Structure):                 # tl_level
    _fields_ = [("dims", _c.c_int32 * 4), ("n", _c.c_int64), ("bitmap", _c.c_void_p), ("prefix", _c.c_void_p), ("coords", _c.c_void_p),
                ("nbr", _c.c_void_p), ("compact", _c.c_void_p), ("child", _c.c_void_p), ("parent", _c.c_void_p), ("inv", _c.c_void_p),
                ("o2n", _c.c_void_p), ("inv_packed", _c.c_void_p)]


class Blk(_c.Structure):                   # tl_blk
    _fields_ = [("o2n", _vp), ("perm", _vp), ("coords_new", _vp), ("unit", _vp), ("counter", _vp), ("halo", _vp), ("lrb", _vp), ("pmask", _vp),
                ("cap_units", _i64), ("halo_max", _i32), ("reserved", _i32), ("nn", _vp)]


class HdbGrid(_c.Structure):               # TlHdbGrid
    _fields_ = [("lo", _c.c_double * 2), ("h", _c.c_double), ("levels", _c.c_int32), ("reserved", _c.c_int32)]


TL_MAX_LEVELS = 8
TL_ERR_ARENA, TL_ERR_REACH_ZERO, TL_ERR_EXTENT, TL_ERR_BLK = -4, -5, -6, -7


class Affine(_c.Structure):                # tl_affine
    _fields_ = [("scale", _vp), ("shift", _vp)]


class Weight(_c.Structure):                # tl_weight
    _fields_ = [("w", _vp), ("frag", _vp), ("K", _i32), ("Cout", _i32), ("Cin", _i32), ("reserved", _i32), ("x3", _vp)]


class ResDesc(_c.Structure):               # tl_res_desc
    _fields_ = [("bn0", Affine), ("w1", Weight), ("bn3", Affine), ("w2", Weight), ("w1x1", Weight), ("w1_half", Weight * 2)]


class UBlockDesc(_c.Structure):            # tl_ublock_desc
    _fields_ = [("C", _i32), ("deeper", _i32), ("blocks", ResDesc * 2), ("bn_down", Affine), ("wd", Weight), ("bn_up", Affine), ("wu", Weight),
                ("tail", ResDesc * 2), ("bn_cat_l", Affine), ("bn_cat_r", Affine)]


class NetDesc(_c.Structure):               # tl_net_desc
    _fields_ = [("dtype", _i32), ("num_levels", _i32), ("voxel_size", _f32), ("has_shape", _i32), ("spatial_shape", _i32 * 3), ("blocked", _i32),
                ("in_channels", _i32), ("use_coords", _i32), ("use_feats", _i32), ("max_points_per_voxel", _i32), ("reserved0", _i32),
                ("w_in", Weight), ("u", UBlockDesc * TL_MAX_LEVELS), ("out_bn", Affine),
                ("head_w1", _vp), ("head_b1", _vp), ("head_w2", _vp), ("head_b2", _vp)]


class LaunchRec(_c.Structure):             # tl_launch_rec
    _fields_ = [("level", _i32), ("kind", _i32), ("K", _i32), ("Cin", _i32), ("Cout", _i32), ("residual", _i32), ("esize", _i32), ("split_part", _i32),
                ("split_cin", _i32), ("in_prologue", _i32), ("n_out", _i64), ("n_in", _i64), ("ms", _f32)]


class ForwardArgs(_c.Structure):           # tl_forward_args
    _fields_ = [("xyz", _vp), ("batch_ids", _vp), ("N", _i64), ("B", _i32), ("reserved", _i32), ("point_feats", _vp), ("arena", _vp), ("arena_bytes", _i64),
                ("backbone", _vp), ("logits", _vp), ("offsets", _vp), ("side_stream", _vp), ("needed_bytes", _i64),
                ("level_n", _i64 * TL_MAX_LEVELS), ("blocked_used", _i32), ("launches", _i32)]


_I4 = _i32 * 4
_I3 = _i32 * 3

# name -> (restype, argtypes); every symbol include/treelearn_hip.h declares
PROTOTYPES = {
    "tl_version": (_i32, []),
    "tl_error_string": (_c.c_char_p, [_i32]),
    "tl_set_tuning": (_i32, [_c.c_char_p, _i64]),
    "tl_voxel_point_coords": (_i32, [_vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "tl_voxel_point_coords_one": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp, _vp, _c.POINTER(_i32), _vp]),
    "tl_bitmap_from_points": (_i32, [_vp, _i64, _I4, _vp, _vp]),
    "tl_bitmap_down": (_i32, [_vp, _I4, _I3, _vp, _I4, _vp]),
    "tl_scan_ws_words": (_i64, [_i64]),
    "tl_bitmap_scan": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "tl_expand_coords": (_i32, [_vp, _vp, _I4, _vp, _vp]),
    "tl_point_rank": (_i32, [_vp, _i64, _vp, _vp, _I4, _vp, _vp]),
    "tl_voxel_mean_feats": (_i32, [_vp, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "tl_voxel_feats": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "tl_rulebook_subm": (_i32, [_vp, _i64, _vp, _vp, _I4, _vp, _vp, _vp]),
    "tl_rulebook_down": (_i32, [_vp, _i64, _vp, _vp, _I4, _i64, _vp, _vp, _vp, _vp]),
    "tl_rulebook_compact": (_i32, [_vp, _i64, _vp, _vp]),
    "tl_pyramid_ws_words": (_i64, [_I4, _i32, _c.POINTER(_i64)]),
    "tl_pyramid_build": (_i32, [_vp, _i64, _I4, _I3, _i32, _vp, _vp, _vp, _vp, _vp]),
    "tl_rulebooks_build": (_i32, [_c.POINTER(Level), _i32, _vp, _i64, _vp, _i64, _vp, _vp]),
    "tl_blk_ws_words": (_i64, [_I4]),
    "tl_blk_build": (_i32, [_vp, _vp, _I4, _i64, _c.POINTER(Blk), _vp, _i32, _vp]),
    "tl_conv_fwd": (_i32, [_c.POINTER(ConvArgs), _vp]),
    "tl_conv_red_parts": (_i64, [_i64]),
    "tl_exec_create": (_vp, []),
    "tl_exec_destroy": (None, [_vp]),
    "tl_forward": (_i32, [_vp, _c.POINTER(NetDesc), _c.POINTER(ForwardArgs), _vp]),
    "tl_exec_check": (_i32, [_vp]),
    "tl_exec_profile": (_i32, [_vp, _i32]),
    "tl_exec_profile_read": (_i32, [_vp, _c.POINTER(LaunchRec), _i32]),
    "tl_bn_train_finish": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tl_bn_train_bwd_from_parts": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "tl_pack_weight": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "tl_pack_weight_x3": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "tl_pack_weight_x3_bytes": (_i64, [_i32, _i32, _i32]),
    "tl_pack_weight_frag": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "tl_pack_weights_batch": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "tl_pack_weight_dgrad": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "tl_conv_wgrad_ws_floats": (_i64, [_i64, _i32, _i32, _i32]),
    "tl_conv_wgrad": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "tl_conv_wgrad_ref": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "tl_conv_wgrad_blk_ws_floats": (_i64, []),
    "tl_conv_wgrad_blk": (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp]),
    "tl_head_mlp": (_i32, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tl_affine_relu": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _vp]),
    "tl_bn_ws_doubles": (_i64, [_i64, _i32]),
    "tl_bn_train_stats": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tl_bn_train_bwd": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "tl_linear_small_f32": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _i64, _vp, _i64, _vp]),
    "tl_gather_rows": (_i32, [_vp, _i64, _i32, _i32, _i64, _vp, _i64, _vp, _i64, _vp]),
    "tl_scatter_add_rows": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "tl_compact_ws_words": (_i64, [_i64]),
    "tl_compact_rows": (_i32, [_vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "tl_tile_crop_ws_words": (_i64, [_i64]),
    "tl_tile_crop": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tl_cell_keys": (_i32, [_vp, _i64, _c.c_double, _c.c_double, _vp, _i32, _vp, _vp, _vp]),
    "tl_downsample_ws_words": (_i64, [_i64]),
    "tl_downsample_reduce": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tl_group_mean": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tl_verticality": (_i32, [_vp, _vp, _i64, _c.c_double, _vp, _vp, _vp]),
    "tl_cluster_ws_bytes": (_i64, [_i64]),
    "tl_cluster_grid": (_i32, [_vp, _i64, _c.c_double, _vp, _vp, _vp, _vp]),
    "tl_hdbscan_ws_bytes": (_i64, [_i64]),
    "tl_hdbscan_mst": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tl_hdbscan_labels_host": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "tl_hdbscan_prim_order_host": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "tl_hdbscan_grid_plan_ws_bytes": (_i64, []),
    "tl_hdbscan_grid_plan": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "tl_hdbscan_grid_ws_bytes": (_i64, [_i64, _vp]),
    "tl_hdbscan_grid_ws_bytes_k": (_i64, [_i64, _vp, _i32]),
    "tl_hdbscan_mst_grid": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tl_knn_vote": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    "tl_knn_vote_grid": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _c.c_float * 3, _f32, _I3, _vp, _i64, _i32, _vp, _vp]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m treelearn_amd.build` "
                "(or __graft_entry__.build()). There is no CPU fallback for the product path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        if WIN_KERNEL and L.tl_set_tuning(b"win", 1) != 0:
            raise RuntimeError("TL_CONV_WIN=1 needs the developer build of the library (python -m treelearn_amd.build --dev): the window conv "
                               "kernel is not part of the release build")
        for kv in filter(None, os.environ.get("TL_TUNING", "").split(",")):          # developer: TL_TUNING="key=value,..." -> tl_set_tuning at load
            k, _, v = kv.partition("=")
            if L.tl_set_tuning(k.strip().encode(), int(v)) != 0:
                raise RuntimeError(f"TL_TUNING: unknown key {k!r}")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().tl_error_string(rc).decode()} ({rc})")


def ptr(t):
    return None if t is None else _vp(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (the raw handle: no Stream object per launch)."""
    if _raw_stream is not None:
        return _vp(_raw_stream(torch.cuda.current_device()))
    return _vp(torch.cuda.current_stream().cuda_stream)


def dims4(d):
    return _I4(*[int(v) for v in d])


def dims3(d):
    return _I3(*[int(v) for v in d])


def dtype_code(dt):
    if dt == torch.float32:
        return TL_F32
    if dt == torch.bfloat16:
        return TL_BF16
    if dt == torch.float16:
        return TL_F16                # inference path (tl_conv_fwd, tl_pack_weight*, tl_head_mlp, tl_affine_relu)
    raise ValueError(f"unsupported compute dtype {dt}")


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
