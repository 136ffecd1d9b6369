"""ctypes description of the C ABI declared in ``include/sgcdet_amd.h``.

One table drives every binding of that header: the product library
(``sgcdet_amd/csrc/libsgcdet_amd.so``, HIP/gfx950) and -- from the test side
only -- the CPU oracle that exports the same symbols.  Nothing here touches
torch; pointers are plain integers (``tensor.data_ptr()``).
"""
import ctypes as C

_p = C.c_void_p
_i = C.c_int
_f = C.c_float

# name -> argtypes (restype is int for all but the introspection / query calls)
SIGNATURES = {
    "sgc_depth_score_forward": [_p, _p, _p, _p, _p] + [_i] * 7 + [_p],
    "sgc_wms_forward": [_p, _p, _p, _p, _p, _p, _p] + [_i] * 7 + [_p],
    "sgc_wms_backward": [_p] * 11 + [_i] * 7 + [_p],
    "sgc_depth_score_backward": [_p] * 7 + [_i] * 7 + [_p],
    "sgc_dfa3d_forward": [_p] * 8 + [_i] * 9 + [_p],
    "sgc_dfa3d_backward": [_p] * 11 + [_i] * 9 + [_p],
    "sgc_dfa3d_forward_items": [_p] * 9 + [_i] * 9 + [_p],
    "sgc_dfa3d_backward_items": [_p] * 12 + [_i] * 9 + [_p],
    "sgc_dfa3d_backward_binned": [_p] * 11 + [_i] * 13 + [_p],
    "sgc_project_points": [_p] * 6 + [_i, _i, _f, _f, _f, _f, _p],
    "sgc_compact_pairs": [_p, _i, _i] + [_p] * 9 + [_p],
    "sgc_pairs_geometry_sample": [_p] * 7 + [_i] * 9 + [_p],
    "sgc_pairs_geometry_linear_bf16x3": [_p] * 11 + [_i] * 10 + [_p],
    "sgc_pairs_deform_gather": [_p] * 9 + [_i] * 12 + [_p],
    "sgc_depth_pairs": [_p, _p] + [_i] * 5 + [_p],
    "sgc_bin_pairs": [_p] * 9 + [_i] * 7 + [_p],
    "sgc_pairs_deform_gather_tiled": [_p, _i] + [_p] * 6 + [_i] * 15 + [_p],
    "sgc_linear_rows_headmajor_bf16x3": [_p] * 5 + [_i] * 6 + [_p],
    "sgc_tile_window": [_i] * 12 + [C.POINTER(C.c_int)] * 5,
    "sgc_view_mean": [_p] * 4 + [_i] * 3 + [_p, _i] + [_p],
    "sgc_view_attend": [_p] * 5 + [_i] * 4 + [_p, _i] + [_p],
    "sgc_view_attend_backward": [_p] * 8 + [_i] * 5 + [_p],
    "sgc_view_attend_pq": [_p] * 5 + [_i] * 4 + [_p, _i] + [_p],
    "sgc_scatter_rows": [_p] * 4 + [_p, _i, _i, _p],
    "sgc_nchw_to_nhwc_crop": [_p, _p] + [_i] * 7 + [_p],
    "sgc_nhwc_to_nchw_pad": [_p, _p] + [_i] * 6 + [_p],
    "sgc_conv3d_cl_f32": [_p] * 6 + [_i] * 9 + [_p, C.c_int64] + [_p],
    "sgc_conv3d_cl_bf16x3": [_p] * 7 + [_i] * 9 + [_p, C.c_int64] + [_p],
    "sgc_conv3d_cl_bf16x3_masked": [_p] * 8 + [_i] * 6 + [_p, C.c_int64] + [_p],
    "sgc_conv3d_cl_bf16x3_act": [_p] * 8 + [_i] * 8 + [_p, _p, C.c_int64] + [_p],
    "sgc_conv3d_winograd_z_bf16x3": [_p] * 7 + [_i] * 6 + [_p, C.c_int64] + [_p],
    "sgc_conv2d_nhwc_bf16x3": [_p] * 7 + [_i] * 7 + [_p],
    "sgc_conv3d_wgrad_bf16x3": [_p] * 3 + [_i] * 7 + [_p, C.c_int64] + [_p],
    "sgc_mask_dilate3": [_p, _p, _i, _i, _i, _p],
    "sgc_valid_pyramid": [_p, _p, _i, _i, _i, _i, _p],
    "sgc_linear_rows_bf16x3": [_p] * 6 + [_i] * 3 + [_p],
    "sgc_linear_rows_zrow_bf16x3": [_p] * 6 + [_i] * 3 + [_p],
    "sgc_linear_rows_blockdiag_bf16x3": [_p] * 6 + [_i] * 4 + [_p],
    "sgc_level_tail": [_p] * 7 + [_f] + [_p] * 8 + [_f] + [_p] + [_i] * 3 + [_p],
    "sgc_topk_select": [_p, _i, _i, _p, _p, _p, _p],
    "sgc_topk_select_ws": [_p, _i, _i, _p, _p, _p, _p, C.c_int64, _p],
    "sgc_layer_norm_rows": [_p, _p, _p, _f, _p, _p, _i, _i, _p],
    "sgc_bn_rows_forward": [_p] * 5 + [_f, _f] + [_p] * 4 + [C.c_int64, _i, _i, _p],
    "sgc_bn_rows_backward": [_p] * 9 + [C.c_int64, _i, _i, _p],
    "sgc_bn_rows_act_forward": [_p] * 5 + [_f, _f] + [_p, _i] + [_p] * 4 + [C.c_int64, _i, _i, _p],
    "sgc_bn_rows_act_backward": [_p] * 11 + [C.c_int64, _i, _i, _p],
    "sgc_aligned_nms3d": [_p] * 3 + [_f] + [_p] * 3 + [_i] + [_p],
    "sgc_nms_rotated_bev": [_p] * 3 + [_f] + [_p] * 3 + [_i, _i] + [_p],
    "sgc_box_iou_rotated": [_p] * 3 + [_i, _i] + [_p],
    "sgc_assign_targets": [_p] * 4 + [_i] * 4 + [_p] * 5 + [_i, _i] + [_p],
    "sgc_plane_sweep_corr": [_p] * 5 + [_i] * 6 + [_p],
    "sgc_upsample2x_occ": [_p] * 5 + [_i] * 4 + [_p],
    "sgc_upsample2x_backward": [_p, _p] + [_i] * 4 + [_p],
    "sgc_scatter_add_rows": [_p] * 3 + [_i, _i, _p],
    "sgc_set_tuning": [C.c_char_p, _i],
    "sgc_set_conv_products": [_i],
    "sgc_pack_conv_weight": [_p] * 3 + [_i] * 7 + [_p],
    "sgc_unpack_conv_wgrad": [_p] * 2 + [_i] * 7 + [_p],
    "sgc_pack_conv_weight_batch": [_p] + [_i] * 3 + [_p],
}

INTROSPECTION = {
    "sgc_abi_version": (C.c_int, []),
    "sgc_last_error": (C.c_char_p, []),
    "sgc_backend": (C.c_char_p, []),
    "sgc_conv3d_workspace_floats": (C.c_int64, [_i] * 9),
    "sgc_conv3d_wgrad_workspace_floats": (C.c_int64, [_i] * 7),
    "sgc_bn_rows_workspace_floats": (C.c_int64, [_i] * 2),
    "sgc_topk_select_workspace_bytes": (C.c_int64, [_i]),
    "sgc_level_tail_supported": (C.c_int, [_i] * 2),
    "sgc_view_attend_pq_supported": (C.c_int, [_i] * 3),
    "sgc_conv3d_winograd_z_supported": (C.c_int, [_i] * 5),
    "sgc_pack_conv_weight_blocks": (C.c_int, [_i] * 4),
    "sgc_conv3d_winograd_z_workspace_floats": (C.c_int64, [_i] * 5),
    "sgc_get_conv_products": (C.c_int, []),
    "sgc_bin_pairs_workspace_bytes": (C.c_int64, [_i] * 7),
    "sgc_dfa3d_backward_binned_lds_bytes": (C.c_int64, [_i] * 8),
    "sgc_pairs_geometry_linear_supported": (C.c_int, [_i] * 4),
    "sgc_linear_rows_blockdiag_supported": (C.c_int, [_i] * 3),
    "sgc_pairs_geometry_linear_workspace_bytes": (C.c_int64, [_i]),
}

ABI_VERSION = 4      # == SGC_ABI_VERSION of include/sgcdet_amd.h (tests/test_abi_cpu.py compares the two)


class SgcError(RuntimeError):
    """Raised when an entry point returns a negative status."""


class Library:
    """A loaded shared object exporting the sgcdet_amd C ABI."""

    def __init__(self, path):
        self.path = str(path)
        self._dll = C.CDLL(self.path)
        missing = []
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.argtypes = argtypes
            fn.restype = C.c_int
        for name, (res, argtypes) in INTROSPECTION.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.argtypes = argtypes
            fn.restype = res
        if missing:
            raise ImportError(f"{self.path} does not export: {', '.join(missing)}")
        if self._dll.sgc_abi_version() != ABI_VERSION:
            raise ImportError(
                f"{self.path}: ABI version {self._dll.sgc_abi_version()} != {ABI_VERSION}")

    @property
    def backend(self):
        return self._dll.sgc_backend().decode()

    def last_error(self):
        return self._dll.sgc_last_error().decode()

    def call(self, name, *args):
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise SgcError(f"{name} failed with status {rc}: {self.last_error()}")
        return rc
