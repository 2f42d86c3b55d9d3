"""ctypes binding of libbyolo.so (include/byolo.h).  No fallback: if the HIP library is missing
the import fails loudly -- there is no CPU / eager path in the product."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BYOLO_LIB: load another build of the same library (the timing-ablation builds of csrc/build.py --ablate)
LIB_PATH = os.environ.get("BYOLO_LIB") or os.path.join(_HERE, "libbyolo.so")

OK, ERR_ARG, ERR_STATE, ERR_HIP, ERR_NOMEM, ERR_RANGE = 0, -1, -2, -3, -4, -5
DET_STANDARD, DET_ALEATORIC, DET_EPISTEMIC = 0, 1, 2
NMS_AGNOSTIC, NMS_TWO_CLASS = 0, 1
NORM_BN, NORM_DROPOUT = 1, 2


class ByoloError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbyolo error %d: %s" % (code, msg))
        self.code = code


class Cfg(ctypes.Structure):
    _fields_ = [("img_h", ctypes.c_int32), ("img_w", ctypes.c_int32), ("img_c", ctypes.c_int32),
                ("cls_cnt", ctypes.c_int32), ("drop_prob", ctypes.c_float), ("max_out", ctypes.c_int32),
                ("iou_thresh", ctypes.c_float), ("nms_mode", ctypes.c_int32), ("keep_all_outputs", ctypes.c_int32)]


class PlanOpts(ctypes.Structure):
    """include/byolo.h byolo_plan_opts (field for field; tests/test_abi.py compares the two)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "struct_bytes", "graphs", "serialize_convs", "serialize_heads", "dedup", "lowmain", "kx3", "p1", "b2b", "kx3_wide", "wino_split", "wino_split_min_c",
        "wino_split_bn", "wino_split_rounds", "winograd", "wino_fused", "stream1x1", "gemm_stream", "ksplit", "streamk",
        "plain_epilogue", "wshift_per_layer", "nms_general", "wino_split_persist")] + [(n, ctypes.c_float) for n in (
        "wino_split_min_gflop", "wino_split_chunk_mb", "wino_min_gflop", "wino_chunk_mb", "wino_min_ratio")]


_i32, _i64, _f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_vp, _cp, _sz, _u64 = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64
_P = ctypes.POINTER

# name -> (restype, argtypes); this table IS the Python view of include/byolo.h and is checked against
# the header by tests/test_abi.py
PROTOTYPES = {
    "byolo_create": (_i32, [_P(Cfg), _i32, _P(_vp)]),
    "byolo_destroy": (_i32, [_vp]),
    "byolo_last_error": (_cp, [_vp]),
    "byolo_version": (_cp, []),
    "byolo_add_conv": (_i32, [_vp, _cp, _i32, _i32, _i32, _i32]),
    "byolo_add_residual": (_i32, [_vp, _i32]),
    "byolo_add_route": (_i32, [_vp, _P(_i32), _i32]),
    "byolo_add_upsample": (_i32, [_vp]),
    "byolo_add_stack": (_i32, [_vp, _i32]),
    "byolo_add_detection": (_i32, [_vp, _cp, _i32, _P(_f32)]),
    "byolo_mark_backbone_end": (_i32, [_vp]),
    "byolo_num_params": (_i32, [_vp]),
    "byolo_param_info": (_i32, [_vp, _i32, _P(_cp), _P(_i32), _P(_i64)]),
    "byolo_set_param": (_i32, [_vp, _cp, _vp, _i64]),
    "byolo_get_param": (_i32, [_vp, _cp, _vp, _i64]),
    "byolo_finalize": (_i32, [_vp]),
    "byolo_num_layers": (_i32, [_vp]),
    "byolo_num_boxes": (_i32, [_vp, _P(_i64), _P(_i32)]),
    "byolo_workspace_bytes": (_i32, [_vp, _i32, _i32, _P(_sz)]),
    "byolo_forward": (_i32, [_vp, _vp, _i32, _i32, _u64, _i32, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "byolo_num_dropout": (_i32, [_vp]),
    "byolo_mask_layout": (_i32, [_vp, _i32, _i32, _i32, _P(_i64), _P(_i64)]),
    "byolo_set_async": (_i32, [_vp, _i32]),
    "byolo_status": (_i32, [_vp, _vp, _P(ctypes.c_uint32), _P(_i32)]),
    "byolo_clear_status": (_i32, [_vp, _vp]),
    "byolo_precision_note": (_cp, [_vp]),
    "byolo_set_first_image": (_i32, [_vp, _i64]),
    "byolo_max_images": (_i32, [_vp, _i32, _P(_i32)]),
    "byolo_layer_output": (_i32, [_vp, _i32, _P(_vp), _P(_i64)]),
    "byolo_copy_layer_output": (_i32, [_vp, _i32, _vp, _i64, _vp]),
    "byolo_set_precision": (_i32, [_vp, _i32]),
    "byolo_get_precision": (_i32, [_vp]),
    "byolo_decode": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _P(_f32), _i32, _vp, _i64, _i64, _vp]),
    "byolo_epistemic_stats": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "byolo_nms_workspace_bytes": (_sz, [_i32, _i64]),
    "byolo_sort_nms": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _sz, _vp, _vp, _vp, _vp]),
    "byolo_calibrate_bn": (_i32, [_vp, _vp, _i32, _vp, _sz, _vp]),
    "byolo_set_profiling": (_i32, [_vp, _i32]),
    "byolo_resume_profiling": (_i32, [_vp, _i32]),
    "byolo_stage_ms": (_i32, [_vp, _P(_f32)]),
    "byolo_set_profile_depth": (_i32, [_vp, _i32]),
    "byolo_select_profile": (_i32, [_vp, _i32]),
    "byolo_num_steps": (_i32, [_vp]),
    "byolo_step_profile": (_i32, [_vp, _i32, _P(_i32), _P(_i32), _P(_i64), _P(_f32), _P(ctypes.c_double)]),
    "byolo_step_split": (_i32, [_vp, _i32, _P(_i32), _P(_i32)]),
    "byolo_flops": (_i32, [_vp, _i32, _i32, _P(ctypes.c_double)]),
    "byolo_crc32c": (ctypes.c_uint32, [_vp, _sz]),
    "byolo_abi_version": (_i32, []),
    "byolo_get_plan_opts": (_i32, [_vp, _P(PlanOpts)]),
    "byolo_set_plan_opts": (_i32, [_vp, _P(PlanOpts)]),
    "byolo_graph_stats": (_i32, [_vp, _P(_i32), _P(_i64), _P(_i64), _P(_i64)]),
    "byolo_plan_num": (_i32, [_vp, _i32, _i32, _i32, _P(_i32), _P(_i32), _P(_i64)]),
    "byolo_plan_step": (_i32, [_vp, _i32, _P(_i32), _P(_i32), _P(_i32), _P(_i32)]),
    "byolo_plan_tensor": (_i32, [_vp, _i32, _P(_i64), _P(_i64), _P(_i32)]),
    "byolo_set_tshard": (_i32, [_vp, _i32, _i32]),
    "byolo_finish_tshard": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "byolo_normalize_u8": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "byolo_copy_status": (_i32, [_vp, _vp, _vp]),
    "byolo_png_decode_batch": (_i32, [_P(_vp), _P(_sz), _i32, _i32, _i32, _i32, _vp, _i32, _P(_i32), _P(_i32)]),
    "byolo_feed_records": (_i32, [_P(_i32), _P(_i64), _P(_i64), _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _P(_i32), _P(_i32)]),
    "byolo_encode_gt": (_i32, [_vp, _i32, _P(_i32), _P(ctypes.c_double), _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "byolo_encode_gt_workspace_bytes": (_sz, [_i32, _i32]),
    "byolo_loss_workspace_bytes": (_sz, []),
    "byolo_loss": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
    "byolo_format_ecp_json": (_i64, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _P(_cp), _i32, _vp, _sz]),
}


def _load():
    # PyTorch-ROCm bundles its own libamdhip64.so.7; device pointers and HIP streams are handed from
    # torch to libbyolo, so both MUST live in the same HIP runtime instance: import torch first, the
    # dynamic linker then resolves libbyolo's NEEDED libamdhip64.so.7 to the already-loaded one.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libbyolo.so not found at %s -- build it with `python bayesian-yolov3_amd/csrc/build.py` "
            "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    have = lib.byolo_abi_version()
    if have != ABI_VERSION:
        raise ImportError("%s speaks ABI %d, this binding was written for %d (include/byolo.h: BYOLO_ABI_VERSION) -- rebuild it with "
                          "`python bayesian-yolov3_amd/csrc/build.py --force`" % (LIB_PATH, have, ABI_VERSION))
    return lib


ABI_VERSION = 7
PNG_OK, PNG_UNSUPPORTED, PNG_SHAPE, PNG_CORRUPT, FEED_IO, FEED_CRC, FEED_PROTO = 0, 1, 2, 3, 4, 5, 6
lib = _load()


def check(handle, rc):
    if rc < 0:
        msg = lib.byolo_last_error(handle)
        raise ByoloError(rc, msg.decode("utf-8", "replace") if msg else "?")
    return rc
