"""ctypes binding of libdemon_hip.so (include/demon_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is visible, loading or
context creation raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEMON_HIP_LIB", os.path.join(HERE, "libdemon_hip.so"))

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_int64_p = ctypes.POINTER(ctypes.c_int64)


class DemonOutputs(ctypes.Structure):
    _fields_ = [(name, c_float_p) for name in (
        "predict_flow5", "predict_conf5", "predict_flow2", "predict_conf2", "predict_depth2",
        "predict_normal2", "predict_rotation", "predict_translation", "predict_scale")]


class LanesEntry(ctypes.Structure):
    _fields_ = [("lanes", ctypes.c_int), ("placeholder_streams", ctypes.c_int), ("pairs_per_s", ctypes.c_float)]


class LanesResult(ctypes.Structure):
    """demon_lanes_result (DEMON_LANES_TABLE_CAP = 64)"""
    _fields_ = [("lanes", ctypes.c_int), ("placeholder_streams", ctypes.c_int), ("pairs_per_s", ctypes.c_float),
                ("verified_pairs_per_s", ctypes.c_float), ("attempts", ctypes.c_int), ("ntable", ctypes.c_int), ("table", LanesEntry * 64)]


class LaunchRecord(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64), ("kernel", ctypes.c_char * 32), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double), ("ms", ctypes.c_float), ("reduce_ms", ctypes.c_float)]


# every symbol include/demon_hip.h declares: name -> (restype, argtypes)
_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
SIGNATURES = {
    "demon_device_count": (_I, []),
    "demon_create": (_I, [ctypes.POINTER(_P), _I, _I, _I, _I]),
    "demon_create_v2": (_I, [ctypes.POINTER(_P), _I, _I, _I, _I]),
    "demon_create_ops": (_I, [ctypes.POINTER(_P), _I]),
    "demon_debug_timeline": (_I, [_P, ctypes.c_char_p, _I, ctypes.POINTER(ctypes.c_uint64), _I, c_int_p, c_float_p, ctypes.c_char_p, _I]),
    "demon_comm_get_unique_id": (_I, [ctypes.c_char_p]),
    "demon_comm_init_rank": (_I, [ctypes.POINTER(_P), _I, ctypes.c_char_p, _I, _I]),
    "demon_comm_destroy": (_I, [_P]),
    "demon_broadcast_weights": (_I, [_P, _P, _I, _I]),
    "demon_weights_slab_bytes": (ctypes.c_int64, [_P]),
    "demon_weights_slab_layout": (ctypes.c_uint64, [_P]),
    "demon_comm_count": (_I, [_P, c_int_p]),
    "demon_copy_weights_from": (_I, [_P, _P]),
    "demon_variant": (_I, [_P]),
    "demon_destroy": (_I, [_P]),
    "demon_last_error": (ctypes.c_char_p, [_P]),
    "demon_device": (_I, [_P]),
    "demon_num_variables": (_I, [_P]),
    "demon_variable_info": (_I, [_P, _I, ctypes.c_char_p, _I, c_int64_p, c_int_p]),
    "demon_set_weight": (_I, [_P, ctypes.c_char_p, c_float_p, c_int64_p, _I]),
    "demon_weights_blob_size": (ctypes.c_int64, [_P]),
    "demon_set_weights_blob": (_I, [_P, c_float_p, ctypes.c_int64]),
    "demon_set_weights_blob_device": (_I, [_P, _P, ctypes.c_int64]),
    "demon_set_option": (_I, [_P, ctypes.c_char_p, _I]),
    "demon_get_option": (_I, [_P, ctypes.c_char_p, c_int_p]),
    "demon_plan_clear": (_I, [_P, _I]),
    "demon_lanes_apply": (_I, [ctypes.POINTER(_P), _I, _I]),
    "demon_lanes_calibrate": (_I, [ctypes.POINTER(_P), _I, _I, _I, _I, _I, _I, ctypes.c_uint, ctypes.POINTER(LanesResult)]),
    "demon_set_cu_mask": (_I, [_P, ctypes.POINTER(ctypes.c_uint32), _I]),
    "demon_lanes_run_group": (_I, [ctypes.POINTER(_P), _I, _I, _I, _I]),
    "demon_hw_queues_hint": (_I, [_I]),
    "demon_autotune": (_I, [_P, _I]),
    "demon_num_layers": (_I, [_P]),
    "demon_plan_get": (_I, [_P, _I, _I, ctypes.c_char_p, _I, c_int_p, c_int_p, c_int_p]),
    "demon_plan_set": (_I, [_P, _I, ctypes.c_char_p, _I, _I, _I]),
    "demon_bootstrap": (_I, [_P, _I, c_float_p, c_float_p, ctypes.POINTER(DemonOutputs)]),
    "demon_iterative": (_I, [_P, _I, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                             ctypes.POINTER(DemonOutputs)]),
    "demon_refine": (_I, [_P, _I, c_float_p, c_float_p, c_float_p]),
    "demon_full": (_I, [_P, _I, c_float_p, c_float_p, _I, ctypes.POINTER(DemonOutputs), c_float_p]),
    "demon_upload_inputs": (_I, [_P, _I, c_float_p, c_float_p]),
    "demon_run_full": (_I, [_P, _I, _I]),
    "demon_run_bootstrap": (_I, [_P, _I]),
    "demon_synchronize": (_I, [_P]),
    "demon_release_streams": (_I, [_P]),
    "demon_acquire_streams": (_I, [_P]),
    "demon_download_outputs": (_I, [_P, _I, ctypes.POINTER(DemonOutputs), c_float_p]),
    "demon_download_normal0": (_I, [_P, _I, c_float_p]),
    "demon_upload_inputs_async": (_I, [_P, _I, c_float_p, c_float_p]),
    "demon_download_outputs_async": (_I, [_P, _I, ctypes.POINTER(DemonOutputs), c_float_p]),
    "demon_host_register": (_I, [ctypes.c_void_p, ctypes.c_int64]),
    "demon_host_unregister": (_I, [ctypes.c_void_p]),
    "demon_time_full": (_I, [_P, _I, _I, _I, c_float_p]),
    "demon_profile_full": (_I, [_P, _I, _I, _I, ctypes.POINTER(LaunchRecord), _I, c_int_p]),
    "demon_op_depth_to_flow": (_I, [_P, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, _I, _I, _I, _I, _I, _I]),
    "demon_op_flow_to_depth": (_I, [_P, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, _I, _I, _I, _I, _I, _I]),
    "demon_op_warp2d": (_I, [_P, c_float_p, c_float_p, c_float_p, _I, _I, _I, _I, _I, _I, _F]),
    "demon_op_leaky_relu": (_I, [_P, c_float_p, c_float_p, ctypes.c_int64, _F]),
    "demon_op_replace_nonfinite": (_I, [_P, c_float_p, c_float_p, ctypes.c_int64, _F]),
    "demon_op_scale_invariant_gradient": (_I, [_P, c_float_p, c_float_p, _I, _I, _I, c_int_p, c_float_p, _I, _F]),
    "demon_op_median3x3_downsample": (_I, [_P, c_float_p, c_float_p, _I, _I, _I]),
    "demon_op_depth_to_normals": (_I, [_P, c_float_p, c_float_p, c_float_p, _I, _I, _I, _I]),
    "demon_op_pointwise_l2_loss": (_I, [_P, c_float_p, c_float_p, c_float_p, _I, _I, _I, _I, ctypes.c_float]),
    "demon_op_conv2d": (_I, [_P, c_float_p, c_float_p, c_float_p, c_float_p] + [_I] * 12),
    "demon_op_deconv4x4s2": (_I, [_P, c_float_p, c_float_p, c_float_p, c_float_p] + [_I] * 6),
    "demon_op_dense": (_I, [_P, c_float_p, c_float_p, c_float_p, c_float_p] + [_I] * 4),
    "demon_bench_layer": (_I, [_P] + [_I] * 13 + [c_float_p, ctypes.POINTER(ctypes.c_double)]),
    "demon_last_kernel": (_I, [ctypes.c_char_p, _I]),
    "demon_debug_check_guards": (_I, [_P, c_int_p]),
}

_lib = None


def load():
    """Loads libdemon_hip.so and binds every symbol of include/demon_hip.h; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libdemon_hip.so not found at %s -- build it with `python -m demon_amd.build` "
            "(there is no CPU fallback for the DeMoN hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
