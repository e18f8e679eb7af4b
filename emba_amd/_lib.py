"""ctypes binding of the C ABI in include/emba_hip.h (libemba_hip.so, hand-written HIP for gfx950).

The library is built IN-TREE by `__graft_entry__.build()` / `emba_amd.build.build_hip()`.  There is no
fallback: if the shared object is missing or no GPU is visible, the product path raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMBA_LIB", os.path.join(_HERE, "libemba_hip.so"))   # EMBA_LIB: development builds of the same ABI

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_TIME_RANGE, ERR_STATE, ERR_CAPACITY, ERR_NUMERIC = range(8)
STATUS_NAMES = ["EMBA_OK", "EMBA_ERR_INVALID_ARG", "EMBA_ERR_NO_DEVICE", "EMBA_ERR_HIP", "EMBA_ERR_TIME_RANGE",
                "EMBA_ERR_STATE", "EMBA_ERR_CAPACITY", "EMBA_ERR_NUMERIC"]

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u16p = C.POINTER(C.c_uint16)
_u8p = C.POINTER(C.c_uint8)
_i64p = C.POINTER(C.c_int64)
_szp = C.POINTER(C.c_size_t)
_fp = C.POINTER(C.c_float)


class EmbaCfg(C.Structure):
    _fields_ = [("sensor_w", C.c_int32), ("sensor_h", C.c_int32), ("pano_w", C.c_int32), ("pano_h", C.c_int32),
                ("bearing_lut", _dp), ("C_th", C.c_double), ("event_batch", C.c_int32), ("outlier_px", C.c_double),
                ("device", C.c_int32), ("stream", C.c_void_p)]


# symbol -> (restype, argtypes); exactly the entry points include/emba_hip.h declares
SIGNATURES = {
    "emba_abi_version": (C.c_int, []),
    "emba_build_info": (C.c_char_p, []),
    "emba_create": (C.c_int, [C.POINTER(EmbaCfg), C.POINTER(C.c_void_p)]),
    "emba_destroy": (None, [C.c_void_p]),
    "emba_last_error": (C.c_char_p, [C.c_void_p]),
    "emba_set_events": (C.c_int, [C.c_void_p, _u16p, _u16p, _u8p, _i64p, C.c_size_t, _u16p, _u16p, _i64p, C.c_size_t]),
    "emba_set_events_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t]),
    "emba_last_setup_ms": (C.c_int, [C.c_void_p, _dp, _dp, _i32p, _szp, _szp]),
    "emba_last_order_stats": (C.c_int, [C.c_void_p, _dp, _dp]),
    "emba_last_order_inlier_estimate": (C.c_int, [C.c_void_p, _dp]),
    "emba_last_tile_geometry": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i32p, _i32p]),
    "emba_last_tile_drift": (C.c_int, [C.c_void_p, _szp, _i32p]),
    "emba_event_counts": (C.c_int, [C.c_void_p, _szp, _szp]),
    "emba_eval_data_error": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int64, C.c_int64, _dp, _dp, C.c_int32, _dp, _szp, _i32p]),
    "emba_form_normal_eq": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32, C.c_double, C.c_double, _dp, _dp, _szp, _u32p,
                                      C.c_size_t, _dp, _dp, _dp]),
    "emba_get_A12_sparse": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _dp, _dp, _dp, _dp]),
    "emba_compact_ep": (C.c_int, [C.c_void_p]),
    "emba_get_ep": (C.c_int, [C.c_void_p, _dp, C.c_size_t, _szp]),
    "emba_get_inlier_pixels": (C.c_int, [C.c_void_p, _u32p]),
    "emba_get_inlier_pixel_starts": (C.c_int, [C.c_void_p, _u32p]),
    "emba_get_ep_by_pixel": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_uint64)]),
    "emba_data_cost": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, _dp]),
    "emba_reg_cost": (C.c_int, [C.c_void_p, C.c_double, _dp]),
    "emba_costs": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, _dp, _dp]),
    "emba_costs_launch": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int32]),
    "emba_costs_finish": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, _dp, _dp]),
    "emba_dump_state": (C.c_int, [C.c_void_p, _dp, _dp, _i32p, _i32p, _i32p, _dp, _dp, _dp]),
    "emba_upload_map": (C.c_int, [C.c_void_p, _dp, _dp]),
    "emba_bind_map_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "emba_solve_normal_eq": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, _dp, _dp]),
    "emba_last_solve_info": (C.c_int, [C.c_void_p, _i32p]),
    "emba_solve_shard_size": (C.c_int, [C.c_void_p, _szp]),
    "emba_solve_shard_count": (C.c_int, [C.c_void_p, C.c_int32, _szp]),
    "emba_solve_shard_pack": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "emba_cg_shard_size": (C.c_int, [C.c_void_p, _szp]),
    "emba_cg_shard_begin": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_double, C.c_int32, C.c_void_p]),
    "emba_cg_shard_apply": (C.c_int, [C.c_void_p, C.c_void_p]),
    "emba_cg_shard_pt": (C.c_int, [C.c_void_p, C.c_void_p, _dp]),
    "emba_cg_shard_update": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
    "emba_cg_shard_direction": (C.c_int, [C.c_void_p, C.c_double]),
    "emba_cg_shard_end": (C.c_int, [C.c_void_p, _dp, C.c_void_p]),
    "emba_solve_shard_cached": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _i32p, _szp]),
    "emba_solve_shard_partial": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]),
    "emba_solve_shard_finish": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_double, C.c_int32, C.c_void_p, _dp, C.c_void_p]),
    "emba_solve_normal_eq_cg": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32, C.c_double, _dp, _dp, _i32p, _dp]),
    "emba_update_map": (C.c_int, [C.c_void_p, _dp, C.c_double]),
    "emba_update_map_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double]),
    "emba_map_accept": (C.c_int, [C.c_void_p]),
    "emba_map_reject": (C.c_int, [C.c_void_p]),
    "emba_trial_reject": (C.c_int, [C.c_void_p]),
    "emba_download_map": (C.c_int, [C.c_void_p, _dp, _dp]),
    "emba_get_map_active": (C.c_int, [C.c_void_p, _dp, C.c_size_t]),
    "emba_set_cost": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "emba_reconstruct_intensity": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "emba_bind_exchange_buffers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "emba_count_map_ready": (C.c_int, [C.c_void_p]),
    "emba_count_compress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "emba_count_expand": (C.c_int, [C.c_void_p, C.c_void_p]),
    "emba_step_form_active": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "emba_eval_launch": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int64, C.c_int64]),
    "emba_eval_finish": (C.c_int, [C.c_void_p, _dp, _szp, _i32p]),
    "emba_form_active": (C.c_int, [C.c_void_p, C.c_int32, _szp, _szp]),
    "emba_form_accumulate": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_double]),
    "emba_form_finish": (C.c_int, [C.c_void_p, C.c_double, _dp, _dp, _u32p, C.c_size_t, _dp, _dp, _dp]),
    "emba_step": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, _szp, _szp]),
    "emba_last_counts": (C.c_int, [C.c_void_p, _szp, _szp]),
    "emba_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "emba_get_option": (C.c_int, [C.c_void_p, C.c_char_p, _i32p]),
    "emba_sync": (C.c_int, [C.c_void_p]),
    "emba_timer_start": (C.c_int, [C.c_void_p, C.c_int32]),
    "emba_timer_stop": (C.c_int, [C.c_void_p, C.c_int32]),
    "emba_timer_elapsed_ms": (C.c_int, [C.c_void_p, C.c_int32, _fp]),
    "emba_enable_kernel_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "emba_last_kernel_ms": (C.c_int, [C.c_void_p, _fp, _fp]),
    "emba_kernel_ms_slot": (C.c_int, [C.c_void_p, C.c_int32, _fp, _fp]),
    "emba_kernel_timing_all": (C.c_int, [C.c_void_p, C.c_int32]),
    "emba_kernel_ms_all": (C.c_int, [C.c_void_p, C.c_int32, _fp]),
    "emba_bracket_overhead_us": (C.c_int, [C.c_void_p, C.c_int32, _fp]),
    "emba_clock_probe": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _i32p, _i32p, _i32p]),
    "emba_device_pci_bus_id": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    # single-process multi-GPU host
    "emba_group_create": (C.c_int, [C.POINTER(EmbaCfg), _i32p, C.c_int32, C.POINTER(C.c_void_p)]),
    "emba_group_create_flags": (C.c_int, [C.POINTER(EmbaCfg), _i32p, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "emba_group_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "emba_group_destroy": (None, [C.c_void_p]),
    "emba_group_last_error": (C.c_char_p, [C.c_void_p]),
    "emba_group_size": (C.c_int32, [C.c_void_p]),
    "emba_group_uses_rccl": (C.c_int32, [C.c_void_p]),
    "emba_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "emba_group_set_events": (C.c_int, [C.c_void_p, _u16p, _u16p, _u8p, _i64p, C.c_size_t]),
    "emba_group_upload_map": (C.c_int, [C.c_void_p, _dp, _dp]),
    "emba_group_step": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, _szp, _szp]),
    "emba_group_eval": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int64, C.c_int64, _dp, _dp, _dp, _szp, _i32p]),
    "emba_group_get_ep": (C.c_int, [C.c_void_p, _dp, C.c_size_t, _szp]),
    "emba_group_form": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, _szp, _szp]),
    "emba_group_set_cost": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "emba_group_apply_l2": (C.c_int, [C.c_void_p, C.c_double]),
    "emba_group_last_solve_exchanged": (C.c_int, [C.c_void_p, _i32p]),
    "emba_group_solve_cg": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32, C.c_double, _dp, _dp, _i32p, _dp]),
    "emba_group_trial_reject": (C.c_int, [C.c_void_p]),
    "emba_group_download": (C.c_int, [C.c_void_p, _dp, _dp, _u32p, C.c_size_t, _dp, _dp]),
    "emba_group_costs": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, _dp, _dp]),
    "emba_group_solve": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, _dp, _dp]),
    "emba_group_update_map": (C.c_int, [C.c_void_p, _dp, C.c_double]),
    "emba_group_map_accept": (C.c_int, [C.c_void_p]),
    "emba_group_map_reject": (C.c_int, [C.c_void_p]),
    "emba_group_download_map": (C.c_int, [C.c_void_p, _dp, _dp]),
    "emba_group_get_map_active": (C.c_int, [C.c_void_p, _dp, C.c_size_t]),
}


class EmbaError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES[status] if 0 <= status < len(STATUS_NAMES) else status}: {message}")
        self.status = status


_lib = None


def load():
    """Load libemba_hip.so (no GPU needed to load and resolve symbols). Raises if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). emba_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
