"""ctypes binding of librobustcap_hip.so (include/robustcap_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the product path raises.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C robustcap_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RC_LIB_PATH") or os.path.join(_HERE, "csrc", "librobustcap_hip.so")   # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "robustcap_hip.h")

RC_FLAG_FIRST_FRAME = 1


class RcParams(C.Structure):
    _fields_ = [("conf_lo", C.c_double), ("conf_hi", C.c_double),
                ("contact_threshold", C.c_float), ("distance_threshold", C.c_float),
                ("height_threshold", C.c_float), ("tran_filter_num", C.c_double),
                ("use_flat_floor", C.c_int32), ("use_vision_updater", C.c_int32),
                ("use_imu_updater", C.c_int32), ("live", C.c_int32),
                ("update_vision_freq", C.c_int32), ("use_reproj_opt", C.c_int32), ("smooth", C.c_float),
                ("reserved", C.c_int32)]


class RcSmplifyInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_iter", C.c_int32), ("n_eval", C.c_int32), ("reserved", C.c_int32),
                ("first_loss", C.c_double), ("final_loss", C.c_double), ("host_ms", C.c_double), ("device_ms", C.c_double)]


OBJECTIVE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)


class RobustcapLibraryError(RuntimeError):
    pass


_P, _I32, _I64, _U32, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float
SIGNATURES = {
    "rc_create": (_I32, [_I32, _I32, C.POINTER(_P)]),
    "rc_destroy": (_I32, [_P]),
    "rc_last_error": (C.c_char_p, [_P]),
    "rc_default_params": (_I32, [_I32, C.POINTER(RcParams)]),
    "rc_get_params": (_I32, [_P, C.POINTER(RcParams)]),
    "rc_set_params": (_I32, [_P, C.POINTER(RcParams)]),
    "rc_load_weight": (_I32, [_P, C.c_char_p, _P, _I64]),
    "rc_finalize_weights": (_I32, [_P]),
    "rc_set_body": (_I32, [_P, _P, _P, _P, _P]),
    "rc_set_gravity": (_I32, [_P, _P]),
    "rc_reset": (_I32, [_P, _P, _P]),
    "rc_step": (_I32, [_P, _P, _P, _P, _P, _U32, _P, _P, _P]),
    "rc_sequence": (_I32, [_P, _I32, _P, _I64, _P, _I64, _P, _I64, _P, _U32, _P, _I64, _P, _I64, _P]),
    "rc_set_gemm_mode": (_I32, [_P, _I32]),
    "rc_get_gemm_mode": (_I32, [_P]),
    "rc_default_gemm_mode": (_I32, [_I32]),
    "rc_set_sequence_mode": (_I32, [_P, _I32, _I32]),
    "rc_get_sequence_stats": (_I32, [_P, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    "rc_get_launch_stats": (_I32, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "rc_get_launch_stats_w32": (_I32, [_P, C.POINTER(_I64)]),
    "rc_set_resident": (_I32, [_P, _I32, _I32]),
    "rc_get_resident_stats": (_I32, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "rc_plan_sequence": (_I32, [_P, _I32, _I32, _P, _U32, _I32, _P]),
    "rc_plan_wave": (_I32, [_P, _I32, _I32, _I32, _P, _P, _I32, _I32, _P, _I64, C.POINTER(_I32), C.POINTER(_I32), _P, _P]),
    "rc_live_begin": (_I32, [_P]),
    "rc_live_step": (_I32, [_P, _P, _P, _P, _P, _U32, _P, _P]),
    "rc_live_end": (_I32, [_P]),
    "rc_get_live_stats": (_I32, [_P, _P, _P]),
    "rc_get_live_prestep": (_I32, [_P, _P, _P]),
    "rc_get_live_replayed": (_I32, [_P, _P]),
    "rc_get_live_spin": (_I32, [_P, _P, _P]),
    "rc_get_live_profile": (_I32, [_P, _P]),
    "rc_get_live_last_profile": (_I32, [_P, _P]),
    "rc_get_live_backend": (_I32, [_P, _P, _P, _P, _I32]),
    "rc_get_fusion_state": (_I32, [_P, _P, _P]),
    "rc_r6d_to_rotmat": (_I32, [_P, _P, _I64, _P]),
    "rc_axis_angle_to_rotmat": (_I32, [_P, _P, _I64, _P]),
    "rc_rotmat_to_axis_angle": (_I32, [_P, _P, _I64, _P]),
    "rc_rotmat_to_r6d": (_I32, [_P, _P, _I64, _P]),
    "rc_angle_between": (_I32, [_P, _P, _P, _I64, _P]),
    "rc_lerp": (_I32, [_P, _P, C.c_double, _P, _I64, _P]),
    "rc_normalize_rows": (_I32, [_P, _P, _P, _I64, _I32, _P]),
    "rc_bbox_normalise": (_I32, [_P, _P, _I64, _P]),
    "rc_shape_body": (_I32, [_P, _P, _P, _P, _P, _I32, _P, _P]),
    "rc_fk_r": (_I32, [_P, _P, _P, _I64, _P]),
    "rc_bone_to_joint": (_I32, [_P, _P, _P, _I64, _P]),
    "rc_joint_to_bone": (_I32, [_P, _P, _P, _I64, _P]),
    "rc_zero_pose": (_I32, [_P, _P, _P, _P]),
    "rc_ik_r": (_I32, [_P, _P, _P, _I64, _P]),
    "rc_fk_bone": (_I32, [_P, _P, _P, _I64, _P]),
    "rc_body_fk": (_I32, [_P, _P, _P, _P, _P, _P, _I64, _P]),
    "rc_set_mesh": (_I32, [_P, _P, _P, _I32]),
    "rc_body_mesh": (_I32, [_P, _P, _P, _P, _I64, _P]),
    "rc_set_regressor": (_I32, [_P, _P, _I32, _I32]),
    "rc_mesh_metrics": (_I32, [_P, _P, _P, _I64, _P, C.POINTER(C.c_double), _P]),
    "rc_synth_imu": (_I32, [_P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P]),
    "rc_syn_acc": (_I32, [_P, _I64, _I64, _I32, _P, _P]),
    "rc_procrustes_error": (_I32, [_P, _P, _I64, _I32, _P, _P]),
    "rc_position_error": (_I32, [_P, _P, _I64, _P, C.POINTER(C.c_double), _P]),
    "rc_lstm_step": (_I32, [_P, C.c_char_p, _P, _P, _P, _P]),
    "rc_set_ignored_landmarks": (_I32, [_P, _P, _I32]),
    "rc_reproj_residual": (_I32, [_P, _P, _P, _P, _P, _F, _P, _I64, _P]),
    "rc_smplify_set_prior": (_I32, [_P, _P, _P, _P]),
    "rc_smplify_set_ref3d": (_I32, [_P, _P]),
    "rc_smplify_loss_grad": (_I32, [_P, _P, _P, _P, _P, _P, _I64, C.POINTER(C.c_double), _P, _P]),
    "rc_smplify_run": (_I32, [_P, _P, _P, _P, _P, _P, _I64, _F, _I32, _F, _P, _P, _P, C.POINTER(RcSmplifyInfo), _P]),
    "rc_smplify_run_batch": (_I32, [_P, _I32, _P, _P, _P, _P, _P, _P, _F, _I32, _F, _P, _P, _P, _P, _P]),
    "rc_lbfgs_minimize": (_I32, [OBJECTIVE_FN, _P, _I64, C.POINTER(C.c_double), C.c_double, _I32, _I32, _I32, C.c_double,
                                 C.c_double, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(C.c_double), _I64]),
    "rc_camera_inputs": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "rc_camera_inputs_rows": (_I32, [_P, _P, _P, _P, _P, _P, _P, _F, _F, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "rc_get_state": (_I32, [_P, C.c_char_p, _P, _P, _P]),
    "rc_get_trace": (_I32, [_P, _P, _P]),
    "rc_gemm_timing": (_I32, [_P, _I32]),
    "rc_gemm_timing_read": (_I32, [_P, C.POINTER(C.c_double), C.POINTER(_I64)]),
    "rc_gemm_timing_busy": (_I32, [_P, C.POINTER(C.c_double)]),
}

_lib = None


# entry points newer than round 4: the only ones an A/B library under RC_LIB_PATH may lack
OPTIONAL_IN_AB_BUILDS = frozenset({"rc_get_launch_stats_w32", "rc_set_resident", "rc_get_resident_stats", "rc_get_launch_stats", "rc_get_live_spin", "rc_get_live_replayed", "rc_get_live_profile", "rc_get_live_last_profile",
                                   "rc_get_live_prestep"})


def load():
    """Load the shared library once and attach prototypes. Raises RobustcapLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RobustcapLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c \"import __graft_entry__ as g; g.build()\"` at the repo root. There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RobustcapLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    # A library under RC_LIB_PATH (tools/ab.py: an A/B build of an older revision) may lack entry points added since -- but only those on the
    # allow-list below, and it says so; anything else missing is a header / library mismatch and fails HERE, not at the first call.
    ab_build = bool(os.environ.get("RC_LIB_PATH"))
    skipped = []
    for name, (res, args) in SIGNATURES.items():
        if ab_build and name in OPTIONAL_IN_AB_BUILDS and not hasattr(lib, name):
            skipped.append(name)
            continue
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RobustcapLibraryError(f"{LIB_PATH} does not export {name}: the library does not match include/robustcap_hip.h") from e
        fn.restype, fn.argtypes = res, args
    if skipped:
        import warnings
        warnings.warn(f"RC_LIB_PATH={LIB_PATH}: entry points missing in this build (calls to them will fail): {', '.join(skipped)}")
    _lib = lib
    return lib


def check(ctx, rc, what):
    if rc != 0:
        msg = load().rc_last_error(ctx)
        raise RobustcapLibraryError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device/host pointer of a contiguous torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
