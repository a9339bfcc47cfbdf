"""ctypes binding of include/avoid_mpc_amd.h (the C ABI of libavoid_mpc_amd.so).

This is the only way Python reaches the hot path; there is no Python/torch/CPU fallback: if the
HIP library is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libavoid_mpc_amd.so")

AMK_OK, AMK_ERR_INVALID_ARG, AMK_ERR_HIP, AMK_ERR_NO_DEVICE, AMK_ERR_UNSUPPORTED, AMK_ERR_TIMEOUT = 0, 1, 2, 3, 4, 5   # include/avoid_mpc_amd.h
AMK_TIES_LOWEST_INDEX, AMK_TIES_NANOFLANN = 0, 1
AMK_EXACT_OFF, AMK_EXACT_IN_USE, AMK_EXACT_GAVE_UP, AMK_EXACT_TOO_DEEP = -1, 0, 1, 2   # amk_kd_exact_status
AMK_MAX_K = 64
AMK_MAX_QUERIES = 64
AMK_MAX_HORIZON = 32
AMK_MAX_OUTER_ITER = 8
AMK_MPC_DEFAULT_MAX_ITER = 100

# the functions of the CasADi plugin ABI and the helper every one of them carries (include/avoid_mpc_amd/casadi_plugin.h)
PLUGIN_FUNCTIONS = ["nlp", "nlp_f", "nlp_g", "nlp_grad_f", "nlp_jac_g", "nlp_hess_l", "nlp_grad"]
PLUGIN_HELPERS = ["alloc_mem", "init_mem", "free_mem", "checkout", "release", "incref", "decref", "n_in", "n_out",
                  "default_in", "name_in", "name_out", "sparsity_in", "sparsity_out", "work"]
PLUGIN_SYMBOLS = [f for f in PLUGIN_FUNCTIONS] + [f + "_" + h for f in PLUGIN_FUNCTIONS for h in PLUGIN_HELPERS] + \
                 ["amk_plugin_configure", "amk_plugin_dims"]

# every symbol include/avoid_mpc_amd.h declares (tests check the library exports all of them)
SYMBOLS = [
    "amk_version", "amk_status_string", "amk_last_hip_error", "amk_device_count",
    "amk_kd_create", "amk_kd_destroy", "amk_kd_build", "amk_kd_build_pair", "amk_kd_sizes", "amk_kd_search",
    "amk_kd_build_host", "amk_kd_search_host", "amk_kd_tie_flags", "amk_kd_set_tie_order", "amk_kd_exact_status", "amk_kd_exact_status_host", "amk_kd_keyframe_sweep",
    "amk_kd_keyframe_sweep_host", "amk_kd_points_host",
    "amk_mpc_create", "amk_mpc_destroy", "amk_mpc_horizon", "amk_mpc_nx", "amk_mpc_ref_len",
    "amk_mpc_setup_weights", "amk_mpc_setup_tau", "amk_mpc_setup_gains", "amk_mpc_set_drag_coefficient", "amk_mpc_set_drone_radius",
    "amk_kfmap_pool_bytes", "amk_kfmap_create", "amk_kfmap_destroy", "amk_kfmap_reset", "amk_kfmap_scenes", "amk_kfmap_frames", "amk_kfmap_twc", "amk_kfmap_add_vertex",
    "amk_kfmap_update", "amk_kfmap_step", "amk_kfmap_state_host",
    "amk_mpc_set_drone_accel_limits", "amk_mpc_set_solver_options", "amk_mpc_set_solve_budget", "amk_mpc_set_precision", "amk_mpc_solve",
    "amk_mpc_get_warm_start", "amk_mpc_set_warm_start", "amk_mpc_reset_warm_start",
    "amk_mpc_solve_host", "amk_mpc_ng", "amk_mpc_jac_nnz", "amk_mpc_hess_nnz", "amk_mpc_jac_sparsity",
    "amk_mpc_hess_sparsity", "amk_mpc_eval", "amk_mpc_eval_host", "amk_mpc_np", "amk_mpc_eval_gamma",
    "amk_mpc_eval_gamma_host", "amk_step_batch", "amk_step_batch_frames", "amk_step_batch_host",
    "amk_pipeline_create", "amk_pipeline_destroy", "amk_pipeline_slots", "amk_pipeline_gang", "amk_pipeline_mpc", "amk_pipeline_kd",
    "amk_pipeline_kfmap",
    "amk_pipeline_stream", "amk_pipeline_submit", "amk_pipeline_wait", "amk_pipeline_query", "amk_pipeline_drain",
    "amk_pipeline_outputs", "amk_pipeline_wait_stream",
    "amk_shard_scene_range", "amk_shard_unique_id", "amk_shard_create", "amk_shard_destroy", "amk_shard_rank",
    "amk_shard_world", "amk_shard_last_rccl_error", "amk_shard_gather", "amk_shard_gather_u", "amk_shard_max", "amk_shard_padded_count",
    "amk_shard_rccl_info", "amk_shard_wait",
    "amk_depth_out_size", "amk_depth_to_cloud", "amk_depth_to_cloud_host",
    "amk_depth_to_edge_cloud", "amk_depth_to_edge_cloud_host",
]


class AmkError(RuntimeError):
    pass


class StepParams(C.Structure):
    _fields_ = [("speed", C.c_double), ("safety_distance", C.c_double),
                ("mpc_max_iter", C.c_int), ("reserved", C.c_int)]


class DepthParams(C.Structure):
    """amk_depth_params (FrameKDMap's perception parameters, mpc_parameters.yaml:59-66)."""
    _fields_ = [("pixel2meter", C.c_double), ("depth_min", C.c_double), ("depth_max", C.c_double),
                ("resize_scale", C.c_double), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("Tbc", C.c_double * 16)]


class TaskParams(C.Structure):
    """amk_task_params"""
    _fields_ = [("decay", C.c_double), ("iter_time", C.c_double), ("farest_point", C.c_double), ("height", C.c_double),
                ("slow_down_kp", C.c_double), ("slow_down_kd", C.c_double), ("a_max_xy", C.c_double), ("a_max_z", C.c_double),
                ("use_odom_est", C.c_int), ("task", C.c_int)]


class KfmapParams(C.Structure):
    """amk_kfmap_params"""
    _fields_ = [("max_frame_count", C.c_int), ("keyframe_th_count", C.c_int), ("keyframe_th_dist", C.c_double), ("depth_min", C.c_double),
                ("Tbc", C.c_double * 16)]


class PipelineConfig(C.Structure):
    """amk_pipeline_config"""
    _fields_ = [("n_slots", C.c_int), ("n_scenes", C.c_int), ("max_points", C.c_int), ("max_edge_points", C.c_int),
                ("T", C.c_double), ("dt", C.c_double), ("nearest_point_num", C.c_int), ("queue_depth", C.c_int),
                ("gang", C.c_int), ("step", StepParams), ("task", TaskParams), ("depth", DepthParams), ("keyframes", KfmapParams)]


class PipelineFrame(C.Structure):
    """amk_pipeline_frame"""
    _fields_ = [("d_cloud", C.c_void_p), ("d_cloud_counts", C.c_void_p), ("d_edge", C.c_void_p), ("d_edge_counts", C.c_void_p),
                ("point_stride", C.c_int), ("keep_warm_start", C.c_int), ("d_state_quad", C.c_void_p), ("d_pos_x", C.c_void_p),
                ("d_ref_path_init", C.c_void_p), ("d_u_out", C.c_void_p), ("d_odom", C.c_void_p), ("odom_age", C.c_double),
                ("d_cmd_out", C.c_void_p), ("d_depth", C.c_void_p), ("depth_type", C.c_int), ("depth_rows", C.c_int), ("depth_cols", C.c_int),
                ("reserved", C.c_int), ("d_Twb", C.c_void_p), ("kf_obstacle", C.c_void_p), ("kf_edge", C.c_void_p), ("n_keyframes", C.c_int),
                ("reserved2", C.c_int), ("d_Twc_cur", C.c_void_p), ("camera", C.c_void_p), ("input_ready", C.c_void_p),
                ("d_global_goal", C.c_void_p)]


class FrameCamera(C.Structure):
    """amk_frame_camera: PtIsInFrame's camera model (FrameKDMap.cpp:215-231), intrinsics already divided by the resize scale."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("depth_max", C.c_double), ("width", C.c_int), ("height", C.c_int)]


AMK_DEPTH_U16, AMK_DEPTH_F32 = 0, 1

_lib = None


def load():
    """dlopen the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmkError(f"{LIB_PATH} is missing: build it with `python -m avoid_mpc_amd.build` "
                       "(hipcc, gfx950).  There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64; if the system
    # copy (which this library is linked against) is mapped first, the process ends up with two
    # runtimes and the second one sees no device.  Map torch's first when torch is installed; a
    # torch-free C++ host simply uses the system runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i, d, ll = C.c_void_p, C.c_int, C.c_double, C.c_longlong
    sig = {
        "amk_version": (i, []),
        "amk_status_string": (C.c_char_p, [i]),
        "amk_last_hip_error": (i, []),
        "amk_device_count": (i, []),
        "amk_kd_create": (i, [i, i, C.POINTER(vp)]),
        "amk_kd_destroy": (i, [vp]),
        "amk_kd_build": (i, [vp, vp, i, ll, vp, vp]),
        "amk_kd_build_pair": (i, [vp, vp, vp, vp, vp, vp, i, vp]),
        "amk_kd_sizes": (i, [vp, vp, vp]),
        "amk_kd_search": (i, [vp, vp, i, i, vp, vp, vp, vp, vp]),
        "amk_kd_tie_flags": (i, [vp, vp, i, i, i, vp, vp]),
        "amk_kd_set_tie_order": (i, [vp, i]),
        "amk_kd_exact_status": (i, [vp, vp, vp]),
        "amk_kd_exact_status_host": (i, [vp, vp]),
        "amk_kd_keyframe_sweep": (i, [vp, vp, d, i, vp, vp, vp]),
        "amk_kd_keyframe_sweep_host": (i, [vp, vp, d, i, vp, vp]),
        "amk_kd_points_host": (i, [vp, vp, vp]),
        "amk_kd_build_host": (i, [vp, vp, i, ll, vp]),
        "amk_kd_search_host": (i, [vp, vp, i, i, vp, vp, vp, vp]),
        "amk_mpc_create": (i, [d, d, i, i, C.POINTER(vp)]),
        "amk_mpc_destroy": (i, [vp]),
        "amk_mpc_horizon": (i, [vp]),
        "amk_mpc_nx": (i, [vp]),
        "amk_mpc_ref_len": (i, [vp]),
        "amk_mpc_setup_weights": (i, [vp, vp]),
        "amk_mpc_setup_tau": (i, [vp, vp]),
        "amk_mpc_setup_gains": (i, [vp, vp]),
        "amk_mpc_set_drag_coefficient": (i, [vp, d, d, d]),
        "amk_mpc_set_drone_radius": (i, [vp, d]),
        "amk_mpc_set_drone_accel_limits": (i, [vp, d, d, d, d]),
        "amk_mpc_set_solver_options": (i, [vp, d, i]),
        "amk_mpc_set_precision": (i, [vp, i]),
        "amk_mpc_solve": (i, [vp, vp, vp, vp, vp, i, vp]),
        "amk_mpc_get_warm_start": (i, [vp, vp, vp]),
        "amk_mpc_set_warm_start": (i, [vp, vp, vp]),
        "amk_mpc_reset_warm_start": (i, [vp, vp]),
        "amk_mpc_solve_host": (i, [vp, vp, vp, vp, vp, i]),
        "amk_mpc_ng": (i, [vp]),
        "amk_mpc_jac_nnz": (i, [vp]),
        "amk_mpc_hess_nnz": (i, [vp]),
        "amk_mpc_jac_sparsity": (i, [vp, vp, vp]),
        "amk_mpc_hess_sparsity": (i, [vp, vp, vp]),
        "amk_mpc_eval": (i, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "amk_mpc_eval_host": (i, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "amk_mpc_np": (i, [vp]),
        "amk_mpc_eval_gamma": (i, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "amk_mpc_eval_gamma_host": (i, [vp, vp, vp, vp, vp, vp, vp]),
        "amk_step_batch": (i, [vp, vp, vp, C.POINTER(StepParams), vp, vp, vp, vp, vp, vp, vp]),
        "amk_step_batch_host": (i, [vp, vp, vp, C.POINTER(StepParams), vp, vp, vp, vp, vp, vp]),
        "amk_step_batch_frames": (i, [vp, vp, i, vp, C.POINTER(FrameCamera), vp, C.POINTER(StepParams), vp, vp, vp, vp, vp,
                                      vp, vp]),
        "amk_pipeline_create": (i, [C.POINTER(PipelineConfig), C.POINTER(vp)]),
        "amk_pipeline_destroy": (i, [vp]),
        "amk_pipeline_slots": (i, [vp]),
        "amk_mpc_set_solve_budget": (i, [vp, i, i]),
        "amk_pipeline_gang": (i, [vp]),
        "amk_pipeline_mpc": (vp, [vp, i]),
        "amk_pipeline_kd": (vp, [vp, i, i]),
        "amk_pipeline_kfmap": (vp, [vp, i]),
        "amk_pipeline_stream": (vp, [vp, i]),
        "amk_pipeline_submit": (i, [vp, C.POINTER(PipelineFrame), C.POINTER(i)]),
        "amk_pipeline_wait": (i, [vp, i]),
        "amk_pipeline_wait_stream": (i, [vp, i, vp]),
        "amk__pipeline_inject_failure": (i, [vp, i]),   # internal (tests)
        "amk_pipeline_query": (i, [vp, i]),
        "amk_pipeline_drain": (i, [vp]),
        "amk_pipeline_outputs": (i, [vp, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "amk_kfmap_pool_bytes": (i, [i, i, i, i, C.POINTER(C.c_longlong)]),
        "amk_kfmap_create": (i, [i, i, i, C.POINTER(KfmapParams), C.POINTER(vp)]),
        "amk_kfmap_destroy": (i, [vp]),
        "amk_kfmap_reset": (i, [vp, i, i, vp]),
        "amk_kfmap_scenes": (i, [vp]),
        "amk_kfmap_frames": (i, [vp]),
        "amk_kfmap_twc": (vp, [vp]),
        "amk_kfmap_add_vertex": (i, [vp, i, i, vp, vp, vp, vp, i, vp, vp]),
        "amk_kfmap_update": (i, [vp, vp]),
        "amk_kfmap_step": (i, [vp, vp, vp, C.POINTER(StepParams), vp, vp, vp, vp, vp, vp, vp]),
        "amk_kfmap_state_host": (i, [vp, vp, vp, vp, vp]),
        "amk_shard_scene_range": (i, [i, i, i, C.POINTER(i), C.POINTER(i)]),
        "amk_shard_unique_id": (i, [C.c_char_p]),
        "amk_shard_create": (i, [C.c_char_p, i, i, C.POINTER(vp)]),
        "amk_shard_destroy": (i, [vp]),
        "amk_shard_rank": (i, [vp]),
        "amk_shard_world": (i, [vp]),
        "amk_shard_last_rccl_error": (i, []),
        "amk_shard_gather": (i, [vp, vp, ll, vp, vp]),
        "amk_shard_gather_u": (i, [vp, vp, i, vp, vp]),
        "amk_shard_max": (i, [vp, vp, i, vp]),
        "amk_shard_padded_count": (i, [i, i]),
        "amk_shard_rccl_info": (i, [C.c_char_p, i, C.POINTER(i), C.POINTER(i)]),
        "amk_shard_wait": (i, [vp, vp, d]),
        "amk_depth_out_size": (i, [i, i, d, C.POINTER(i), C.POINTER(i)]),
        "amk_depth_to_cloud": (i, [vp, i, i, i, C.c_longlong, i, C.POINTER(DepthParams), vp, vp, i, C.c_longlong, vp, vp]),
        "amk_depth_to_cloud_host": (i, [vp, i, i, i, C.c_longlong, i, C.POINTER(DepthParams), vp, vp, i, C.c_longlong, vp]),
        "amk_depth_to_edge_cloud": (i, [vp, i, i, i, C.c_longlong, i, C.POINTER(DepthParams), vp, vp, i, C.c_longlong, vp, vp]),
        "amk_depth_to_edge_cloud_host": (i, [vp, i, i, i, C.c_longlong, i, C.POINTER(DepthParams), vp, vp, i, C.c_longlong, vp]),
    }
    sig["amk__kd_set_mode"] = (i, [vp, i])  # internal: 0 bucketed index, 1 streaming scan
    sig["amk__sweep_set_target"] = (None, [i])  # internal (tests, A/B): the pool's sweep against 1 a fine hashed grid of the current frame (default), 0 the frame's own index
    sig["amk__sweep_set_order"] = (None, [i])   # internal (tests, A/B): 1 keyframe points in last sweep's grid order where it is theirs (default), 0 record order
    for name, (res, args) in sig.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue  # reported by check_symbols()
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def missing_symbols():
    lib = load()
    return [s for s in SYMBOLS if not hasattr(lib, s)]


def check(status, what=""):
    if status != AMK_OK:
        lib = load()
        msg = lib.amk_status_string(status).decode()
        raise AmkError(f"{what}: {msg} (status {status}, hipError {lib.amk_last_hip_error()})")


def hip_memcpy_dtoh(host_array, dev_ptr):
    """Synchronous device -> host copy of a raw device pointer into a numpy array (hipMemcpy through torch's HIP runtime);
    returns the hipError_t."""
    import torch
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.restype = C.c_int
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return hip.hipMemcpy(host_array.ctypes.data_as(C.c_void_p), C.c_void_p(dev_ptr), host_array.nbytes, 2)


def check_hip(err, what=""):
    if err != 0:
        raise AmkError(f"{what}: hipError {err}")


def dptr(t):
    """Device pointer of a torch tensor (None -> NULL).  Tensor must be contiguous and on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AmkError("expected a device tensor")
    if not t.is_contiguous():
        raise AmkError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


class _DevArray:
    """A raw device pointer dressed as a CUDA-array-interface object, so that torch can alias it (no copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tensor_from_ptr(ptr, shape, dtype):
    """torch tensor aliasing device memory owned by the library (the pointer must outlive the tensor)."""
    import torch
    typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=torch.device("cuda", torch.cuda.current_device()))


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)
