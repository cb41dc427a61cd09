"""ctypes binding of libpwclo_hip.so (the C ABI declared in include/pwclo_ops.h).

There is deliberately no CPU or pure-PyTorch fallback: if the library is missing or a launch
fails, the call raises.  ``load()`` never builds anything; run
``python -m pwclonet_pylidarslam_amd.build`` (or ``__graft_entry__.build()``) first.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PWCLO_TRACE_LIB=1 (tools/wgtrace.py sets it) loads the developer variant built with the workgroup-trace hooks
LIB_PATH = os.path.join(_HERE, "lib", "libpwclo_hip_trace.so" if os.environ.get("PWCLO_TRACE_LIB", "0") != "0"
                        else "libpwclo_hip.so")

_F = ctypes.c_void_p  # device pointers travel as integers
_i = ctypes.c_int

# name -> argtypes, exactly the prototypes of include/pwclo_ops.h
SIGNATURES = {
    "pwclo_abi_version": ([], _i),
    "pwclo_set_stream": ([ctypes.c_void_p], None),
    "pwclo_get_stream": ([], ctypes.c_void_p),
    "pwclo_last_error": ([], _i),
    "pwclo_last_error_message": ([], ctypes.c_char_p),
    "pwclo_clear_error": ([], None),
    "pwclo_fps_large_cloud_launch": ([_i], None),
    "pwclo_fps_large_cloud_exchange": ([_i], None),
    "pwclo_trace_enable": ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint], None),
    "gather_points_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F], None),
    "gather_points_grad_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F], None),
    "furthest_point_sampling_kernel_wrapper": ([_i, _i, _i, _F, _F, _F], None),
    "group_points_kernel_wrapper": ([_i, _i, _i, _i, _i, _F, _F, _F], None),
    "group_points_grad_kernel_wrapper": ([_i, _i, _i, _i, _i, _F, _F, _F], None),
    "query_ball_point_kernel_wrapper": ([_i, _i, _i, ctypes.c_float, _i, _F, _F, _F], None),
    "three_nn_kernel_wrapper": ([_i, _i, _i, _F, _F, _F, _F], None),
    "three_interpolate_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F], None),
    "three_interpolate_grad_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F], None),
    "knn_point_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F], None),
    "knn_point_workspace_bytes": ([_i, _i], ctypes.c_longlong),
    "knn_point_ws_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F, _F], None),
    "knn_point_build_bytes": ([_i, _i], ctypes.c_longlong),
    "knn_point_slabs": ([_i], _i),
    "knn_build_kernel_wrapper": ([_i, _i, _F, _F, _F], None),
    "knn_point_prebuilt_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F], None),
    "knn_point_prebuilt_slice_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F, _i, _i], None),
    "furthest_point_sampling_sorted_kernel_wrapper": ([_i, _i, _i, _F, _F, _F, _F, _F, _F], None),
    "furthest_point_sampling_slab_kernel_wrapper": ([_i, _i, _i, _F, _F, _F, _F, _i, _F, _F, _F], None),
    "quat_warp_kernel_wrapper": ([_i, _i, _F, _F, _F, _F], None),
    "sa_fused_kernel_wrapper": ([_i] * 8 + [_F] * 6, None),
    "furthest_point_sampling_xyz_kernel_wrapper": ([_i, _i, _i, _F, _F, _F, _F], None),
    "furthest_point_sampling_chain_kernel_wrapper": ([_i, _i, _i, _F, _F, _F, _F, _F, _i, _F], None),
    "hamilton_product_kernel_wrapper": ([_i] * 6 + [_F] * 3, None),
    "quat_warp_pm_kernel_wrapper": ([_i, _i, _F, _F, _F, _F], None),
    "ingest_pairs_kernel_wrapper": ([_i, _i, _F, _F, _F], None),
    "ingest_frames_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F], None),
    "kitti_transform_filter_kernel_wrapper": ([_i, _F, _F, _F, _F], None),
    "kitti360_filter_kernel_wrapper": ([_i, ctypes.c_float, ctypes.c_float, _F, _F, _F], None),
    "compact_frames_kernel_wrapper": ([_i, _i, _i, _F, _F, _F, _F, _F], None),
    "fps_spatial_order_workspace_bytes": ([_i, _i], ctypes.c_longlong),
    "fps_spatial_order_kernel_wrapper": ([_i, _i, _F, _F, _F, _F], None),
    "softmax_wsum_supported_k": ([_i], _i),
    "softmax_wsum_forward_kernel_wrapper": ([ctypes.c_longlong, _i, _F, _F, _F], None),
    "softmax_wsum_backward_kernel_wrapper": ([ctypes.c_longlong, _i, _F, _F, _F, _F, _F], None),
    "compact_frames_scan_kernel_wrapper": ([_i, _i, _i, _F, _F, _F, _F], None),
    "batchnorm_train_workspace_bytes": ([_i], ctypes.c_longlong),
    "batchnorm_train_forward_kernel_wrapper": ([_i, _i, _i, _F, _F, _F, ctypes.c_float, ctypes.c_float, _F, _F, _F, _F,
                                                _F, _F, _i], None),
    "batchnorm_train_backward_kernel_wrapper": ([_i, _i, _i] + [_F] * 10 + [_i], None),
    "batchnorm_train_relu_maxk_forward_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, ctypes.c_float, ctypes.c_float]
                                                         + [_F] * 8, None),
    "batchnorm_train_relu_maxk_backward_kernel_wrapper": ([_i, _i, _i, _i] + [_F] * 12, None),
    "conv1x1_forward_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _i, _F], None),
    "conv1x1_affine_forward_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F, _i, _F], None),
    "conv1x1_affine_maxk_forward_kernel_wrapper": ([_i, _i, _i, _i, _i, _F, _F, _F, _F, _i, _F], None),
    "conv1x1_bnrelu_forward_kernel_wrapper": ([_i, _i, _i, _i] + [_F] * 7, None),
    "conv1x1_bnrelu_wgrad_kernel_wrapper": ([_i, _i, _i, _i] + [_F] * 8, None),
    "conv1x1_wgrad_workspace_bytes": ([_i, _i, _i, _i], ctypes.c_longlong),
    "conv1x1_stats_workspace_bytes": ([_i, _i, _i, _i], ctypes.c_longlong),
    "conv1x1_dgrad_bnstats_kernel_wrapper": ([_i, _i, _i, _i] + [_F] * 11, None),
    "batchnorm_train_backward_apply_kernel_wrapper": ([_i, _i, _i] + [_F] * 9 + [_i], None),
    "conv1x1_forward_bnstats_kernel_wrapper": ([_i, _i, _i, _i] + [_F] * 7 + [ctypes.c_float, ctypes.c_float] + [_F] * 5, None),
    "batchnorm_train_apply_kernel_wrapper": ([_i, _i, _i] + [_F] * 6 + [_i], None),
    "batchnorm_train_relu_maxk_apply_kernel_wrapper": ([_i, _i, _i, _i] + [_F] * 8, None),
    "conv1x1_wgrad_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F], None),
    "group_points_grad_sorted_kernel_wrapper": ([_i, _i, _i, _i, _i, _F, _F, _F, _F], None),
    "xyz_diff_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F, ctypes.c_longlong], None),
    "geometry_encode_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F, ctypes.c_longlong], None),
    "geometry_encode_grad_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, _F, _F, ctypes.c_longlong, _F, _F, _F], None),
    "broadcast_centre_kernel_wrapper": ([_i, _i, _i, _i, _F, _F, ctypes.c_longlong], None),
    "broadcast_centre_grad_kernel_wrapper": ([_i, _i, _i, _i, _F, ctypes.c_longlong, _F], None),
    "group_points_strided_kernel_wrapper": ([_i, _i, _i, _i, _i, _F, _F, _F, ctypes.c_longlong], None),
    "group_points_grad_strided_kernel_wrapper": ([_i, _i, _i, _i, _i, _F, ctypes.c_longlong, _F, _F], None),
    "group_points_grad_sorted_strided_kernel_wrapper": ([_i, _i, _i, _i, _i, _F, ctypes.c_longlong, _F, _F, _F], None),
    "upconv_fused_kernel_wrapper": ([_i] * 4 + [_F] * 6, None),
    "pointwise_fused_kernel_wrapper": ([_i] * 7 + [_F] * 5, None),
    "pointwise_tail_fused_kernel_wrapper": ([_i] * 8 + [_F] * 7 + [_i], None),
    "cv_fused_a1_kernel_wrapper": ([_i] * 5 + [_F] * 7 + [_i], None),
    "cv_fused_a2_kernel_wrapper": ([_i] * 4 + [_F] * 6 + [_i] * 3, None),
    "cv_fused_b_kernel_wrapper": ([_i] * 4 + [_F] * 6, None),
    "masked_pool_kernel_wrapper": ([_i, _i, _F, _F, _F], None),
    "pose_head_fused_kernel_wrapper": ([_i, _i] + [_F] * 13 + [_i], None),
    "pose_head_warp_fused_kernel_wrapper": ([_i, _i] + [_F] * 13 + [_i, _i, _F, _F], None),
    "linear_jobs_kernel_wrapper": ([_i] + [ctypes.POINTER(ctypes.c_int)] * 3 + [ctypes.POINTER(ctypes.c_void_p)] * 3
                                   + [ctypes.POINTER(ctypes.c_int)], None),
    "sa_fused_h_kernel_wrapper": ([_i] * 7 + [_F] * 6 + [_i] * 3, None),
    "upconv_fused_h_kernel_wrapper": ([_i] * 4 + [_F] * 6 + [_i] * 2, None),
    "upconv_post_fused_h_kernel_wrapper": ([_i] * 6 + [_F] * 4 + [ctypes.POINTER(ctypes.c_void_p)] * 4 + [_i] * 2, None),
    "cv_fused_a1_h_kernel_wrapper": ([_i] * 4 + [_F] * 7 + [_i] * 3, None),
    "cv_fused_a_lane6_kernel_wrapper": ([_i] * 3 + [_F] * 10 + [_i] * 3, None),
    "cv_fused_b_h_kernel_wrapper": ([_i] * 3 + [_F] * 7 + [_i] * 2, None),
    "odom_rows_to_transforms_kernel_wrapper": ([_i, _i, _F, _F, _i], None),
    "odom_accumulate_kernel_wrapper": ([_i, _F, _F, _F], None),
    "odom_cumulative_distance_kernel_wrapper": ([_i, _F, _F, _F], None),
    "odom_sequence_errors_kernel_wrapper": ([_i, _i] + [_F] * 5 + [_i, _i] + [_F] * 3, None),
}

_lib = None


def load():
    """Load the shared library and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libpwclo_hip.so is not built (%s missing). Build it with "
                "`python -m pwclonet_pylidarslam_amd.build`; there is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header and library out of sync
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = lib
    return _lib


def check(what):
    """Raise if the previous launch recorded an error (the reference would exit(-1))."""
    lib = _lib
    code = lib.pwclo_last_error()
    if code != 0:
        msg = lib.pwclo_last_error_message().decode("utf-8", "replace")
        lib.pwclo_clear_error()
        raise RuntimeError("%s failed (error %d): %s" % (what, code, msg))


def synchronize(device=None):
    """Wait for the device and raise if a kernel that has run reported a device-side failure (e.g. the
    cooperative large-cloud sampler timing out: PWCLO_ECOOP_TIMEOUT).  Launches are asynchronous, so such a
    failure surfaces at the next library call made after the kernel ran, or here."""
    load()
    torch.cuda.synchronize(device)
    check("device-side check after synchronize")


# Optional launch profiler (bench.py / tools): an object with .add(name, meta, start_evt, end_evt).
# Wrappers describe the next launch's algorithmic work with annotate(); both are no-ops otherwise.
profiler = None
_meta = None


def annotate(**meta):
    """Algorithmic work of the NEXT launch (flops / bytes / units), consumed by the profiler."""
    global _meta
    _meta = meta


_fn_cache = {}


def call(name, device, *args):
    """Launch `name` on torch's current stream of `device` and check the sticky error.
    (The host cost of this function is the floor of every eager launch: function pointers are looked up once, the device
    guard is skipped when `device` already is the current one.)"""
    global _meta
    fn = _fn_cache.get(name)
    if fn is None:
        lib = load()
        fn = _fn_cache[name] = (getattr(lib, name), lib.pwclo_set_stream, lib.pwclo_last_error)
    launch, set_stream, last_error = fn
    idx = device.index if isinstance(device, torch.device) and device.index is not None else None
    if idx is None or idx == torch.cuda.current_device():
        stream = torch.cuda.current_stream()
        set_stream(stream.cuda_stream)
        if profiler is None:
            launch(*args)
        else:   # HIP events on the very stream the kernel is launched on
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            launch(*args)
            e.record(stream)
            profiler.add(name, _meta, s, e)
    else:
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device)
            set_stream(stream.cuda_stream)
            if profiler is None:
                launch(*args)
            else:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(stream)
                launch(*args)
                e.record(stream)
                profiler.add(name, _meta, s, e)
    _meta = None
    if last_error() != 0:
        check(name)
