"""ctypes front-end of the C oracle (``pointnet2_oracle.c``) -- TEST INFRASTRUCTURE.

The function names and argument order mirror the reference's pybind surface
(``P2/_ext-src/src/bindings.cpp:6-19``) so the oracle can also stand in for
``pointnet2_ops._ext`` when the reference's Python layers are imported on CPU
(``oracle/gen_golden.py``).  All functions take/return torch CPU tensors
(float32 / int32, contiguous), exactly like the extension does on CUDA.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)
_i = ctypes.c_int


def build(force=False):
    """Compile the C oracle with gcc (strict IEEE flags, see Makefile)."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_opt_n_threads.argtypes = [_i]
        L.oracle_opt_n_threads.restype = _i
        L.oracle_furthest_point_sampling.argtypes = [_i, _i, _i, _F, _F, _I]
        L.oracle_gather_points.argtypes = [_i, _i, _i, _i, _F, _I, _F]
        L.oracle_gather_points_grad.argtypes = [_i, _i, _i, _i, _F, _I, _F]
        L.oracle_group_points.argtypes = [_i, _i, _i, _i, _i, _F, _I, _F]
        L.oracle_group_points_grad.argtypes = [_i, _i, _i, _i, _i, _F, _I, _F]
        L.oracle_ball_query.argtypes = [_i, _i, _i, ctypes.c_float, _i, _F, _F, _I]
        L.oracle_three_nn.argtypes = [_i, _i, _i, _F, _F, _F, _I]
        L.oracle_three_interpolate.argtypes = [_i, _i, _i, _i, _F, _I, _F, _F]
        L.oracle_three_interpolate_grad.argtypes = [_i, _i, _i, _i, _F, _I, _F, _F]
        L.oracle_knn_point.argtypes = [_i, _i, _i, _i, _F, _F, _I, _F]
        L.oracle_num_threads.argtypes = []
        L.oracle_num_threads.restype = _i
        L.oracle_set_num_threads.argtypes = [_i]
        L.oracle_set_num_threads.restype = None
        _lib = L
    return _lib


def _f(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu", \
        "oracle expects contiguous float32 CPU tensors"
    return ctypes.cast(t.data_ptr(), _F)


def _n(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu", \
        "oracle expects contiguous int32 CPU tensors"
    return ctypes.cast(t.data_ptr(), _I)


def num_threads():
    """Host threads the C oracle's FPS (over clouds) and knn (over queries) loops use."""
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def usable_host_cores():
    """Cores this process may actually use: the cgroup CPU quota when there is one (a container on a 256-thread host
    is typically given a share; 128 OpenMP threads on a 16-core quota run SLOWER than 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def opt_n_threads(work_size):
    return lib().oracle_opt_n_threads(int(work_size))


# ---- the nine functions of bindings.cpp:6-19 -------------------------------------------

def furthest_point_sampling(points, nsamples):
    """sampling.cpp:66-87 -- points (B,N,3) f32 -> (B,nsamples) i32."""
    B, N, _ = points.shape
    out = torch.zeros((B, nsamples), dtype=torch.int32)
    tmp = torch.full((B, N), 1e10, dtype=torch.float32)
    lib().oracle_furthest_point_sampling(B, N, nsamples, _f(points), _f(tmp), _n(out))
    return out


def gather_points(points, idx):
    """sampling.cpp:15-38 -- (B,C,N), (B,M) -> (B,C,M)."""
    B, C, N = points.shape
    M = idx.shape[1]
    out = torch.zeros((B, C, M), dtype=torch.float32)
    lib().oracle_gather_points(B, C, N, M, _f(points), _n(idx), _f(out))
    return out


def gather_points_grad(grad_out, idx, n):
    """sampling.cpp:40-65 -- (B,C,M), (B,M), n -> (B,C,n)."""
    B, C, M = grad_out.shape
    out = torch.zeros((B, C, n), dtype=torch.float32)
    lib().oracle_gather_points_grad(B, C, n, M, _f(grad_out), _n(idx), _f(out))
    return out


def group_points(points, idx):
    """group_points.cpp:12-36 -- (B,C,N), (B,S,K) -> (B,C,S,K)."""
    B, C, N = points.shape
    S, K = idx.shape[1], idx.shape[2]
    out = torch.zeros((B, C, S, K), dtype=torch.float32)
    lib().oracle_group_points(B, C, N, S, K, _f(points), _n(idx), _f(out))
    return out


def group_points_grad(grad_out, idx, n):
    """group_points.cpp:38-62 -- (B,C,S,K), (B,S,K), n -> (B,C,n)."""
    B, C, S, K = grad_out.shape
    out = torch.zeros((B, C, n), dtype=torch.float32)
    lib().oracle_group_points_grad(B, C, n, S, K, _f(grad_out), _n(idx), _f(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """ball_query.cpp:8-32 -- note the (new_xyz, xyz) order of the ext entry point."""
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    out = torch.zeros((B, M, nsample), dtype=torch.int32)
    lib().oracle_ball_query(B, N, M, float(radius), nsample, _f(new_xyz), _f(xyz), _n(out))
    return out


def three_nn(unknowns, knows):
    """interpolate.cpp:14-40 -- returns [dist2 (B,n,3) f32, idx (B,n,3) i32]."""
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    idx = torch.zeros((B, n, 3), dtype=torch.int32)
    dist2 = torch.zeros((B, n, 3), dtype=torch.float32)
    lib().oracle_three_nn(B, n, m, _f(unknowns), _f(knows), _f(dist2), _n(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """interpolate.cpp:42-70 -- (B,c,m), (B,n,3), (B,n,3) -> (B,c,n)."""
    B, c, m = points.shape
    n = idx.shape[1]
    out = torch.zeros((B, c, n), dtype=torch.float32)
    lib().oracle_three_interpolate(B, c, m, n, _f(points), _n(idx), _f(weight), _f(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """interpolate.cpp:71-99 -- (B,c,n), (B,n,3), (B,n,3), m -> (B,c,m)."""
    B, c, n = grad_out.shape
    out = torch.zeros((B, c, m), dtype=torch.float32)
    lib().oracle_three_interpolate_grad(B, c, n, m, _f(grad_out), _n(idx), _f(weight), _f(out))
    return out


# ---- knn_point (pytorch_utils.py:32-49) ---------------------------------------------------

def knn_point_with_dist(nsample, xyz, new_xyz):
    """Returns (dist (B,S,K) f32 ascending, idx (B,S,K) i32); ties -> lower index."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    assert nsample <= N, "knn_point: nsample must not exceed the number of points"
    idx = torch.zeros((B, S, nsample), dtype=torch.int32)
    dist = torch.zeros((B, S, nsample), dtype=torch.float32)
    lib().oracle_knn_point(B, N, S, nsample, _f(xyz.contiguous()), _f(new_xyz.contiguous()),
                           _n(idx), _f(dist))
    return dist, idx


def knn_point(nsample, xyz, new_xyz):
    """Same return convention as the reference (pytorch_utils.py:46-49): the first
    value is the index tensor again (a reference quirk every caller ignores)."""
    _, idx = knn_point_with_dist(nsample, xyz, new_xyz)
    return idx, idx
