"""On-device KITTI / KITTI-360 frame preprocessing.

The reference does this in NumPy inside DataLoader workers (``slam/dataset/kitti_odometry_dataset.py``:
``__getitem__`` :375-397 and ``filter_pcd`` :149-172): calibration transform of the raw velodyne
points, ground / range filter, then a random choice of ``npoints`` survivors (with replacement when
there are too few).  Here the transform + filter is one HIP kernel on the raw ``(n, 4)`` frame already
in HBM; compaction and the random choice are torch index ops (the reference's NumPy RNG stream is not
reproducible on the device, so only the deterministic part is bit-comparable), and ``sample="fps"``
replaces the random choice by furthest point sampling (BASELINE.json configs[4]; clouds above
24 576 points use the cooperative multi-workgroup sampler).

KITTI-360 (``slam/dataset/kitti_360_dataset_2.py``: ``filter_pcd`` :113-135) keeps the sensor frame: no
transform, ground = ``z < -(1.73 - 0.3)``, range test on x and y (``kitti360_filter``).  Both filters take
a whole batch of equally long frames ``(b, n, 4)``; ``compact`` packs every frame's survivors to the front
of a zero-padded ``(b, cap, 3)`` batch (zero rows are never selected by the sampler, sampling_gpu.cu:100-101)
and ``frames_to_clouds`` chains filter -> compaction -> one batched furthest-point-sampling call.
"""
import torch

from . import _lib
from .pointnet2_ops import _ext


KITTI360_GROUND_Z = -(1.73 - 0.3)      # -(VELODYNE_HEIGHT - WHEEL_AXIS_HEIGHT), slam/common/kitti360_utils.py:24-27


def _check_frames(points):
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    assert points.dim() in (2, 3) and points.size(-1) == 4 and points.dtype == torch.float32
    return points.contiguous()


def transform_filter(points, tr):
    """points (n,4) or (b,n,4) f32 cuda raw frames sharing one calibration, tr (3,4) or (4,4) array-like
    -> (xyz (...,3) f32, keep (...) i32)."""
    points = _check_frames(points)
    lead = points.shape[:-1]
    n = points.numel() // 4
    tr = torch.as_tensor(tr, dtype=torch.float64).reshape(-1)[:12].contiguous().to(points.device)
    xyz = torch.empty(lead + (3,), dtype=torch.float32, device=points.device)
    keep = torch.empty(lead, dtype=torch.int32, device=points.device)
    _lib.call("kitti_transform_filter_kernel_wrapper", points.device, n, tr.data_ptr(), points.data_ptr(),
              xyz.data_ptr(), keep.data_ptr())
    return xyz, keep


def kitti360_filter(points, near_threshold, ground_z=KITTI360_GROUND_Z):
    """points (n,4) or (b,n,4) f32 cuda raw KITTI-360 frames -> (xyz (...,3) f32, keep (...) i32)."""
    points = _check_frames(points)
    lead = points.shape[:-1]
    xyz = torch.empty(lead + (3,), dtype=torch.float32, device=points.device)
    keep = torch.empty(lead, dtype=torch.int32, device=points.device)
    _lib.call("kitti360_filter_kernel_wrapper", points.device, points.numel() // 4, float(ground_z),
              float(near_threshold), points.data_ptr(), xyz.data_ptr(), keep.data_ptr())
    return xyz, keep


def compact(xyz, keep, cap=None):
    """xyz (b,n,3) f32, keep (b,n) i32 -> (packed (b,cap,3) f32: survivors in frame order, then zero rows;
    counts (b,) i32).  ``cap`` defaults to n (no host sync); survivors beyond cap are dropped."""
    assert xyz.dim() == 3 and keep.shape == xyz.shape[:2] and keep.dtype == torch.int32
    b, n, _ = xyz.shape
    cap = n if cap is None else int(cap)
    xyz = xyz.contiguous()
    keep = keep.contiguous()
    out = torch.zeros((b, cap, 3), dtype=torch.float32, device=xyz.device)
    counts = torch.empty((b,), dtype=torch.int32, device=xyz.device)
    # scan + scatter in one hand-written kernel (csrc/warp.hip: compact_frames_scan_kernel; the scan used to be torch.cumsum)
    _lib.call("compact_frames_scan_kernel_wrapper", xyz.device, b, n, cap, keep.data_ptr(), xyz.data_ptr(),
              out.data_ptr(), counts.data_ptr())
    return out, counts


def frames_to_clouds(points, npoints, dataset="kitti", tr=None, near_threshold=30.0, cap=None):
    """A batch of raw frames (b,n,4) -> (clouds (b,npoints,3) f32, counts (b,) i32), deterministic
    (furthest point sampling of each frame's survivors; BASELINE.json configs[4]).  A frame with fewer
    than ``npoints`` survivors gets index-0 repeats past its count exactly as the reference's sampler
    returns them for m > #valid; check ``counts`` when that matters (the dataset's own rule for that case
    is a random draw with replacement, see ``kitti_frame_to_cloud``)."""
    assert points.dim() == 3
    if dataset == "kitti":
        xyz, keep = transform_filter(points, tr)
    elif dataset == "kitti360":
        xyz, keep = kitti360_filter(points, near_threshold)
    else:
        raise ValueError(f"unknown dataset {dataset!r}")
    packed, counts = compact(xyz, keep, cap)
    idx = _ext.furthest_point_sampling(packed, npoints)
    clouds = torch.gather(packed, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    return clouds, counts


def kitti_frame_to_cloud(points, tr, npoints, sample="random", generator=None):
    """One raw frame -> (npoints, 3) f32 cloud in the camera frame, as ``KITTIOdometry.filter_pcd`` returns it.
    ``sample``: "random" (the reference's rule, torch RNG) or "fps" (deterministic)."""
    xyz, keep = transform_filter(points, tr)
    indices = keep.nonzero(as_tuple=False).flatten()
    cnt = indices.numel()
    dev = xyz.device
    if sample == "fps":
        cand = xyz[indices] if cnt > 0 else xyz
        if cand.size(0) >= npoints:
            sel = _ext.furthest_point_sampling(cand.unsqueeze(0).contiguous(), npoints)[0].long()
            return cand[sel]
        extra = torch.randint(cand.size(0), (npoints - cand.size(0),), device=dev, generator=generator)
        return torch.cat((cand, cand[extra]))
    if cnt >= npoints:                                   # np.random.choice(indices, npoints, replace=False)
        sel = indices[torch.randperm(cnt, device=dev, generator=generator)[:npoints]]
    elif cnt > 0:                                        # all survivors + a draw with replacement
        sel = torch.cat((indices, indices[torch.randint(cnt, (npoints - cnt,), device=dev, generator=generator)]))
    else:                                                # empty: random over the whole frame (the reference warns)
        sel = torch.randint(xyz.size(0), (npoints,), device=dev, generator=generator)
    return xyz[sel]
