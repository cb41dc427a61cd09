"""On-device KITTI frame preprocessing (first slice).

The reference does this in NumPy inside DataLoader workers (``slam/dataset/kitti_odometry_dataset.py``:
``__getitem__`` :375-397 and ``filter_pcd`` :149-172): calibration transform of the raw velodyne
points, ground / range filter, then a random choice of ``npoints`` survivors (with replacement when
there are too few).  Here the transform + filter is one HIP kernel on the raw ``(n, 4)`` frame already
in HBM; compaction and the random choice are torch index ops (the reference's NumPy RNG stream is not
reproducible on the device, so only the deterministic part is bit-comparable), and ``sample="fps"``
replaces the random choice by furthest point sampling (BASELINE.json configs[4]; clouds above
24 576 points use the cooperative multi-workgroup sampler).
"""
import torch

from . import _lib
from .pointnet2_ops import _ext


def transform_filter(points, tr):
    """points (n,4) f32 cuda raw frame, tr (3,4) or (4,4) array-like -> (xyz (n,3) f32, keep (n,) i32)."""
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    assert points.dim() == 2 and points.size(1) == 4 and points.dtype == torch.float32
    points = points.contiguous()
    n = points.size(0)
    tr = torch.as_tensor(tr, dtype=torch.float64).reshape(-1)[:12].contiguous().to(points.device)
    xyz = torch.empty((n, 3), dtype=torch.float32, device=points.device)
    keep = torch.empty((n,), dtype=torch.int32, device=points.device)
    _lib.call("kitti_transform_filter_kernel_wrapper", points.device, n, tr.data_ptr(), points.data_ptr(),
              xyz.data_ptr(), keep.data_ptr())
    return xyz, keep


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
