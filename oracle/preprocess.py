"""NumPy restatement of the reference's KITTI frame preprocessing -- TEST INFRASTRUCTURE.

Follows slam/dataset/kitti_odometry_dataset.py:375-397 (homogeneous transform with the 4x4 ``Tr``, float64
because ``np.ones`` promotes the float32 points) and ``filter_pcd`` :149-160 (ground / range mask).
``kitti360_filter`` follows slam/dataset/kitti_360_dataset_2.py:113-123 (sensor frame, no transform).
Parity unpinned by execution: the dataset classes need the KITTI files to be instantiated, so this is a
line-by-line restatement checked only against hand-computed values.
"""
import numpy as np


def transform_filter(points_n4, tr):
    tr = np.asarray(tr, dtype=np.float64).reshape(-1)[:12].reshape(3, 4)
    tr4 = np.vstack((tr, np.array([0, 0, 0, 1.0])))
    p = np.asarray(points_n4)[:, :3]                                  # float32
    p = np.concatenate([p, np.ones((p.shape[0], 1))], axis=-1)        # float64 (n,4)
    q = np.matmul(tr4, p.T).T[:, :3]                                  # float64 (n,3)
    is_ground = q[:, 1] > 1.1
    near_x = np.logical_and(q[:, 0] < 30, q[:, 0] > -30)
    near_z = np.logical_and(q[:, 2] < 30, q[:, 2] > -30)
    keep = np.logical_and(np.logical_not(is_ground), np.logical_and(near_x, near_z))
    return q, keep


def kitti360_filter(points_n4, near_threshold, velodyne_height=1.73, wheel_axis_height=0.3):
    points = np.asarray(points_n4)[:, :3]                             # float32, compared with Python scalars
    wheel_axis_z = -(velodyne_height - wheel_axis_height)
    is_ground = points[:, 2] < wheel_axis_z
    not_ground = np.logical_not(is_ground)
    near_x = np.logical_and(points[:, 0] < near_threshold, points[:, 0] > -near_threshold)
    near_y = np.logical_and(points[:, 1] < near_threshold, points[:, 1] > -near_threshold)
    return points, np.logical_and(not_ground, np.logical_and(near_x, near_y))
