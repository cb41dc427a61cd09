"""NumPy restatement of the reference's odometry evaluation of predicted poses -- TEST INFRASTRUCTURE.

Follows, function by function (R = /root/reference):
  * ``quat2mat``                          R/train.py:762-795
  * ``rows_to_relative``                  R/train.py:866-893 (kitti_odometry branch: T = [[R t],[0 1]], stored inverted)
  * ``convert_to_absolute``               R/slam/common/kitti360_utils.py:406-431 (dict branch, velo_to_pose=False)
  * ``shift_poses`` .. ``compute_are``    R/slam/eval/eval_odometry.py:236-427
  * ``load_poses_text`` / ``kitti_odom_eval``  R/evaluation.py:161-290, 578-630, 644-722 (kittiOdomEval without plots),
    fed through the same text round trip as R/train.py:945-960 (``np.savetxt(fmt='%.08f')`` of the 3x4 rows)
Pinned by ``oracle/gen_eval_golden.py``: the imported reference functions on synthetic trajectories ->
``tests/golden/eval_cases.npz`` (tests/test_oracle_cpu.py compares this file's functions with those values).
Only tests import this module.
"""
import numpy as np

DEFAULT_SEGMENTS = [100, 200, 300, 400, 500, 600, 700, 800]


def quat2mat(q):
    w, x, y, z = q
    Nq = w * w + x * x + y * y + z * z
    if Nq < 1e-8:
        return np.eye(3)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY],
                     [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
                     [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def rows_to_relative(rows):
    """rows (n,7) [t, q(w,x,y,z)] -> list of the reference's "relative poses" inv([[R t],[0 0 0 1]])."""
    out = []
    filler = np.array([[0.0, 0.0, 0.0, 1.0]])
    for r in rows:
        R = quat2mat(r[3:].reshape(4))
        T = np.concatenate([np.concatenate([R, r[:3].reshape(3, 1)], axis=-1), filler], axis=0)
        out.append(np.linalg.inv(T))
    return out


def convert_to_absolute(relative_poses):
    """dict frame -> 4x4 relative pose; abs[f] = inv(rel[f] @ inv(abs[prev])), abs[-1] = I."""
    prev = np.eye(4)
    out = {}
    for f in sorted(relative_poses.keys()):
        out[f] = np.linalg.inv(relative_poses[f] @ np.linalg.inv(prev))
        prev = out[f]
    return out


def shift_poses(poses):
    return np.concatenate([np.expand_dims(np.eye(4), axis=0), poses[:-1, :4, :4]], axis=0)


def compute_relative_poses(poses):
    return np.linalg.inv(shift_poses(poses)) @ poses


def compute_absolute_poses(relative_poses):
    absolute = relative_poses.copy()
    for i in range(absolute.shape[0] - 1):
        absolute[i + 1, :] = np.dot(absolute[i].copy(), relative_poses[i + 1].copy())
    return absolute


def compute_cumulative_trajectory_length(trajectory):
    shifted = shift_poses(trajectory)
    return np.cumsum(np.linalg.norm(shifted[:, :3, 3] - trajectory[:, :3, 3], axis=1))


def rotation_error(pose_err):
    d = 0.5 * (pose_err[0, 0] + pose_err[1, 1] + pose_err[2, 2] - 1.0)
    return np.arccos(max(min(d, 1.0), -1.0))


def translation_error(pose_err):
    return np.sqrt(pose_err[0, 3] ** 2 + pose_err[1, 3] ** 2 + pose_err[2, 3] ** 2)


def last_frame_from_segment_length(dist, first_frame, segment):
    for i in range(first_frame, len(dist)):
        if dist[i] > dist[first_frame] + segment:
            return i
    return -1


def calc_sequence_errors_arrays(trajectory, ground_truth, segments=DEFAULT_SEGMENTS, step_size=10):
    """eval_odometry.py:316-361 on (n,4,4) arrays (note: ITS distance starts with |p0 - 0|, the identity-shifted
    first entry).  Rows [first, r_err/len, t_err/len, len, speed, last]."""
    dist = compute_cumulative_trajectory_length(ground_truth)
    rows = []
    for first in range(0, ground_truth.shape[0], step_size):
        for seg in segments:
            last = last_frame_from_segment_length(dist, first, seg)
            if last == -1:
                continue
            dg = np.linalg.inv(ground_truth[first]).dot(ground_truth[last])
            dt = np.linalg.inv(trajectory[first]).dot(trajectory[last])
            e = np.linalg.inv(dt).dot(dg)
            rows.append([first, rotation_error(e) / seg, translation_error(e) / seg, seg,
                         seg / (0.1 * (last - first + 1)), last])
    return np.array(rows, dtype=np.float64).reshape(-1, 6)


def compute_ate(relative_predicted, relative_ground_truth):
    tr_err = np.linalg.norm(relative_predicted[:, :3, 3] - relative_ground_truth[:, :3, 3], axis=1)
    ate = tr_err.mean()
    return ate, np.sqrt(np.power(tr_err - ate, 2).mean())


def compute_are(relative_trajectory, relative_ground_truth):
    diff = np.linalg.inv(relative_ground_truth[:, :3, :3]) @ relative_trajectory[:, :3, :3] - np.eye(3)
    r_err = np.linalg.norm(diff, axis=(1, 2))
    are = r_err.mean()
    return are, np.sqrt(np.power(r_err - are, 2).mean())


# ---- kittiOdomEval (evaluation.py) ------------------------------------------------------------------------------

def text_round_trip(poses_3x4):
    """np.savetxt(fmt='%.08f') then float(): what train.py:945-960 writes and evaluation.py:161-196 reads."""
    return np.array([[float("%.08f" % v) for v in row] for row in np.asarray(poses_3x4).reshape(-1, 12)])


def load_poses_rows(rows12, relative):
    """evaluation.py:161-196 on already-parsed (n,12) rows (no frame index column, toCameraCoord=False)."""
    poses = {}
    for cnt, row in enumerate(rows12):
        P = np.eye(4)
        P[:3, :4] = np.asarray(row, dtype=np.float64).reshape(3, 4)
        if relative and cnt > 0:
            P = np.linalg.inv(poses[0]) @ P          # NOTE: frame 0 itself is stored un-rebased (reference quirk)
        poses[cnt] = P
    return poses


def trajectory_distances(poses):
    dist = [0]
    keys = sorted(poses.keys())
    for i in range(len(keys) - 1):
        P1, P2 = poses[keys[i]], poses[keys[i + 1]]
        dx, dy, dz = P1[0, 3] - P2[0, 3], P1[1, 3] - P2[1, 3], P1[2, 3] - P2[2, 3]
        dist.append(dist[i] + np.sqrt(dx ** 2 + dy ** 2 + dz ** 2))
    return dist


def calc_sequence_errors(poses_gt, poses_result, lengths=DEFAULT_SEGMENTS, step_size=10):
    """evaluation.py:236-271.  Rows [first_frame, r_err/len, t_err/len, len, speed]."""
    err = []
    dist = trajectory_distances(poses_gt)
    for first in range(0, len(poses_gt), step_size):
        for len_ in lengths:
            last = last_frame_from_segment_length(dist, first, len_)
            if last == -1 or last not in poses_result or first not in poses_result:
                continue
            dg = np.dot(np.linalg.inv(poses_gt[first]), poses_gt[last])
            dr = np.dot(np.linalg.inv(poses_result[first]), poses_result[last])
            e = np.dot(np.linalg.inv(dr), dg)
            num_frames = last - first + 1.0
            err.append([first, rotation_error(e) / len_, translation_error(e) / len_, len_, len_ / (0.1 * num_frames)])
    return err


def compute_overall_err(seq_err):
    t = sum(e[2] for e in seq_err) / len(seq_err)
    r = sum(e[1] for e in seq_err) / len(seq_err)
    return t, r


def compute_segment_err(seq_err, lengths=DEFAULT_SEGMENTS):
    out = {}
    for len_ in lengths:
        rows = [(e[2], e[1]) for e in seq_err if e[3] == len_]
        out[len_] = [float(np.mean([r[0] for r in rows])), float(np.mean([r[1] for r in rows]))] if rows else []
    return out


def compute_speed_err(seq_err):
    out = {}
    for key in range(2, 25, 2):
        rows = [(e[2], e[1]) for e in seq_err if np.abs(e[4] - key) < 2.0]
        out[key] = [float(np.mean([r[0] for r in rows])), float(np.mean([r[1] for r in rows]))] if rows else []
    return out


def kitti_odom_eval(pred_rows, gt_rows):
    """The numeric part of train.py:866-990 + kittiOdomEval.eval for ONE sequence: pose rows (n,7) of the prediction
    and of the ground truth -> dict(abs_pred, abs_gt, seq_err, ave_t_err, ave_r_err, segment, speed)."""
    rel_p = {i: m for i, m in enumerate(rows_to_relative(np.asarray(pred_rows)))}
    rel_g = {i: m for i, m in enumerate(rows_to_relative(np.asarray(gt_rows)))}
    abs_p, abs_g = convert_to_absolute(rel_p), convert_to_absolute(rel_g)
    frames = sorted(abs_p)
    p12 = text_round_trip([abs_p[f][:3, :].reshape(12) for f in frames])
    g12 = text_round_trip([abs_g[f][:3, :].reshape(12) for f in frames])
    poses_result = load_poses_rows(p12, relative=True)
    poses_gt = load_poses_rows(g12, relative=False)
    seq_err = calc_sequence_errors(poses_gt, poses_result)
    t, r = compute_overall_err(seq_err) if seq_err else (None, None)
    return dict(abs_pred=np.stack([abs_p[f] for f in frames]), abs_gt=np.stack([abs_g[f] for f in frames]),
                seq_err=np.array(seq_err, dtype=np.float64).reshape(-1, 5), ave_t_err=t, ave_r_err=r,
                segment=compute_segment_err(seq_err), speed=compute_speed_err(seq_err))
