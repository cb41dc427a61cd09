"""KITTI odometry evaluation of predicted poses, on the device.

Host-side mirror of the reference's evaluation path (R = /root/reference):
  * the accumulation loop of ``PWCLONetTrainer.test_model`` (R/train.py:866-893): level-1 pose rows
    ``[t, q]`` of every batch -> 4x4 "relative poses" per (sequence, frame);
  * ``KITTI360_TRANSFORMATIONS.convert_to_absolute`` (R/slam/common/kitti360_utils.py:406-431);
  * the array helpers of R/slam/eval/eval_odometry.py:236-427 under their own names
    (``compute_relative_poses``, ``compute_absolute_poses``, ``compute_cumulative_trajectory_length``,
    ``calcSequenceErrors``, ``compute_kitti_metrics``, ``compute_ate``, ``compute_are``);
  * the numeric part of ``kittiOdomEval`` (R/evaluation.py:161-290, 578-630): t_rel / r_rel over the 100..800 m
    segments, per-length and per-speed averages -- including its text round trip (R/train.py:945-960 writes
    ``%.08f``) and its ``loadPoses(relative=True)`` quirk (frame 0 of a prediction is stored un-rebased).
Plots, file layout and the evo calls of the reference are out of scope.

What runs where: pose rows never leave the device between the network and the metrics.  ``csrc/odometry_eval.hip``
turns rows into fp64 transforms, composes every sequence's trajectory with a wave-level scan, accumulates the
travelled distance and evaluates all (first frame, segment length) pairs of all sequences in one launch; the means
are torch reductions.  There is no CPU path: host tensors raise like every other entry point of the package.
"""
import numpy as np
import torch

from . import _lib

DEFAULT_SEGMENTS = (100, 200, 300, 400, 500, 600, 700, 800)       # R/evaluation.py:108, eval_odometry.py:313


def _gpu64(t, name):
    if not t.is_cuda:
        raise RuntimeError("CPU not supported")
    if t.dtype != torch.float64 or not t.is_contiguous():
        raise RuntimeError(name + " must be a contiguous float64 tensor")


def _starts(lengths, device):
    s = [0]
    for n in lengths:
        s.append(s[-1] + int(n))
    return torch.tensor(s, dtype=torch.int32, device=device), s


# ---- kernels behind the mirrors -------------------------------------------------------------------------------------

def rows_to_transforms(rows, invert=False):
    """rows (n,7) or a strided view such as ``pose_params[:, 0, :]`` of (B,4,7): [tx ty tz qw qx qy qz] fp32 ->
    (n,4,4) fp64 [[R(q) t],[0 0 0 1]] with the reference's quat2mat (R/train.py:762-795: valid for non-unit
    quaternions, identity below 1e-8); ``invert=True`` gives the matrices the reference stores (:878)."""
    if not rows.is_cuda:
        raise RuntimeError("CPU not supported")
    if rows.dtype != torch.float32 or rows.dim() != 2 or rows.shape[1] != 7 or rows.stride(1) != 1:
        raise RuntimeError("rows must be float32 (n,7) with unit inner stride")
    n = rows.shape[0]
    T = torch.empty((n, 4, 4), dtype=torch.float64, device=rows.device)
    _lib.call("odom_rows_to_transforms_kernel_wrapper", rows.device, n, int(rows.stride(0)) if n > 1 else 7,
              rows.data_ptr(), T.data_ptr(), int(bool(invert)))
    return T


def accumulate(transforms, lengths=None):
    """transforms (n,4,4) fp64 -> absolute poses abs[f] = T[first] ... T[f]; ``lengths``: frames per sequence (the
    frames of all sequences are stored back to back); one sequence if None."""
    _gpu64(transforms, "transforms")
    n = transforms.shape[0]
    lengths = [n] if lengths is None else list(lengths)
    assert sum(lengths) == n
    st, _ = _starts(lengths, transforms.device)
    out = torch.empty_like(transforms)
    _lib.call("odom_accumulate_kernel_wrapper", transforms.device, len(lengths), st.data_ptr(), transforms.data_ptr(),
              out.data_ptr())
    return out


def trajectory_distances(poses, lengths=None):
    """R/evaluation.py:198-215: dist[first] = 0, dist[f] = dist[f-1] + |p[f] - p[f-1]|, per sequence."""
    _gpu64(poses, "poses")
    n = poses.shape[0]
    lengths = [n] if lengths is None else list(lengths)
    st, _ = _starts(lengths, poses.device)
    dist = torch.empty((n,), dtype=torch.float64, device=poses.device)
    _lib.call("odom_cumulative_distance_kernel_wrapper", poses.device, len(lengths), st.data_ptr(), poses.data_ptr(),
              dist.data_ptr())
    return dist


def sequence_errors(poses_gt, poses_result, dist, lengths=None, segments=DEFAULT_SEGMENTS, step_size=10):
    """R/evaluation.py:236-271 for every sequence at once.  Returns (err (slots,5) fp64 = [first_frame, r_err/len,
    t_err/len, len, speed], valid (slots,) bool, slot_start list): slot = (sequence, first frame, segment)."""
    _gpu64(poses_gt, "poses_gt"); _gpu64(poses_result, "poses_result"); _gpu64(dist, "dist")
    n = poses_gt.shape[0]
    lengths = [n] if lengths is None else list(lengths)
    dev = poses_gt.device
    st, _ = _starts(lengths, dev)
    nlen = len(segments)
    slots = [((m + step_size - 1) // step_size) * nlen for m in lengths]
    sl, sl_host = _starts(slots, dev)
    total = sl_host[-1]
    err = torch.zeros((total, 5), dtype=torch.float64, device=dev)
    valid = torch.zeros((total,), dtype=torch.int32, device=dev)
    seg = torch.tensor([float(s) for s in segments], dtype=torch.float64, device=dev)
    _lib.call("odom_sequence_errors_kernel_wrapper", dev, len(lengths), total, st.data_ptr(), sl.data_ptr(),
              poses_gt.data_ptr(), poses_result.data_ptr(), dist.data_ptr(), int(step_size), nlen, seg.data_ptr(),
              err.data_ptr(), valid.data_ptr())
    return err, valid.bool(), sl_host


# ---- R/slam/eval/eval_odometry.py under its own names (device tensors in, device tensors out) ----------------------

def shift_poses(poses):
    eye = torch.eye(4, dtype=poses.dtype, device=poses.device).unsqueeze(0)
    return torch.cat((eye, poses[:-1, :4, :4]), dim=0)


def compute_relative_poses(poses):
    """eval_odometry.py:247-253: inv(abs[t-1]) @ abs[t], abs[-1] = I."""
    return torch.linalg.inv(shift_poses(poses)) @ poses


def compute_absolute_poses(relative_poses):
    """eval_odometry.py:256-266: abs[0] = rel[0], abs[i+1] = abs[i] @ rel[i+1] -- the scan kernel."""
    return accumulate(relative_poses.contiguous())


def compute_cumulative_trajectory_length(trajectory):
    """eval_odometry.py:268-276 (its first entry is |p[0] - 0|: the shift inserts the identity)."""
    traj = trajectory.contiguous()
    return trajectory_distances(traj) + torch.linalg.norm(traj[0, :3, 3])


def calcSequenceErrors(trajectory, ground_truth, all_segments=DEFAULT_SEGMENTS, step_size=10):
    """eval_odometry.py:316-361 -> list of dicts with the reference's keys (values are Python floats / ints)."""
    gt, tr = ground_truth.contiguous(), trajectory.contiguous()
    dist = compute_cumulative_trajectory_length(gt)
    err, valid, _ = sequence_errors(gt, tr, dist, None, all_segments, step_size)
    rows = err[valid].cpu().numpy()
    out = []
    for first, r, t, seg, speed in rows:
        frames = seg / (0.1 * speed)                                    # = last - first + 1
        out.append({"tr_err": t, "r_err": r, "segment": seg, "speed": speed, "first_frame": int(first),
                    "last_frame": int(first) + int(round(frames)) - 1})
    return out


def compute_kitti_metrics(trajectory, ground_truth, segments_sizes=DEFAULT_SEGMENTS):
    """eval_odometry.py:364-382 -> (avg_tr_err, avg_rot_err, errors) or (None, None)."""
    errors = calcSequenceErrors(trajectory, ground_truth, segments_sizes)
    if len(errors) > 0:
        return (sum(e["tr_err"] for e in errors) / len(errors), sum(e["r_err"] for e in errors) / len(errors), errors)
    return None, None


def compute_ate(relative_predicted, relative_ground_truth):
    """eval_odometry.py:385-393."""
    tr_err = torch.linalg.norm(relative_predicted[:, :3, 3] - relative_ground_truth[:, :3, 3], dim=1)
    ate = tr_err.mean()
    return float(ate), float(torch.sqrt(torch.pow(tr_err - ate, 2).mean()))


def compute_are(relative_trajectory, relative_ground_truth):
    """eval_odometry.py:396-405."""
    eye = torch.eye(3, dtype=relative_trajectory.dtype, device=relative_trajectory.device)
    diff = torch.linalg.inv(relative_ground_truth[:, :3, :3]) @ relative_trajectory[:, :3, :3] - eye
    r_err = torch.linalg.norm(diff, dim=(1, 2))
    are = r_err.mean()
    return float(are), float(torch.sqrt(torch.pow(r_err - are, 2).mean()))


# ---- test_model's accumulation + kittiOdomEval's numbers -------------------------------------------------------------

def _text_round_trip(poses):
    """What R/train.py:945-960 + R/evaluation.py:161-196 do to every pose: written with '%.08f', read back."""
    rows = poses[:, :3, :].reshape(-1, 12).cpu().numpy()
    back = np.char.mod("%.08f", rows).astype(np.float64)
    out = torch.zeros_like(poses)
    out[:, :3, :] = torch.from_numpy(back).reshape(-1, 3, 4).to(poses.device)
    out[:, 3, 3] = 1.0
    return out


class OdometryEvaluator:
    """Collects the network's pose rows batch by batch (on the device) and evaluates every sequence.

        ev = OdometryEvaluator(device)
        for batch ...:
            pose_params, _ = net(...)                       # (B,4,7) on the device
            ev.add_batch(seq_ids, frame_ids, pose_params, gt_q, gt_t)
        results = ev.evaluate()                             # {seq: {"ave_t_err", "ave_r_err", "seq_err", ...}}

    ``add_batch`` is R/train.py:866-893 (kitti_odometry branch) without the per-sample D2H copies: row 0 (level 1)
    of ``pose_params`` and the ground-truth ``[t, q]`` rows are appended to device buffers.  ``evaluate`` is
    R/train.py:918-990 + kittiOdomEval without files and plots."""

    def __init__(self, device, segments=DEFAULT_SEGMENTS, step_size=10, scalar_last=False):
        self.device = torch.device(device)
        self.segments, self.step_size, self.scalar_last = tuple(segments), int(step_size), bool(scalar_last)
        self._pred, self._gt, self._seq, self._frame = [], [], [], []

    def add_batch(self, seq_ids, frame_ids, pose_params, gt_q, gt_t):
        if not pose_params.is_cuda:
            raise RuntimeError("CPU not supported")
        B = pose_params.shape[0]
        assert pose_params.shape[1:] == (4, 7) and len(seq_ids) == B and len(frame_ids) == B, \
            "Sizes from prediction and batch are not matching"
        gt_q = gt_q.reshape(B, 4).to(self.device, torch.float32)
        if self.scalar_last:                                # R/train.py:884-888
            gt_q = torch.cat((gt_q[:, 3:], gt_q[:, :3]), dim=1)
        self._pred.append(pose_params[:, 0, :].detach().to(torch.float32).clone())
        self._gt.append(torch.cat((gt_t.reshape(B, 3).to(self.device, torch.float32), gt_q), dim=1))
        self._seq.extend(int(s) for s in seq_ids)
        self._frame.extend(int(f) for f in frame_ids)

    def _ordered(self):
        """Rows grouped by sequence, frames ascending (the reference's dicts are consumed in sorted key order)."""
        order = sorted(range(len(self._seq)), key=lambda i: (self._seq[i], self._frame[i]))
        idx = torch.tensor(order, dtype=torch.long, device=self.device)
        pred = torch.cat(self._pred).index_select(0, idx).contiguous()
        gt = torch.cat(self._gt).index_select(0, idx).contiguous()
        seqs, lengths = [], []
        for i in order:
            if not seqs or seqs[-1] != self._seq[i]:
                seqs.append(self._seq[i]); lengths.append(0)
            lengths[-1] += 1
        return pred, gt, seqs, lengths

    def trajectories(self):
        """{seq: (abs_pred (n,4,4), abs_gt (n,4,4))} fp64 on the device: what R/train.py:930-937 saves as
        ``XX_pred.txt`` / ``XX_gt.txt``."""
        pred, gt, seqs, lengths = self._ordered()
        ap = accumulate(rows_to_transforms(pred), lengths)
        ag = accumulate(rows_to_transforms(gt), lengths)
        out, lo = {}, 0
        for s, n in zip(seqs, lengths):
            out[s] = (ap[lo:lo + n], ag[lo:lo + n])
            lo += n
        return out

    def evaluate(self, through_text=True):
        """Per sequence: ``ave_t_err`` (fraction; x100 = KITTI t_rel %), ``ave_r_err`` (rad/m; /pi*180*100 = deg/100 m),
        ``seq_err`` (rows of R/evaluation.py:270), ``segment`` / ``speed`` averages (R/evaluation.py:578-630).
        ``through_text=True`` reproduces the reference's digits (its poses pass through '%.08f' text files);
        False skips that host round trip (results agree to ~1e-8)."""
        pred, gt, seqs, lengths = self._ordered()
        ap = accumulate(rows_to_transforms(pred), lengths)
        ag = accumulate(rows_to_transforms(gt), lengths)
        if through_text:
            ap, ag = _text_round_trip(ap), _text_round_trip(ag)
        # loadPoses(relative=True) of the prediction: P[k] = inv(P[0]) @ P[k] for k > 0, frame 0 kept as read
        res = ap.clone()
        lo = 0
        for n in lengths:
            if n > 1:
                res[lo + 1:lo + n] = torch.linalg.inv(ap[lo]) @ ap[lo + 1:lo + n]
            lo += n
        dist = trajectory_distances(ag, lengths)
        err, valid, slot_start = sequence_errors(ag, res, dist, lengths, self.segments, self.step_size)
        out = {}
        for k, s in enumerate(seqs):
            e = err[slot_start[k]:slot_start[k + 1]]
            e = e[valid[slot_start[k]:slot_start[k + 1]]]
            r = {"frames": lengths[k], "seq_err": e, "ave_t_err": None, "ave_r_err": None, "segment": {}, "speed": {}}
            if e.shape[0] > 0:
                r["ave_t_err"], r["ave_r_err"] = float(e[:, 2].mean()), float(e[:, 1].mean())
                for L in self.segments:
                    m = e[:, 3] == float(L)
                    r["segment"][L] = [float(e[m, 2].mean()), float(e[m, 1].mean())] if bool(m.any()) else []
                for key in range(2, 25, 2):
                    m = (e[:, 4] - key).abs() < 2.0
                    r["speed"][key] = [float(e[m, 2].mean()), float(e[m, 1].mean())] if bool(m.any()) else []
                r["t_rel_percent"] = 100.0 * r["ave_t_err"]
                r["r_rel_deg_per_100m"] = r["ave_r_err"] / np.pi * 180.0 * 100.0
            out[s] = r
        return out
