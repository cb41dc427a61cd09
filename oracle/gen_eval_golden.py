#!/usr/bin/env python3
"""Golden vectors for the odometry-evaluation row -- build container only.

Imports the REFERENCE's own evaluation code on CPU and records its outputs on synthetic trajectories:
  * ``slam/eval/eval_odometry.py``: compute_relative_poses, compute_absolute_poses,
    compute_cumulative_trajectory_length, calcSequenceErrors, compute_kitti_metrics, compute_ate, compute_are;
  * ``slam/common/kitti360_utils.py``: KITTI360_TOOLS.quat2mat (the same nibabel routine train.py:762-795 carries as a
    method), KITTI360_TRANSFORMATIONS.convert_to_absolute (train.py:930-931);
  * ``evaluation.py``: kittiOdomEval.loadPoses / calcSequenceErrors / computeOverallErr / computeSegmentErr /
    computeSpeedErr on the very text files train.py:945-960 would have written (``np.savetxt(fmt='%.08f')``).
``train.py`` itself cannot be imported here (it pulls in the datasets, which need open3d; hydra is absent too), so its
per-sample loop :866-893 is restated in oracle/eval_oracle.py and pinned through the functions above.
In-process stubs: ``seaborn`` and ``cv2`` (plotting-only imports of eval_odometry.py, absent from this image);
matplotlib runs on the Agg backend; no plot is produced.  Output: tests/golden/eval_cases.npz (inputs regenerated
from seeds by the tests, expected outputs stored).

    python oracle/gen_eval_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import eval_oracle as EO  # noqa: E402
from oracle import ref_import  # noqa: E402

CASES = {"drive_a": dict(seed=11, n=900, speed=9.0, noise=0.02), "drive_b": dict(seed=12, n=1500, speed=14.0, noise=0.05),
         "short": dict(seed=13, n=60, speed=6.0, noise=0.01)}


def synthetic_rows(seed, n, speed, noise):
    """Ground-truth and 'predicted' pose rows [t(3), q(w,x,y,z)] of n frame pairs in the camera frame (z forward,
    ~0.1 s apart): a smooth drive with yaw changes; the prediction is the ground truth plus noise, non-unit
    quaternions included (the network's rows are normalised, the evaluation must not rely on it)."""
    rng = np.random.default_rng(seed)
    yaw = 0.02 * np.sin(np.arange(n) / 40.0) + rng.normal(0, 0.002, n)
    gt = np.zeros((n, 7), dtype=np.float32)
    gt[:, 2] = 0.1 * speed * (1.0 + 0.2 * np.sin(np.arange(n) / 90.0))
    gt[:, 0] = rng.normal(0, 0.01, n)
    gt[:, 3] = np.cos(yaw / 2)
    gt[:, 5] = np.sin(yaw / 2)                       # rotation about y (camera frame "up")
    pred = gt.copy()
    pred[:, :3] += rng.normal(0, noise, (n, 3)).astype(np.float32)
    pred[:, 3:] += rng.normal(0, noise * 0.05, (n, 4)).astype(np.float32)
    pred[:, 3:] *= (1.0 + rng.normal(0, 0.01, (n, 1))).astype(np.float32)
    return gt, pred


def main():
    ref_import.load()
    import matplotlib
    matplotlib.use("Agg")
    for name in ("seaborn", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    import importlib
    ev = importlib.import_module("slam.eval.eval_odometry")
    ku = importlib.import_module("slam.common.kitti360_utils")
    evaluation = importlib.import_module("evaluation")

    out = {"meta": json.dumps({"cases": CASES, "generator": "oracle/gen_eval_golden.py: synthetic_rows"})}
    # quat2mat: reference routine on a few quaternions (unit, non-unit, near-zero)
    qs = np.array([[1, 0, 0, 0], [0.9, 0.1, -0.2, 0.3], [2.0, 0.4, 0.1, -1.0], [1e-5, 2e-5, 0, 0], [0, 0, 1, 0]],
                  dtype=np.float64)
    out["quat.q"] = qs
    out["quat.R"] = np.stack([ku.KITTI360_TOOLS.quat2mat(q) for q in qs])
    for q in qs:
        assert np.array_equal(EO.quat2mat(q), ku.KITTI360_TOOLS.quat2mat(q))

    for name, cfg in CASES.items():
        gt_rows, pred_rows = synthetic_rows(**cfg)
        rel_p = {i: m for i, m in enumerate(EO.rows_to_relative(pred_rows))}
        rel_g = {i: m for i, m in enumerate(EO.rows_to_relative(gt_rows))}
        # reference: relative -> absolute (train.py:930-931)
        abs_p = ku.KITTI360_TRANSFORMATIONS.convert_to_absolute(rel_p, velo_to_pose=False)
        abs_g = ku.KITTI360_TRANSFORMATIONS.convert_to_absolute(rel_g, velo_to_pose=False)
        frames = sorted(abs_p)
        P = np.stack([abs_p[f] for f in frames])
        G = np.stack([abs_g[f] for f in frames])
        out[name + ".abs_pred"], out[name + ".abs_gt"] = P, G
        # reference: eval_odometry.py array functions
        out[name + ".eo.rel_of_abs"] = ev.compute_relative_poses(P)
        out[name + ".eo.abs_of_rel"] = ev.compute_absolute_poses(ev.compute_relative_poses(P))
        out[name + ".eo.cumlen"] = ev.compute_cumulative_trajectory_length(G)
        errs = ev.calcSequenceErrors(P, G)
        out[name + ".eo.seq_err"] = np.array([[e["first_frame"], float(e["r_err"][0]), float(e["tr_err"][0]), e["segment"],
                                               e["speed"], e["last_frame"]] for e in errs],
                                             dtype=np.float64).reshape(-1, 6)
        km = ev.compute_kitti_metrics(P, G)
        out[name + ".eo.kitti"] = np.array([np.nan, np.nan] if km[0] is None else [float(km[0]), float(km[1])])
        out[name + ".eo.ate"] = np.array(ev.compute_ate(ev.compute_relative_poses(P), ev.compute_relative_poses(G)))
        out[name + ".eo.are"] = np.array(ev.compute_are(ev.compute_relative_poses(P), ev.compute_relative_poses(G)))
        # reference: kittiOdomEval on the text files train.py would write
        with tempfile.TemporaryDirectory() as tmp:
            gt_dir, res_dir = os.path.join(tmp, "gt"), os.path.join(tmp, "pred")
            os.makedirs(gt_dir); os.makedirs(res_dir)
            np.savetxt(os.path.join(res_dir, "00_pred.txt"), P[:, :3, :].reshape(-1, 12), fmt="%.08f")
            np.savetxt(os.path.join(gt_dir, "00.txt"), G[:, :3, :].reshape(-1, 12), fmt="%.08f")
            cfg_obj = ref_import._DictConfig(gt_dir=gt_dir, result_dir=res_dir, eva_seqs="00_pred", toCameraCoord=False)
            ke = evaluation.kittiOdomEval(cfg_obj)
            poses_result = ke.loadPoses(os.path.join(res_dir, "00_pred.txt"), toCameraCoord=False)
            poses_gt = ke.loadPoses(os.path.join(gt_dir, "00.txt"), toCameraCoord=False, relative=False)
            seq_err = ke.calcSequenceErrors(poses_gt, poses_result)
            out[name + ".ke.seq_err"] = np.array(seq_err, dtype=np.float64).reshape(-1, 5)
            if seq_err:
                out[name + ".ke.overall"] = np.array(ke.computeOverallErr(seq_err))
                seg = ke.computeSegmentErr(seq_err)
                spd = ke.computeSpeedErr(seq_err)
                out[name + ".ke.segment"] = np.array([[k] + (v if v else [np.nan, np.nan]) for k, v in seg.items()])
                out[name + ".ke.speed"] = np.array([[k] + (v if v else [np.nan, np.nan]) for k, v in spd.items()])
        # the restatement agrees with all of it (the committed CPU test repeats these checks from the fixture)
        mine = EO.kitti_odom_eval(pred_rows, gt_rows)
        np.testing.assert_allclose(mine["abs_pred"], P, rtol=0, atol=1e-12)
        np.testing.assert_allclose(mine["seq_err"], out[name + ".ke.seq_err"], rtol=1e-12, atol=1e-15)
        print(name, "frames", len(frames), "segments", len(seq_err),
              "t_rel %% %.4f r_rel deg/100m %.4f" % ((out[name + ".ke.overall"][0] * 100, out[name + ".ke.overall"][1] / np.pi * 180 * 100)
                                                      if seq_err else (float("nan"), float("nan"))))
    path = os.path.join(ROOT, "tests", "golden", "eval_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
