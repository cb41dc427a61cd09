"""CPU tests of the oracle (``-m "not gpu"``): hand-derived known answers for the nine extension
kernels' restatement, and the golden fixtures generated from the imported reference.

The reference has no tests or vectors for this path, so:
  * the C restatement of the CUDA kernels is pinned by the known-answer cases below, each derived
    by hand from the cited .cu lines (tie rule, origin skip, padding semantics, ...);
  * ``oracle.model`` and ``oracle.ops.knn_point`` are pinned by tests/golden/*.npz, which hold
    outputs of the reference's own Python layers (oracle/gen_golden.py).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import model as M
from oracle import ops as O
from oracle import params
from pwclonet_pylidarslam_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------- FPS (sampling_gpu.cu:69-173)
def test_opt_n_threads_table():
    # cuda_utils.h:15-19: 2^floor(log2 n) clamped to [1,512]
    assert [O.opt_n_threads(n) for n in (1, 2, 3, 7, 8, 255, 256, 511, 512, 8192, 120000)] == \
           [1, 2, 2, 4, 8, 128, 256, 256, 512, 512, 512]


def test_fps_basic_line():
    # points on a line: 0, 1, 2, 10 -> start at index 0, then the farthest (10), then 2 (min dist 2
    # to {0,10} is 2 for point 2 and 1 for point 1)
    x = torch.tensor([[[0.1, 0, 0], [1.1, 0, 0], [2.1, 0, 0], [10.1, 0, 0]]])
    assert O.furthest_point_sampling(x, 4).tolist() == [[0, 3, 2, 1]]


def test_fps_tie_rule_bit_reversed_thread():
    """8 points -> bs = 8 threads, one point each.  Points 3 and 5 are exactly equidistant from
    point 0 and farther than the rest.  The tree (sampling_gpu.cu:115-168) compares slot t with
    t+4, then t+2, then t+1 and keeps the LOWER slot on ties: after step 1 slot 3 holds thread 3
    (3 vs 7) and slot 1 holds thread 5 (1 vs 5); step 2 compares slot 1 (thread 5) with slot 3
    (thread 3): tie -> slot 1 -> thread 5 wins.  (bit-reversed ids: 3 -> 110b = 6, 5 -> 101b = 5.)"""
    x = torch.zeros(1, 8, 3)
    x[0, :, 0] = torch.tensor([1.0, 1.1, 1.2, 5.0, 1.3, -3.0, 1.4, 1.5])   # |5-1| == |-3-1| == 4
    out = O.furthest_point_sampling(x, 2)
    assert out.tolist() == [[0, 5]]
    # the same two candidates at indices 2 and 6: bitrev(2)=010b=2, bitrev(6)=011b=3 -> 2 wins
    x[0, :, 0] = torch.tensor([1.0, 1.1, 5.0, 1.2, 1.3, 1.4, -3.0, 1.5])
    assert O.furthest_point_sampling(x, 2).tolist() == [[0, 2]]


def test_fps_same_thread_tie_keeps_first():
    """n = 16 -> bs = 16; n = 24 -> bs = 16, thread t owns points t and t+16: a strict > keeps the
    earlier one (sampling_gpu.cu:108-109)."""
    x = torch.zeros(1, 24, 3)
    x[0, :, 0] = 1.0 + torch.arange(24) * 1e-3
    x[0, 2, 0] = 9.0
    x[0, 18, 0] = 9.0           # same thread (18 = 2 + 16), same distance
    assert O.furthest_point_sampling(x, 2).tolist() == [[0, 2]]


def test_fps_origin_skip_and_exhaustion():
    """|p|^2 <= 1e-3 points are never candidates (:100-101); index 0 is emitted first regardless;
    with no valid point every later index is 0 (best = -1, besti = 0 in every thread)."""
    x = torch.zeros(1, 6, 3)
    x[0, 3] = torch.tensor([1.0, 0, 0])
    x[0, 4] = torch.tensor([0.0, 2.0, 0])
    x[0, 5] = torch.tensor([0.01, 0.01, 0.01])       # squared norm 3e-4: skipped
    out = O.furthest_point_sampling(x, 5).tolist()[0]
    assert out[0] == 0 and set(out[1:3]) == {3, 4} and 5 not in out and 1 not in out and 2 not in out
    assert O.furthest_point_sampling(torch.zeros(1, 6, 3), 4).tolist() == [[0, 0, 0, 0]]


# ---------------------------------------------------------------- gather / group
def test_gather_group_known_answers():
    p = torch.arange(2 * 3 * 5, dtype=torch.float32).reshape(2, 3, 5)
    idx = torch.tensor([[4, 0], [1, 1]], dtype=torch.int32)
    g = O.gather_points(p, idx)
    assert g[0, 1].tolist() == [9.0, 5.0] and g[1, 2].tolist() == [26.0, 26.0]
    gi = torch.tensor([[[0, 4], [2, 2]], [[1, 3], [3, 1]]], dtype=torch.int32)
    gg = O.group_points(p, gi)
    assert gg.shape == (2, 3, 2, 2)
    assert gg[1, 0].tolist() == [[16.0, 18.0], [18.0, 16.0]]
    # grads are the transposes: scatter-add
    go = torch.ones(2, 3, 2)
    gr = O.gather_points_grad(go, idx, 5)
    assert gr[0, 0].tolist() == [1, 0, 0, 0, 1] and gr[1, 0].tolist() == [0, 2, 0, 0, 0]
    gr2 = O.group_points_grad(torch.ones(2, 3, 2, 2), gi, 5)
    assert gr2[0, 1].tolist() == [1, 0, 2, 0, 1]


# ---------------------------------------------------------------- ball query (ball_query_gpu.cu:9-44)
def test_ball_query_semantics():
    xyz = torch.tensor([[[0.0, 0, 0], [0.5, 0, 0], [0.9, 0, 0], [3.0, 0, 0], [0.2, 0, 0]]])
    new_xyz = torch.tensor([[[0.0, 0, 0], [3.0, 0, 0], [10.0, 0, 0]]])
    out = O.ball_query(new_xyz, xyz, 1.0, 3)
    # centre 0: hits 0,1,2,4 in scan order -> first three; centre 1: single hit, padded with it;
    # centre 2: no hit -> the zero-initialised row stays
    assert out.tolist() == [[[0, 1, 2], [3, 3, 3], [0, 0, 0]]]
    # strict <: a point exactly at the radius is not a hit
    out = O.ball_query(torch.tensor([[[0.0, 0, 0]]]), torch.tensor([[[1.0, 0, 0], [0.5, 0, 0]]]), 1.0, 2)
    assert out.tolist() == [[[1, 1]]]
    # fewer hits than nsample: the rest repeats the FIRST hit
    out = O.ball_query(new_xyz[:, :1], xyz, 0.6, 4)
    assert out.tolist() == [[[0, 1, 4, 0]]]


# ---------------------------------------------------------------- three_nn / interpolate
def test_three_nn_and_interpolate():
    known = torch.tensor([[[0.0, 0, 0], [1.0, 0, 0], [1.0, 0, 0], [5.0, 0, 0]]])   # 1 and 2 duplicate
    unknown = torch.tensor([[[0.9, 0, 0]]])
    d2, idx = O.three_nn(unknown, known)
    assert idx.tolist() == [[[1, 2, 0]]]                     # tie -> earlier index first (strict <)
    np.testing.assert_allclose(d2.numpy(), [[[0.01, 0.01, 0.81]]], rtol=1e-6)
    d2, idx = O.three_nn(unknown, known[:, :2])               # m < 3: unfilled = (float)1e40 = inf, idx 0
    assert idx.tolist() == [[[1, 0, 0]]] and torch.isinf(d2[0, 0, 2])
    pts = torch.tensor([[[10.0, 20.0, 30.0, 40.0]]])
    w = torch.tensor([[[0.5, 0.25, 0.25]]])
    out = O.three_interpolate(pts, torch.tensor([[[1, 2, 0]]], dtype=torch.int32), w)
    assert out.tolist() == [[[20 * 0.5 + 30 * 0.25 + 10 * 0.25]]]
    g = O.three_interpolate_grad(torch.tensor([[[2.0]]]), torch.tensor([[[1, 2, 0]]], dtype=torch.int32), w, 4)
    assert g.tolist() == [[[0.5, 1.0, 0.5, 0.0]]]


# ---------------------------------------------------------------- knn_point (pytorch_utils.py:12-49)
def test_knn_formula_and_tie_rule():
    xyz = torch.tensor([[[0.0, 0, 0], [3.0, 4.0, 0], [3.0, 4.0, 0], [1.0, 0, 0]]])
    q = torch.tensor([[[0.0, 0, 0]]])
    d, idx = O.knn_point_with_dist(4, xyz, q)
    assert idx.tolist() == [[[0, 3, 1, 2]]]                    # duplicates: lower index first
    exp = np.sqrt(np.array([0, 1, 25, 25], dtype=np.float32) + np.float32(1e-8))
    assert d.numpy().tolist() == [[exp.tolist()]]
    a, b = O.knn_point(2, xyz, q)                              # reference quirk: returns idx twice
    assert torch.equal(a, b)


def test_knn_golden_from_reference():
    """Index lists produced by the reference's own knn_point (dense distance + torch.topk on this
    container's CPU).  Bitwise equal on these duplicate-free clouds."""
    z = np.load(os.path.join(GOLDEN, "knn_cases.npz"))
    ci = 0
    while f"case{ci}_shape" in z:
        k, n, s, seed = (int(v) for v in z[f"case{ci}_shape"])
        g = torch.Generator().manual_seed(seed)
        xyz = torch.rand(2, n, 3, generator=g) * 40 - 20
        new_xyz = torch.rand(2, s, 3, generator=g) * 40 - 20
        _, idx = O.knn_point_with_dist(k, xyz, new_xyz)
        assert torch.equal(idx, torch.from_numpy(z[f"case{ci}_idx"])), (k, n, s)
        ci += 1
    assert ci == 9


# ---------------------------------------------------------------- whole network vs reference golden
def _inputs(meta):
    if meta["generator"] == "uniform":
        pc1, pc2 = synthetic.uniform_pair(meta["seed"], meta["npoints"], meta["batch"])
    else:
        pc1, pc2, _, _ = synthetic.kitti_like_pair(meta["seed"], meta["npoints"], meta["batch"])
    to = lambda p: torch.from_numpy(p[:, :, :3]).permute(0, 2, 1).contiguous()
    return to(pc1), to(pc2)


@pytest.mark.parametrize("case", ["n1024_b2", "n8192_b1"])
def test_oracle_model_matches_reference_golden(case):
    """BASELINE.json configs[0] (2x1024 plumbing case) and configs[1] (one 2x8192 pair) on the CPU
    oracle.  `pose_params_oracle_knn` is the reference run with its knn_point swapped for the
    oracle's (isolates torch.topk's unspecified tie order and MKL's non-IEEE sqrt): bit-identical.
    `pose_params` is the reference as shipped: equal within the end-to-end fp32 bound."""
    z = np.load(os.path.join(GOLDEN, "pwclonet_%s.npz" % case))
    meta = json.loads(str(z["meta"]))
    with open(os.path.join(GOLDEN, "state_shapes.json")) as f:
        sd = params.make_state_dict(json.load(f))
    x1, x2 = _inputs(meta)
    taps = {}
    pose = M.pwclonet_forward(sd, x1, x2, taps)
    assert torch.equal(pose, torch.from_numpy(z["pose_params_oracle_knn"]))
    torch.testing.assert_close(pose, torch.from_numpy(z["pose_params"]), rtol=0, atol=1e-5)
    assert torch.equal(taps["f1.psa_1.fps_idx"], torch.from_numpy(z["f1.psa_1.fps_idx"]))
    assert torch.equal(taps["f2.psa_3.new_xyz"], torch.from_numpy(z["f2.psa_3.new_xyz"]))
    assert pose.shape == (meta["batch"], 4, 7)
    # rows are [t(3), unit quaternion(4)], scalar first
    np.testing.assert_allclose(pose[:, :, 3:].norm(dim=-1).numpy(), 1.0, atol=1e-5)


def test_oracle_train_step_matches_reference_train_golden():
    """``oracle.model.pwclonet_train_step`` (batch-statistic BatchNorm, dropout off, loss + backward) against the
    values recorded from the imported reference in the same mode (oracle/gen_train_golden.py): forward bit for bit,
    gradients to the summation-order noise of torch's CPU convolutions, running statistics after the step."""
    from oracle import gen_golden
    from oracle.gen_grad_golden import ground_truth
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_n1024_b2.npz"))
    meta = json.loads(str(z["meta"]))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_shapes.json")) as f:
        sd = params.make_state_dict(json.load(f))
    x1, x2 = gen_golden.case_inputs(meta["case"])
    pose, loss, grads, gs = M.pwclonet_train_step(sd, x1, x2, ground_truth(x1.shape[0]))
    np.testing.assert_allclose(pose.numpy(), z["pose_params"], rtol=0, atol=2e-6)
    assert abs(loss.item() - float(z["loss"])) <= 1e-6 * abs(float(z["loss"]))
    for k in meta["params"]:
        ref = z["grad." + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 5e-5 * np.abs(ref).max(), k
    np.testing.assert_allclose(gs.numpy(), z["grad_s"], rtol=1e-6)
    for k in meta["bn_layers"]:
        np.testing.assert_allclose(sd[k + ".running_mean"].numpy(), z["buf.%s.running_mean" % k], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(sd[k + ".running_var"].numpy(), z["buf.%s.running_var" % k], rtol=1e-6, atol=1e-7)
        # psa_* run once per frame (PW/pwclo_net.py:140-160): two statistics updates per step there
        assert int(sd[k + ".num_batches_tracked"]) == int(z["buf.%s.num_batches_tracked" % k])
    l2 = np.array([grads[k].double().norm().item() for k in meta["all_names"]])
    np.testing.assert_allclose(l2, z["all_grad_l2"], rtol=1e-4, atol=1e-7 * z["all_grad_l2"].max())
    # the same step in float64 against the reference model run in float64 (the GPU tests' yardstick)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_shapes.json")) as f:
        sd = params.make_state_dict(json.load(f))
    pose, loss, grads, gs = M.pwclonet_train_step(sd, x1, x2, ground_truth(x1.shape[0]), dtype=torch.float64)
    np.testing.assert_allclose(pose.numpy(), z["pose64"], rtol=0, atol=1e-12)
    assert abs(loss.item() - float(z["loss64"])) <= 1e-12 * abs(float(z["loss64"]))
    for k in meta["params"]:
        ref = z["grad64." + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 1e-10 * np.abs(ref).max(), k
    l2 = np.array([grads[k].norm().item() for k in meta["all_names"]])
    np.testing.assert_allclose(l2, z["all_grad64_l2"], rtol=1e-10)


def test_warp_known_answer():
    # rotate (1,0,0) by 90 deg about z (scalar-first quaternion), then translate
    q = torch.tensor([[[np.cos(np.pi / 4)], [0.0], [0.0], [np.sin(np.pi / 4)]]], dtype=torch.float32)
    out = M.warp(torch.tensor([[[1.0], [0.0], [0.0]]]), q, torch.tensor([[[1.0], [2.0], [3.0]]]))
    np.testing.assert_allclose(out.flatten().numpy(), [1.0, 3.0, 3.0], atol=1e-6)


def test_loss_module_matches_reference_golden():
    """SURVEY section 8 f3 (first slice): pwclonet_pylidarslam_amd.loss.PWCLONetLossModule against values
    recorded from the imported reference loss (oracle/gen_loss_golden.py): loss, every log_dict scalar,
    d loss / d pred_params and d loss / d s_param, both weighting modes."""
    import json
    from oracle.gen_loss_golden import inputs
    from pwclonet_pylidarslam_amd.loss import PWCLONetLossModule
    z = np.load(os.path.join(GOLDEN, "loss_cases.npz"))
    meta = json.loads(str(z["meta"]))
    for case in meta["cases"]:
        tag = "s%d" % case["seed"]
        mod = PWCLONetLossModule(dict(with_exp_weights=case["with_exp_weights"], init_weights=case["init_weights"],
                                      loss_weights=case["loss_weights"], loss_option="l2_norm", nb_levels=4,
                                      scalar_last=False))
        pred, gt = inputs(case["seed"], case["batch"])
        pred.requires_grad_(True)
        loss, log = mod(pred, gt)
        loss.backward()
        assert sorted(log.keys()) == case["keys"]
        assert loss.item() == float(z[tag + ".loss"])                      # same ops, same order, same device
        for k in case["keys"]:
            np.testing.assert_allclose(np.asarray(log[k].detach()), z[tag + ".log." + k], rtol=0, atol=0)
        np.testing.assert_allclose(pred.grad.numpy(), z[tag + ".grad_pred"], rtol=1e-6, atol=1e-8)
        if case["with_exp_weights"]:
            np.testing.assert_allclose(mod.exp_weighting.s_param.grad.numpy(), z[tag + ".grad_s"], rtol=1e-6, atol=0)
            assert list(mod.state_dict().keys()) == ["exp_weighting.s_param"]


def test_preprocess_oracle_known_values():
    """oracle.preprocess.transform_filter against hand-computed values."""
    from oracle.preprocess import transform_filter
    tr = [[0, -1, 0, 0.5], [0, 0, -1, -0.25], [1, 0, 0, 2.0]]              # velodyne (x fwd, y left, z up) -> camera
    pts = np.array([[10, 2, 1, 0.3],        # -> (-1.5, -1.25, 12): kept
                    [40, 0, 0, 0.1],        # -> z = 42: too far
                    [5, -31, 0, 0.2],       # -> x = 31.5: too far
                    [5, 0, -2, 0.9],        # -> y = 1.75 > 1.1: ground
                    [28, 29.5, -1.35, 0.0]], dtype=np.float32)  # -> (-29, ~1.1, 30): z == 30 is not < 30
    q, keep = transform_filter(pts, tr)
    np.testing.assert_allclose(q[0], [-1.5, -1.25, 12.0])
    assert keep.tolist() == [True, False, False, False, False]
    assert q.dtype == np.float64


def test_preprocess_oracle_kitti360_known_values():
    """oracle.preprocess.kitti360_filter against hand-computed values (kitti_360_dataset_2.py:113-123)."""
    from oracle.preprocess import kitti360_filter
    pts = np.array([[10, 2, 1, 0.3],          # kept
                    [10, 2, -1.5, 0.3],       # z < -1.43: ground
                    [10, 2, -1.43, 0.3],      # float32(-1.43) is not < float32(-1.43): kept
                    [35, 0, 0, 0.1],          # x == near: not < near
                    [0, -34.9, 0, 0.1],       # kept
                    [0, -35.1, 0, 0.1]], dtype=np.float32)
    xyz, keep = kitti360_filter(pts, 35.0)
    assert keep.tolist() == [True, False, True, False, True, False]
    assert xyz.dtype == np.float32 and xyz.shape == (6, 3)


# ---- odometry evaluation: oracle/eval_oracle.py against values recorded from the imported
# ---- reference (oracle/gen_eval_golden.py -> tests/golden/eval_cases.npz)

def _eval_cases():
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval_cases.npz"))
    return z, json.loads(str(z["meta"]))["cases"]


def test_eval_oracle_matches_reference_values():
    from oracle import eval_oracle as EO
    from oracle.gen_eval_golden import synthetic_rows
    z, cases = _eval_cases()
    for q, R in zip(z["quat.q"], z["quat.R"]):
        assert np.array_equal(EO.quat2mat(q), R)                       # same expressions, bitwise
    assert np.array_equal(EO.quat2mat(np.array([1e-5, 2e-5, 0, 0])), np.eye(3))
    for name, cfg in cases.items():
        gt_rows, pred_rows = synthetic_rows(**cfg)
        r = EO.kitti_odom_eval(pred_rows, gt_rows)
        np.testing.assert_allclose(r["abs_pred"], z[name + ".abs_pred"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["abs_gt"], z[name + ".abs_gt"], rtol=0, atol=1e-12)
        P, G = z[name + ".abs_pred"], z[name + ".abs_gt"]
        np.testing.assert_allclose(EO.compute_relative_poses(P), z[name + ".eo.rel_of_abs"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(EO.compute_absolute_poses(EO.compute_relative_poses(P)), z[name + ".eo.abs_of_rel"],
                                   rtol=0, atol=1e-10)
        np.testing.assert_allclose(EO.compute_cumulative_trajectory_length(G), z[name + ".eo.cumlen"], rtol=1e-14)
        np.testing.assert_allclose(EO.calc_sequence_errors_arrays(P, G), z[name + ".eo.seq_err"], rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(EO.compute_ate(EO.compute_relative_poses(P), EO.compute_relative_poses(G)),
                                   z[name + ".eo.ate"], rtol=1e-12)
        np.testing.assert_allclose(EO.compute_are(EO.compute_relative_poses(P), EO.compute_relative_poses(G)),
                                   z[name + ".eo.are"], rtol=1e-12)
        np.testing.assert_allclose(r["seq_err"], z[name + ".ke.seq_err"], rtol=1e-12, atol=1e-15)
        if r["seq_err"].shape[0]:
            np.testing.assert_allclose([r["ave_t_err"], r["ave_r_err"]], z[name + ".ke.overall"], rtol=1e-12)
            seg = z[name + ".ke.segment"]
            for L, t, rr in seg:
                got = r["segment"][int(L)]
                assert (got == [] and np.isnan(t)) or np.allclose(got, [t, rr], rtol=1e-12)
            for key, t, rr in z[name + ".ke.speed"]:
                got = r["speed"][int(key)]
                assert (got == [] and np.isnan(t)) or np.allclose(got, [t, rr], rtol=1e-12)
        else:
            assert z[name + ".ke.seq_err"].shape[0] == 0 and r["ave_t_err"] is None
