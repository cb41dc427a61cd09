"""Pure-PyTorch CPU restatement of the PWCLO-Net forward -- TEST INFRASTRUCTURE.

Functional style: every layer is a function of ``(state_dict, key_prefix, tensors)`` so the
oracle shares nothing with the product's ``nn.Module`` mirror except the reference's
``state_dict`` key names (SURVEY.md section 5 "checkpoint / resume").  Eval-mode semantics
(BatchNorm running statistics, dropout off) for ``pwclonet_forward``: the mode of the forward parity
fixtures and the headline benchmark (SURVEY.md section 7 "Hard parts").  ``pwclonet_train_step`` is the
TRAINING forward + loss + backward (batch-statistic BatchNorm with its running-statistics update, dropout
off, autograd through the C ops' ``*_grad`` functions), pinned by tests/golden/train_n1024_b2.npz
(oracle/gen_train_golden.py: values recorded from the imported reference in the same mode).

Extension ops and ``knn_point`` come from ``oracle.ops`` (C restatement).  Citations:
PW = /root/reference/slam/models/PWCLONet, P2 = .../pointnet2_ops_lib/pointnet2_ops.

Validated against the imported reference by ``oracle/gen_golden.py`` (bit-identical
``pose_params`` on this container's CPU when the reference's ``knn_point`` is replaced by
the oracle's; see tests/golden/README.md for what is compared and how).
"""
import torch
import torch.nn.functional as F

from oracle import ops

BN_EPS = 1e-5  # nn.BatchNorm2d default, P2/pytorch_utils.py:95-103
BN_MOMENTUM = 0.1  # nn.BatchNorm2d default (the reference passes no momentum, P2/pytorch_utils.py:95-103)
_BN_TRAIN = False  # set by pwclonet_train_step for the duration of one call


class _Group(torch.autograd.Function):
    """P2/pointnet2_utils.py:194-240 (GroupingOperation): backward = ``group_points_grad``."""

    @staticmethod
    def forward(ctx, points, idx):
        ctx.save_for_backward(idx)
        ctx.n = points.shape[2]
        return ops.group_points(points.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ops.group_points_grad(g.contiguous(), idx, ctx.n), None


class _Gather(torch.autograd.Function):
    """P2/pointnet2_utils.py:68-101 (GatherOperation): backward = ``gather_points_grad``."""

    @staticmethod
    def forward(ctx, points, idx):
        ctx.save_for_backward(idx)
        ctx.n = points.shape[2]
        return ops.gather_points(points.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ops.gather_points_grad(g.contiguous(), idx, ctx.n), None


def _group(points, idx):
    if points.dtype == torch.float64:       # float64 evaluation (pwclonet_train_step(dtype=float64)): pure index ops
        B, C, N = points.shape
        _, S, K = idx.shape
        return torch.gather(points.unsqueeze(2).expand(B, C, S, N), 3, idx.long().unsqueeze(1).expand(B, C, S, K))
    return _Group.apply(points, idx) if torch.is_grad_enabled() else ops.group_points(points.contiguous(), idx)


def _gather(points, idx):
    if points.dtype == torch.float64:
        return torch.gather(points, 2, idx.long().unsqueeze(1).expand(points.shape[0], points.shape[1], idx.shape[1]))
    return _Gather.apply(points, idx) if torch.is_grad_enabled() else ops.gather_points(points.contiguous(), idx)


def shared_mlp(sd, prefix, x):
    """P2/pytorch_utils.py:52-83,114-167: stack of Conv2d(1x1, no bias) -> BN -> ReLU over
    (B, C, S, K)."""
    i = 0
    while f"{prefix}.layer{i}.conv.weight" in sd:
        p = f"{prefix}.layer{i}"
        x = F.conv2d(x, sd[p + ".conv.weight"])
        if _BN_TRAIN:   # batch statistics; running_* updated in place (momentum 0.1, unbiased variance)
            x = F.batch_norm(x, sd[p + ".bn.bn.running_mean"], sd[p + ".bn.bn.running_var"],
                             sd[p + ".bn.bn.weight"], sd[p + ".bn.bn.bias"], True, BN_MOMENTUM, BN_EPS)
            sd[p + ".bn.bn.num_batches_tracked"] += 1
        else:
            x = F.batch_norm(x, sd[p + ".bn.bn.running_mean"], sd[p + ".bn.bn.running_var"],
                             sd[p + ".bn.bn.weight"], sd[p + ".bn.bn.bias"], False, 0.0, BN_EPS)
        x = F.relu(x)
        i += 1
    assert i > 0, f"no layers under {prefix}"
    return x


def knn_idx(nsample, xyz, new_xyz):
    # always from float32 coordinates (the float64 evaluation keeps the float32 path's neighbour lists)
    return ops.knn_point_with_dist(nsample, xyz.detach().float().contiguous(), new_xyz.detach().float().contiguous())[1]


def set_abstraction(sd, prefix, npoint, nsample, xyz, features, taps=None, tap=None):
    """P2/pointnet2_modules.py:179-245 (PointnetSAModulePWCLONet.forward).
    xyz (B,N,3), features (B,C,N) or None -> new_xyz (B,npoint,3), new_features (B,C',npoint)."""
    xyz_flipped = xyz.transpose(1, 2).contiguous()
    fps = ops.furthest_point_sampling(xyz.detach().float().contiguous(), npoint)
    new_xyz = _gather(xyz_flipped, fps).transpose(1, 2).contiguous()
    idx = knn_idx(nsample, xyz, new_xyz)
    grouped_xyz = _group(xyz_flipped, idx)
    xyz_diff = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
    if features is not None:
        x = torch.cat((xyz_diff, _group(features.contiguous(), idx)), dim=1)
    else:
        x = torch.cat((xyz_diff, grouped_xyz), dim=1)
    x = shared_mlp(sd, prefix + ".mlp_module", x)
    new_features = x.max(dim=3)[0]
    if taps is not None and tap:
        taps[tap + ".fps_idx"] = fps
        taps[tap + ".knn_idx"] = idx
        taps[tap + ".new_xyz"] = new_xyz
        taps[tap + ".new_features"] = new_features
    return new_xyz, new_features


def set_upconv(sd, prefix, nsample, xyz2, xyz1, features2, features1, taps=None, tap=None):
    """P2/pointnet2_modules.py:479-515 (PointnetFPModulePWCLONet.forward, knn=True branch).
    Propagates features1 (B,C1,N1) at xyz1 (B,N1,3) onto xyz2 (B,N2,3)."""
    idx = knn_idx(nsample, xyz1, xyz2)
    if taps is not None and tap:
        taps[tap + ".idx"] = idx
    x = _group(features1.contiguous(), idx)
    grouped_xyz = _group(xyz1.transpose(1, 2).contiguous(), idx)
    xyz_diff = grouped_xyz - xyz2.transpose(1, 2).unsqueeze(-1)
    x = torch.cat((x, xyz_diff), dim=1)
    x = shared_mlp(sd, prefix + ".mlp", x)
    x = x.max(dim=3)[0]
    if features2 is not None:
        x = torch.cat((x, features2), dim=1)
    x = shared_mlp(sd, prefix + ".post_mlp", x.unsqueeze(-1))
    return x.squeeze(-1)


def _geometry10(centre_xyz, grouped_xyz, k):
    """PW/costvolume.py:92-105: [p, q, q-p, ||q-p||] with sqrt(sum(sq)+1e-20)."""
    p = centre_xyz.unsqueeze(3).repeat(1, 1, 1, k)
    diff = grouped_xyz - p
    euc = torch.sqrt(torch.sum(torch.square(diff), dim=1, keepdim=True) + 1e-20)
    return torch.cat((p, grouped_xyz, diff, euc), dim=1)


def cost_volume(sd, prefix, nsample, nsample_q, warped_xyz, warped_points, f2_xyz, f2_points,
                taps=None, tap=None):
    """PW/costvolume.py:63-190.  warped_xyz (B,3,S), warped_points (B,C1,S), f2_xyz (B,3,N),
    f2_points (B,C2,N) -> (B,64,S)."""
    warped_xyz_t = warped_xyz.permute(0, 2, 1).contiguous()
    f2_xyz_t = f2_xyz.permute(0, 2, 1).contiguous()

    # first aggregate: neighbours of each (warped) frame-1 point among frame 2
    idx_q = knn_idx(nsample_q, f2_xyz_t, warped_xyz_t)
    q_xyz = _group(f2_xyz.contiguous(), idx_q)
    q_pts = _group(f2_points.contiguous(), idx_q)
    geo = _geometry10(warped_xyz, q_xyz, nsample_q)
    p_pts = warped_points.unsqueeze(3).repeat(1, 1, 1, nsample_q)
    feat = shared_mlp(sd, prefix + ".mlp_convs", torch.cat((geo, p_pts, q_pts), dim=1))
    enc = shared_mlp(sd, prefix + ".mlp_conv_xyz_1", geo)
    w = shared_mlp(sd, prefix + ".mlp2_convs", torch.cat((enc, feat), dim=1))
    w = F.softmax(w, dim=3)
    first = torch.sum(w * feat, dim=3)

    # second aggregate: neighbours of each frame-1 point among frame 1
    idx = knn_idx(nsample, warped_xyz_t, warped_xyz_t)
    c_xyz = _group(warped_xyz.contiguous(), idx)
    c_pts = _group(first.contiguous(), idx)
    geo2 = _geometry10(warped_xyz, c_xyz, nsample)
    enc2 = shared_mlp(sd, prefix + ".mlp_conv_xyz_2", geo2)
    p_pts2 = warped_points.unsqueeze(3).repeat(1, 1, 1, nsample)
    w2 = shared_mlp(sd, prefix + ".mlp3_convs", torch.cat((enc2, p_pts2, c_pts), dim=1))
    w2 = F.softmax(w2, dim=3)
    out = torch.sum(w2 * c_pts, dim=3)
    if taps is not None and tap:
        taps[tap + ".idx_q"] = idx_q
        taps[tap + ".idx"] = idx
        taps[tap + ".first"] = first
        taps[tap + ".out"] = out
    return out


def flow_predictor(sd, prefix, points_f1, cost_vol, upsampled=None):
    """PW/flowpredictor.py:53-83."""
    parts = [points_f1, cost_vol] if upsampled is None else [points_f1, cost_vol, upsampled]
    x = torch.cat(parts, dim=1).unsqueeze(3)
    return shared_mlp(sd, prefix + ".mlp_convs", x).squeeze(3)


def pose_calculator(sd, prefix, emb, mask):
    """PW/pose_calculator.py:47-86 in eval mode (dropout is the identity).
    Returns q (B,4,1) normalised, t (B,3,1)."""
    s = torch.sum(emb * mask, dim=2, keepdim=True)
    big = F.conv1d(s, sd[prefix + ".conv1d_q_t.conv.weight"], sd[prefix + ".conv1d_q_t.conv.bias"])
    q = F.conv1d(big, sd[prefix + ".conv1d_q.conv.weight"], sd[prefix + ".conv1d_q.conv.bias"])
    q = q / (torch.sqrt(torch.sum(q * q, dim=1, keepdim=True) + 1e-10) + 1e-10)
    t = F.conv1d(big, sd[prefix + ".conv1d_t.conv.weight"], sd[prefix + ".conv1d_t.conv.bias"])
    return q, t


# ---- quaternion helpers, PW/PWCLO_utils.py:31-132 (scalar-first, `scalar_last` is ignored) ----

def _hamilton(a, b):
    """(B,4,N) x (B,4,N or 1) Hamilton product with the reference's term order
    (PWCLO_utils.py:83-95 / 117-129: both functions expand to the same expression with the
    left operand's components first)."""
    a0, a1, a2, a3 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    b0, b1, b2, b3 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    r0 = a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3
    r1 = a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2
    r2 = a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1
    r3 = a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0
    return torch.stack((r0, r1, r2, r3), dim=1)


def inv_q(q):
    """PWCLO_utils.py:31-39: conj(q) / (|q|^2 + 1e-10); q (B,4)."""
    q2 = torch.sum(q * q, dim=-1, keepdim=True) + 1e-10
    sign = torch.tensor([1, -1, -1, -1])  # int64 in the reference (:36); promotes to fp32
    return (q * sign) / q2


def warp(xyz, q, t):
    """PWCLO_utils.py:42-63: q (x) (0,p) (x) q^-1 + t.  xyz (B,3,N), q (B,4,1), t (B,3,1)."""
    B, _, N = xyz.shape
    qi = inv_q(q.squeeze(2)).reshape(B, 4, 1)
    p = torch.cat((torch.zeros(B, 1, N, dtype=xyz.dtype), xyz), dim=1)
    r = _hamilton(_hamilton(q.reshape(B, 4, 1), p), qi)
    return r[:, 1:, :] + t


def pose_warp_refinement(sd, prefix, last, xyz_f1, points_f1, xyz_f2, points_f2, xyz_f1_prev,
                         points_f1_prev, mask_prev, q_prev, t_prev, taps=None, tap=None):
    """PW/pose_warp_refinement.py:82-158."""
    B = xyz_f1.shape[0]
    q_coarse = q_prev.reshape(B, 4, 1)
    t_coarse = t_prev.reshape(B, 3, 1)
    xyz_f1_t = xyz_f1.permute(0, 2, 1).contiguous()
    xyz_prev_t = xyz_f1_prev.permute(0, 2, 1).contiguous()

    up_feat = set_upconv(sd, prefix + ".setupconv_features", 8, xyz_f1_t, xyz_prev_t, points_f1,
                         points_f1_prev, taps, tap + ".up" if tap else None)
    up_mask = set_upconv(sd, prefix + ".setupconv_mask", 8, xyz_f1_t, xyz_prev_t, points_f1,
                         mask_prev)
    warped = warp(xyz_f1, q_coarse, t_coarse)
    resid = cost_volume(sd, prefix + ".cost_volume", 4, 6, warped, points_f1, xyz_f2, points_f2,
                        taps, tap + ".cv" if tap else None)
    emb = flow_predictor(sd, prefix + ".flow_predictor_features", points_f1, resid, up_feat)
    if not last:
        mask = flow_predictor(sd, prefix + ".flow_predictor_mask", up_mask, emb, points_f1)
    else:
        mask = up_mask
    w = F.softmax(mask, dim=2)
    q_det, t_det = pose_calculator(sd, prefix + ".pose_calculator", emb, w)
    q = _hamilton(q_det, q_coarse).squeeze(2)          # mul_point_q(q_det, q_coarse), :139
    t = warp(t_coarse, q_det, t_det).squeeze(2)        # :148
    if taps is not None and tap:
        taps[tap + ".up_feat"] = up_feat
        taps[tap + ".up_mask"] = up_mask
        taps[tap + ".warped"] = warped
        taps[tap + ".emb"] = emb
        taps[tap + ".mask"] = mask
        taps[tap + ".q"] = q
        taps[tap + ".t"] = t
    return q, t, emb, mask


def _normalise_q(q):
    return q / (torch.sqrt(torch.sum(q * q, dim=-1, keepdim=True) + 1e-10) + 1e-10)


@torch.no_grad()
def pwclonet_forward(sd, xyz_f1, xyz_f2, taps=None):
    """Eval-mode forward without autograd; see ``_forward``."""
    return _forward(sd, xyz_f1, xyz_f2, taps)


def _forward(sd, xyz_f1, xyz_f2, taps=None):
    """PW/pwclo_net.py:109-207 with points_f1 = points_f2 = None.
    xyz_f1, xyz_f2: (B,3,N) float32 CPU.  Returns pose_params (B,4,7), rows = levels 1..4,
    each [tx,ty,tz,qw,qx,qy,qz]."""
    sa_cfg = (("psa_1", 2048, 32), ("psa_2", 1024, 32), ("psa_3", 256, 16), ("psa_4", 64, 16))
    pyr = []
    for f, xyz in ((1, xyz_f1), (2, xyz_f2)):
        x = xyz.permute(0, 2, 1).contiguous()
        feats = None
        levels = []
        for name, npoint, nsample in sa_cfg:
            x, feats = set_abstraction(sd, name, npoint, nsample, x, feats, taps,
                                       f"f{f}.{name}")
            levels.append((x, feats))
        pyr.append(levels)
    (x11t, p11), (x12t, p12), (x13t, p13), (x14t, p14) = pyr[0]
    (x21t, p21), (x22t, p22), (x23t, p23), (_x24t, _p24) = pyr[1]
    cf = lambda z: z.permute(0, 2, 1).contiguous()
    x11, x12, x13 = cf(x11t), cf(x12t), cf(x13t)
    x21, x22, x23 = cf(x21t), cf(x22t), cf(x23t)

    flow = cost_volume(sd, "cost_volume", 4, 32, x13, p13, x23, p23, taps, "cv3")
    x14t_ffe, emb4 = set_abstraction(sd, "flow_feature_encoding", 64, 16, x13t, flow, taps, "ffe")
    x14 = cf(x14t_ffe)

    mask4 = flow_predictor(sd, "l4_flow_predictor", p14, emb4)
    q4, t4 = pose_calculator(sd, "pose_calculator_4", emb4, F.softmax(mask4, dim=2))
    q4, t4 = q4.squeeze(2), t4.squeeze(2)
    if taps is not None:
        taps["l4.emb"], taps["l4.mask"], taps["l4.q"], taps["l4.t"] = emb4, mask4, q4, t4

    q3, t3, emb3, mask3 = pose_warp_refinement(sd, "pose_warp_refinement_3", False, x13, p13, x23,
                                               p23, x14, emb4, mask4, q4, t4, taps, "pwr3")
    q2, t2, emb2, mask2 = pose_warp_refinement(sd, "pose_warp_refinement_2", False, x12, p12, x22,
                                               p22, x13, emb3, mask3, q3, t3, taps, "pwr2")
    q1, t1, _emb1, _mask1 = pose_warp_refinement(sd, "pose_warp_refinement_1", True, x11, p11, x21,
                                                 p21, x12, emb2, mask2, q2, t2, taps, "pwr1")
    rows = [torch.cat((t, _normalise_q(q)), dim=-1).reshape(-1, 1, 7)
            for q, t in ((q1, t1), (q2, t2), (q3, t3), (q4, t4))]
    return torch.cat(rows, dim=1)


def pwclonet_loss(pose_params, gt_params, s_param):
    """slam/training/loss_modules.py:424-544 with ``with_exp_weights`` (:147-197): per level the translation
    term ``mean(sqrt((t - t_gt)^2 + 1e-10))`` and the rotation term ``mean(sqrt(sum((norm(q) - q_gt)^2) + 1e-10))``
    combined as ``l_t * exp(-s_0) + s_0 + l_q * exp(-s_1) + s_1``; levels weighted 1.6 / 0.8 / 0.4 / 0.2 from the
    coarsest (row 3) to the finest (row 0)."""
    rot_gt, trans_gt = gt_params[:, 3:], gt_params[:, :3]
    lvl = []
    for i in range(4):
        p = pose_params[:, i, :]
        q = p[:, 3:] / (torch.sqrt(torch.sum(p[:, 3:] * p[:, 3:], dim=-1, keepdim=True) + 1e-10) + 1e-10)
        l_rot = torch.mean(torch.sqrt(torch.sum((q - rot_gt) * (q - rot_gt), dim=-1, keepdim=True) + 1e-10))
        l_trans = torch.mean(torch.sqrt((p[:, :3] - trans_gt) * (p[:, :3] - trans_gt) + 1e-10))
        acc = 0.0
        for term, s in ((l_trans, s_param[0]), (l_rot, s_param[1])):
            acc = acc + (term * torch.exp(-s) + s)
        lvl.append(acc)
    return 1.6 * lvl[3] + 0.8 * lvl[2] + 0.4 * lvl[1] + 0.2 * lvl[0]


def pwclonet_train_step(sd, xyz_f1, xyz_f2, gt_params, s_init=(0.0, -2.5), dtype=torch.float32):
    """One training-mode forward + loss + backward on CPU (slam/training/trainer.py:624-628 without the optimizer
    step): batch-statistic BatchNorm (P2/pytorch_utils.py:52-83), dropout of the pose heads OFF
    (PW/pose_calculator.py:63-65 -- its random stream is device-specific; the fixtures are recorded with the four
    PoseCalculator modules in eval()).  ``sd`` is updated in place where BatchNorm does (running_mean,
    running_var, num_batches_tracked) -- for ``dtype=float32``; with ``dtype=torch.float64`` the whole step is evaluated
    in double precision on copies (FPS and neighbour lists still from the float32 coordinates, so the same lists): the
    yardstick for how much of a gradient difference is fp32 conditioning (backward through batch-statistic BatchNorm
    loses ~3 digits: tests/test_gpu_train.py).  Returns ``(pose_params, loss, {key: gradient}, grad_s)``."""
    global _BN_TRAIN
    if dtype != torch.float32:
        sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        xyz_f1, xyz_f2, gt_params = xyz_f1.to(dtype), xyz_f2.to(dtype), gt_params.to(dtype)
    work = dict(sd)
    leaves = {}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            leaves[k] = v.detach().clone().requires_grad_(True)
            work[k] = leaves[k]
    s_param = torch.tensor(list(s_init), dtype=dtype, requires_grad=True)
    _BN_TRAIN = True
    try:
        with torch.enable_grad():
            pose = _forward(work, xyz_f1, xyz_f2)
            loss = pwclonet_loss(pose, gt_params, s_param)
            loss.backward()
    finally:
        _BN_TRAIN = False
    return pose.detach(), loss.detach(), {k: v.grad for k, v in leaves.items()}, s_param.grad
