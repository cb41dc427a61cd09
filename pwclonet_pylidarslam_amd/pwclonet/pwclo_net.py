"""PWCLO-Net (``PW/pwclo_net.py:32-207``) on the MI355X operator stack.

Same constructor (``PWCLONet(config, pose)``), ``forward`` signature and ``state_dict`` keys as
the reference, so a reference checkpoint loads unchanged.  Differences are only in how the
work is scheduled:
  * ``forward`` runs the reference-shaped graph on the HIP operators (any mode, autograd ok);
  * an eval-mode ``forward`` under ``torch.no_grad()`` / ``inference_mode`` runs the fused MFMA kernels instead
    (``..fused.FusedPWCLONet``: BatchNorm folded, activations point-major, 75 launches): the weights are packed on the
    first such call (``config["fused"] = "auto"``, the default -- a user of the reference who only swaps the import gets the
    fast path; ``"off"`` keeps the module graph, ``prepare_fused(dtype=...)`` packs explicitly); ``train()``,
    ``load_state_dict()`` and ``.to()`` drop the packed weights again, in-place parameter edits re-pack;
  * ``log_dict`` is the reference's (host tensors, forces a D2H sync, pwclo_net.py:186-193) by
    default -- soft-max and norm computed on the device, their result copied (the reference copies the mask and
    computes on the host: 71 ms per batch of 32); ``log_mode="device"`` keeps the same values on the GPU without a sync and
    ``log_mode="none"`` skips them -- the benchmark states which one it used.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2_ops import pointnet2_utils
from ..pointnet2_ops.pointnet2_modules import PointnetSAModulePWCLONet
from .costvolume import CostVolume
from .flowpredictor import FlowPredictor
from .pose_calculator import PoseCalculator
from .pose_warp_refinement import PoseWarpRefinement


def _cfg(config, key, default=None):
    if hasattr(config, "get"):
        v = config.get(key, default)
        return default if v is None else v
    return getattr(config, key, default)


class LazyLogDict(dict):
    """``log_dict`` whose values are computed on first access.  The reference builds
    ``embedding_mask`` / ``point_cloud`` inside forward (pwclo_net.py:186-193, with a D2H sync);
    here the forward only keeps references and the softmax / norm / copies run when (and if)
    a logger reads them.  ``ready``: optional callable run once before the first read (graph replay:
    waits for the replay that filled the source buffers).  Every way of reading a dict goes through
    the computed values (``[]``, ``get``, ``items``, ``values``)."""

    KEYS = ("embedding_mask", "point_cloud")

    def __init__(self, mask_pm, cloud_pm, to_host, ready=None):
        super().__init__()
        self._src = (mask_pm, cloud_pm, to_host)
        self._ready = ready
        for k in self.KEYS:
            dict.__setitem__(self, k, None)

    def fresh(self, to_host=None, ready=None):
        """A new, un-cached view of the same source buffers (one per graph replay: the buffers are static,
        their contents are the latest batch's)."""
        m1, pc, th = self._src
        return LazyLogDict(m1, pc, th if to_host is None else to_host, ready)

    def __getitem__(self, key):
        if key in self.KEYS and dict.__getitem__(self, key) is None:
            if self._ready is not None:
                self._ready()
                self._ready = None
            m1, pc, to_host = self._src
            # soft-max and norm where the mask lives, THEN the copy of the (B, N1) result: the reference copies the whole
            # (B, C, N1) mask to the host first and runs both there -- 71 ms per batch of 32 (measured), same values up to
            # fp32 rounding of another soft-max implementation
            val = torch.linalg.norm(F.softmax(m1, dim=1), dim=-1, ord=2) if key == "embedding_mask" else pc
            if to_host:
                val = val.cpu()
            dict.__setitem__(self, key, val)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]


def _unit(q):
    return q / (torch.sqrt(torch.sum(q * q, dim=-1, keepdim=True) + 1e-10) + 1e-10)


class PWCLONet(nn.Module):
    """(xyz_f1 (B,3,N), None, xyz_f2 (B,3,N), None) -> (pose_params (B,4,7), log_dict).
    Rows of pose_params = pyramid levels 1 (finest) .. 4, each [tx,ty,tz,qw,qx,qy,qz]."""

    def __init__(self, config, pose=None):
        super().__init__()
        self.config = config
        self.pose = pose
        self.num_out_poses = _cfg(config, "num_out_poses", 1)
        self.num_input_channels = _cfg(config, "num_input_channels", 3)
        self.sequence_len = _cfg(config, "sequence_len", 2)
        self.device = torch.device(_cfg(config, "device", "cuda"))
        self.nb_levels = _cfg(config, "num_out_poses", 4)
        self.log_mode = _cfg(config, "log_mode", "host")
        self.fuse_mode = _cfg(config, "fused", "auto")      # "auto": pack on the first eval-mode no-grad forward; "off": never
        scalar_last = _cfg(config, "scalar_last", False)
        dev = str(_cfg(config, "device", "cuda"))

        # siamese point feature pyramid (pwclo_net.py:66-69)
        self.psa_1 = PointnetSAModulePWCLONet(npoint=2048, nsample=32, mlp=[0, 8, 8, 16], bn=True)
        self.psa_2 = PointnetSAModulePWCLONet(npoint=1024, nsample=32, mlp=[16, 16, 16, 32], bn=True)
        self.psa_3 = PointnetSAModulePWCLONet(npoint=256, nsample=16, mlp=[32, 32, 32, 64], bn=True)
        self.psa_4 = PointnetSAModulePWCLONet(npoint=64, nsample=16, mlp=[64, 64, 64, 128], bn=True)
        # attentive cost volume at level 3 + flow feature encoding (:73-78)
        self.cost_volume = CostVolume(nsample=4, nsample_q=32, in_channel1=64, in_channel2=64,
                                      mlp1=[128, 64, 64], mlp2=[128, 64])
        self.flow_feature_encoding = PointnetSAModulePWCLONet(npoint=64, nsample=16,
                                                              mlp=[64, 128, 64, 64], bn=True)
        # level-4 mask + pose head (:84-86)
        self.l4_flow_predictor = FlowPredictor(in_channel=128 + 64, mlp=[128, 64])
        self.pose_calculator_4 = PoseCalculator(in_channel=64, out_channel=256, kernel_size=1,
                                                padding="valid", activation=None, squeeze=True)
        # pose warp-refinement, levels 3..1 (:94-106)
        common = dict(in_channel_f1_prev=64, in_channel_mask=64, device=dev, scalar_last=scalar_last)
        self.pose_warp_refinement_3 = PoseWarpRefinement(in_channel_f1=64, in_channel_f2=64, radius=2.0,
                                                         last_pose_estimation=False, **common)
        self.pose_warp_refinement_2 = PoseWarpRefinement(in_channel_f1=32, in_channel_f2=32, radius=1.0,
                                                         last_pose_estimation=False, **common)
        self.pose_warp_refinement_1 = PoseWarpRefinement(in_channel_f1=16, in_channel_f2=16, radius=0.5,
                                                         last_pose_estimation=True, **common)

        self._fused = None

    # ---- fused eval-mode path ---------------------------------------------------------------------
    def prepare_fused(self, dtype=None):
        """Fold BatchNorm and pack the weights for the fused kernels (eval mode only).  ``dtype``: "f32" (default:
        fp32 MFMA, the parity path), "bf16" (BASELINE configs[4]: stack layers on v_mfma_f32_16x16x32_bf16 with weights
        and activations rounded to bf16, fp32 accumulation; coordinates, distances, FPS indices and neighbour lists
        stay fp32 / exact) or "bf16x3" (the opt-in three-term split).  The packed copy is tied
        to the parameters it was made from: ``train()``, ``load_state_dict()``, ``.to()`` / ``.cuda()`` /
        ``.float()`` (anything that goes through ``_apply``) drop it, and an in-place edit of any parameter or
        buffer (optimizer step, ``copy_``) is noticed through the tensors' version counters at the next eager
        forward, which re-packs."""
        from ..fused import FusedPWCLONet, packing_dtype
        self.eval()
        if dtype is not None:
            self._fused_dtype = dtype
        with packing_dtype(getattr(self, "_fused_dtype", None)):
            self._fused = FusedPWCLONet(self)
        self._fused_tensors = list(self.parameters()) + list(self.buffers())
        self._fused_versions = self._state_versions()
        return self

    def _state_versions(self):
        return tuple(t._version for t in self._fused_tensors)

    def _apply(self, fn, *args, **kwargs):
        self._fused = None          # packed weights live on the old device / dtype
        return super()._apply(fn, *args, **kwargs)

    def _fused_log_dict(self, inter):
        """The reference's log_dict (pwclo_net.py:186-193) from the fused path's point-major tensors."""
        if self.log_mode == "none":
            return {}
        return LazyLogDict(inter["mask1"], inter["x11"], self.log_mode == "host")

    def train(self, mode=True):
        if mode:
            self._fused = None      # packed weights would go stale
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self._fused = None
        return super().load_state_dict(*args, **kwargs)

    def _pyramid(self, xyz_t, points, samples=None):
        """``samples``: the levels' sample coordinates where the caller has already drawn them (a list, possibly shorter
        than the pyramid: the remaining levels sample for themselves)."""
        levels = []
        x, f = xyz_t, points
        for k, sa in enumerate((self.psa_1, self.psa_2, self.psa_3, self.psa_4)):
            given = samples[k] if samples is not None and k < len(samples) else None
            x, f = sa(x, f, new_xyz=given) if given is not None else sa(x, f)
            levels.append((x, f))
        return levels

    def sample_pyramid(self, xyz_f1, xyz_f2):
        """The sample coordinates the four set-abstraction levels draw for both frames, xyz (B,3,N) -> two lists of
        (B, npoint_l, 3) tensors.  They depend on the input coordinates only (no weights, no gradient), so a training
        loop can draw them for the NEXT batch while the current step runs and hand them to ``forward(...,
        samples=)`` (training.TrainStep(sample_ahead=True)): the level-1 sampler is the longest serial kernel of the
        step -- one workgroup per cloud, 1.9 ms on a quarter of the compute units."""
        cf = lambda z: z.permute(0, 2, 1).contiguous()
        with torch.no_grad():
            s1, s2 = pointnet2_utils.sample_and_gather_pair(cf(xyz_f1), cf(xyz_f2), self.psa_1.npoint)
            a, b = [s1], [s2]
            for sa in (self.psa_2, self.psa_3, self.psa_4):
                a.append(pointnet2_utils.sample_and_gather(a[-1], sa.npoint))
                b.append(pointnet2_utils.sample_and_gather(b[-1], sa.npoint))
        return a, b

    def forward(self, xyz_f1, points_f1, xyz_f2, points_f2, bn_decay=None, samples=None):
        """``samples`` (not in the reference's signature): ``sample_pyramid(xyz_f1, xyz_f2)`` when the caller has
        already drawn it -- module path only."""
        if samples is not None and not (self.training and xyz_f1.is_cuda):
            return self._forward_modules(xyz_f1, points_f1, xyz_f2, points_f2, samples)
        if self._fused is None and self.fuse_mode == "auto" and not self.training and not torch.is_grad_enabled() \
                and points_f1 is None and points_f2 is None and xyz_f1.is_cuda \
                and not torch.cuda.is_current_stream_capturing():
            self.prepare_fused()        # inference call of a drop-in user: same results within 1e-5, five times the speed
        if self._fused is not None and not self.training and not torch.is_grad_enabled() \
                and points_f1 is None and points_f2 is None:
            if not torch.cuda.is_current_stream_capturing() and self._state_versions() != self._fused_versions:
                self.prepare_fused()            # a parameter / buffer was edited in place since packing
            pose, inter = self._fused(xyz_f1, xyz_f2, return_intermediates=True)
            return pose, self._fused_log_dict(inter)
        if self.training and xyz_f1.is_cuda:
            from .. import batchnorm as _hip_bn
            with _hip_bn.deferred_counters():       # the 93 `num_batches_tracked += 1` of a forward as one multi-tensor add
                return self._forward_modules(xyz_f1, points_f1, xyz_f2, points_f2, samples)
        return self._forward_modules(xyz_f1, points_f1, xyz_f2, points_f2)

    def _forward_modules(self, xyz_f1, points_f1, xyz_f2, points_f2, samples=None):
        """The reference-shaped forward (PW/pwclo_net.py:109-207) on the HIP ops."""
        cf = lambda z: z.permute(0, 2, 1).contiguous()
        B = xyz_f1.size(0)
        if samples is not None:                  # every level's samples drawn by the caller (sample_pyramid)
            l1 = self._pyramid(cf(xyz_f1), points_f1, samples[0])
            l2 = self._pyramid(cf(xyz_f2), points_f2, samples[1])
        elif (not self.training) and points_f1 is None and points_f2 is None \
                and xyz_f1.shape == xyz_f2.shape:
            # Eval mode: the pyramid is siamese (shared weights) and every op is per-cloud with
            # BatchNorm running statistics, so both frames go through it as one batch of 2B clouds
            # -- identical results, half the launches, twice the CU fill for FPS / knn.
            both = self._pyramid(cf(torch.cat((xyz_f1, xyz_f2), dim=0)), None)
            l1 = [(x[:B], f[:B]) for x, f in both]
            l2 = [(x[B:], f[B:]) for x, f in both]
        else:
            # train mode: BatchNorm statistics are per frame, so the pyramids run one after the other -- but the first
            # level's sampling (the longest serial kernel of the step, one workgroup per cloud) is drawn for both
            # frames in one launch
            c1, c2 = cf(xyz_f1), cf(xyz_f2)
            sam1 = sam2 = None
            if xyz_f1.is_cuda and xyz_f1.shape == xyz_f2.shape and not (xyz_f1.requires_grad or xyz_f2.requires_grad):
                s1, s2 = pointnet2_utils.sample_and_gather_pair(c1, c2, self.psa_1.npoint)
                sam1, sam2 = [s1], [s2]
            l1 = self._pyramid(c1, points_f1, sam1)
            l2 = self._pyramid(c2, points_f2, sam2)
        (x11t, p11), (x12t, p12), (x13t, p13), (_x14t, p14) = l1
        (x21t, p21), (x22t, p22), (x23t, p23), _ = l2
        x11, x12, x13 = cf(x11t), cf(x12t), cf(x13t)
        x21, x22, x23 = cf(x21t), cf(x22t), cf(x23t)

        flow_embedding = self.cost_volume(x13, p13, x23, p23)
        x14t, emb4 = self.flow_feature_encoding(x13t, flow_embedding)
        x14 = cf(x14t)

        mask4 = self.l4_flow_predictor(p14, emb4)
        q4, t4 = self.pose_calculator_4(emb4, F.softmax(mask4, dim=2))

        q3, t3, emb3, mask3 = self.pose_warp_refinement_3(x13, p13, x23, p23, x14, emb4, mask4, q4, t4)
        q2, t2, emb2, mask2 = self.pose_warp_refinement_2(x12, p12, x22, p22, x13, emb3, mask3, q3, t3)
        q1, t1, _emb1, mask1 = self.pose_warp_refinement_1(x11, p11, x21, p21, x12, emb2, mask2, q2, t2)

        log_dict = {}
        if self.log_mode != "none":
            m1 = mask1.detach()
            pc = x11t.detach()
            emb = torch.linalg.norm(F.softmax(m1, dim=2).permute(0, 2, 1), dim=-1, ord=2)
            if self.log_mode == "host":  # reference behaviour: host tensors (a D2H sync inside forward) -- but the soft-max
                emb, pc = emb.cpu(), pc.cpu()   # and norm run on the device and only their (B, N1) result is copied
            log_dict = {"embedding_mask": emb, "point_cloud": pc}

        rows = [torch.cat((t, _unit(q)), dim=-1).reshape(-1, 1, 7)
                for q, t in ((q1, t1), (q2, t2), (q3, t3), (q4, t4))]
        return torch.cat(rows, dim=1), log_dict
