"""Host side of the fused eval-mode kernels: BatchNorm folding, weight packing, launch wrappers.

Kernels (csrc/fused_*.hip) keep feature tensors point-major ``(B, N, C)`` and take every layer
as one packed buffer (see csrc/mlp_core.hpp):
    [NBO][NBI][64 lanes][4]  with  Wp[o][m][lane][r] = W'[16*o + lane%16][phys(16*m + 4*(lane//16) + r)]
    followed by NBO*16 bias values,
where ``W' = W * gamma/sqrt(var+eps)``, ``bias' = beta - mean*gamma/sqrt(var+eps)`` (eval-mode
BatchNorm folded into the 1x1 convolution) and ``phys`` maps the kernel's physical input
channel order (16-channel blocks, one source per block) to the layer's original input channels.
"""
import contextlib
import os

import torch

from . import _lib


def fold_conv_bn(layer):
    """``layer``: one SharedMLP entry (pytorch_utils._ConvBlock with conv [+ bn] [+ relu]).
    Returns (W (Cout,Cin), bias (Cout,)) of the equivalent affine map in eval mode."""
    conv = layer.conv
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1).double()
    b = conv.bias.detach().double() if conv.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64,
                                                                             device=w.device)
    if hasattr(layer, "bn"):
        bn = layer.bn.bn
        s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        w = w * s[:, None]
        b = (b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()
    return w.float(), b.float()


def pack_layer(w, b, phys_map, nbo=None, kmajor_out=False):
    """Pack one folded layer.  ``phys_map``: for every physical input channel (length multiple of
    16) the original input channel, or -1 for padding.  Output channels are padded to 16*nbo.
    ``kmajor_out`` (cout <= 16): output channel c is produced on row 4*(c % 4) + c // 4, i.e. in lane group c % 4,
    accumulator register c // 4 -- the "k-step major" order in which the NEXT layer needs only ceil(cout / 4) of its
    four MFMA k-steps (its phys_map must be ``kstep_major_map(cout)``)."""
    cout, _ = w.shape
    nbi = len(phys_map) // 16
    assert len(phys_map) == 16 * nbi
    nbo = nbo or (cout + 15) // 16
    if kmajor_out:
        assert cout <= 16 and nbo == 1
        rows = torch.tensor([4 * (c % 4) + c // 4 for c in range(cout)], dtype=torch.long, device=w.device)
        w_perm = torch.zeros((16, w.shape[1]), dtype=w.dtype, device=w.device)
        b_perm = torch.zeros((16,), dtype=b.dtype, device=b.device)
        w_perm[rows], b_perm[rows] = w, b
        w, b, cout = w_perm, b_perm, 16
    pm = torch.as_tensor(phys_map, dtype=torch.long, device=w.device)
    wphys = torch.zeros((16 * nbo, 16 * nbi), dtype=torch.float32, device=w.device)
    valid = pm >= 0
    wphys[:cout, valid] = w[:, pm[valid]]
    # (o,row,m,g,r) -> (o,m,g,row,r): lane = 16*g + row
    wp = wphys.view(nbo, 16, nbi, 4, 4).permute(0, 2, 3, 1, 4).reshape(-1)
    bias = torch.zeros(16 * nbo, dtype=torch.float32, device=w.device)
    bias[:cout] = b
    return torch.cat((wp, bias)).contiguous()


WFMT_F32, WFMT_BF16X3, WFMT_BF16 = 0, 1, 2     # include/pwclo_ops.h: packed-weight format of a stack (csrc/mlp_core.hpp)
_DTYPE_WFMT = {"f32": WFMT_F32, "fp32": WFMT_F32, "float32": WFMT_F32, "bf16x3": WFMT_BF16X3, "bf16": WFMT_BF16,
               "bfloat16": WFMT_BF16}
_forced_wfmt = None


@contextlib.contextmanager
def packing_dtype(dtype):
    """``with packing_dtype("bf16"): ...``: objects packed inside use that format ("f32", "bf16x3", "bf16")."""
    global _forced_wfmt
    saved, _forced_wfmt = _forced_wfmt, (None if dtype is None else _DTYPE_WFMT[str(dtype).replace("torch.", "")])
    try:
        yield
    finally:
        _forced_wfmt = saved


def default_wfmt():
    """Format new packed objects are built in: what ``packing_dtype`` / ``prepare_fused(dtype=...)`` asked for, else
    fp32 operand tiles unless PWCLO_BF16X3=1 selects the opt-in three-term bf16 split.  Read
    ONCE per object, at pack time; the object records it (``.wfmt``) and passes it with every launch, so a later
    change cannot make a kernel index a buffer of another layout."""
    if _forced_wfmt is not None:
        return _forced_wfmt
    return WFMT_BF16X3 if os.environ.get("PWCLO_BF16X3", "0") != "0" else WFMT_F32


def _kname(name, wfmt, split_capable=True):
    """Kernel name as rocprofv3 prints it: the stack kernels carry a trailing `int FMT` template argument."""
    return name[:-1] + (", %d>" % (wfmt if split_capable else 0))


def _a2_kernel_name(kp, B, S, wfmt):
    if kp == 6 and wfmt == WFMT_F32 and os.environ.get("PWCLO_LANE6", "1") != "0" and B * ((S + 15) // 16) > 1024:
        return "cv_a2_lane6_kernel<8>"
    if kp == 6:
        return _kname("cv_a2_dense6_kernel<%d>" % (8 if B * ((S + 7) // 8) > 2048 else 4), wfmt)
    return _kname({32: "cv_a2_kernel<32, 2, 8>", 16: "cv_a2_kernel<16, 1, 16>", 8: "cv_a2_kernel<8, 1, 16>"}[kp],
                  wfmt, split_capable=kp == 32)


def pack_layer_bf3(w, b, phys_map, nbo=None):
    """Pack one folded layer for ``mlp_layer_bf3`` (csrc/mlp_core.hpp): per (o, mp) tile [split][lane][8 bf16],
    lane = 16*g + row, element t of lane group g = physical channel 16*(2*mp + t//4) + 4*g + t%4; the three
    splits are the round-to-nearest bf16 terms hi, mid, lo of every weight.  Returned as float32 storage."""
    cout, _ = w.shape
    nbi = len(phys_map) // 16
    assert len(phys_map) == 16 * nbi and nbi % 2 == 0
    nbo = nbo or (cout + 15) // 16
    pm = torch.as_tensor(phys_map, dtype=torch.long, device=w.device)
    wphys = torch.zeros((16 * nbo, 16 * nbi), dtype=torch.float32, device=w.device)
    valid = pm >= 0
    wphys[:cout, valid] = w[:, pm[valid]]
    # (o, row, mp, half, g, r) -> (o, mp, g, row, half, r): element t = 4*half + r of lane 16*g + row
    wt = wphys.view(nbo, 16, nbi // 2, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(nbo, nbi // 2, 64, 8)
    hi = wt.to(torch.bfloat16)
    r1 = wt - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    tiles = torch.stack((hi, mid, lo), dim=2).contiguous()            # (o, mp, split, lane, 8)
    packed = tiles.view(torch.int16).reshape(-1).view(torch.float32)
    bias = torch.zeros(16 * nbo, dtype=torch.float32, device=w.device)
    bias[:cout] = b
    return torch.cat((packed, bias)).contiguous()


def pack_layer_bf16(w, b, phys_map, nbo=None):
    """Pack one folded layer for ``mlp_layer_bf16`` (csrc/mlp_core.hpp): per (o, mp) tile [lane][8 bf16] in the
    element order of ``pack_layer_bf3``, every weight rounded once to bf16 (round to nearest even); fp32 bias."""
    cout, _ = w.shape
    nbi = len(phys_map) // 16
    assert len(phys_map) == 16 * nbi and nbi % 2 == 0
    nbo = nbo or (cout + 15) // 16
    pm = torch.as_tensor(phys_map, dtype=torch.long, device=w.device)
    wphys = torch.zeros((16 * nbo, 16 * nbi), dtype=torch.float32, device=w.device)
    valid = pm >= 0
    wphys[:cout, valid] = w[:, pm[valid]]
    wt = wphys.view(nbo, 16, nbi // 2, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(nbo, nbi // 2, 64, 8)
    packed = wt.to(torch.bfloat16).contiguous().view(torch.int16).reshape(-1).view(torch.float32)
    bias = torch.zeros(16 * nbo, dtype=torch.float32, device=w.device)
    bias[:cout] = b
    return torch.cat((packed, bias)).contiguous()


def chain_map(cout_prev, nb):
    """Physical->original map of a layer fed by the previous layer's (padded) output."""
    return [c if c < cout_prev else -1 for c in range(16 * nb)]


def layer_floats(nbi, nbo):
    """Floats of one packed layer (csrc/mlp_core.hpp: layer_floats)."""
    return nbo * nbi * 256 + nbo * 16


def stack_macs(shared_mlp):
    """Algorithmic multiply-accumulates per pixel of a SharedMLP (real, unpadded channel counts)."""
    return sum(int(l.conv.weight.shape[0]) * int(l.conv.weight.shape[1]) for l in shared_mlp)


def pack_stack(shared_mlp, first_map, wfmt=WFMT_F32):
    """Pack every layer of a SharedMLP whose first layer reads the physical order `first_map`.
    Returns (packed float tensor, [padded widths]).  ``wfmt`` = WFMT_BF16X3: layers with an even number of
    input blocks use the bf16x3 format (kernels built on mlp_layer_any)."""
    parts, widths = [], []
    pm = first_map
    for layer in shared_mlp:
        w, b = fold_conv_bn(layer)
        nbo = (w.shape[0] + 15) // 16
        parts.append(pack_layer_any(w, b, pm, nbo, wfmt))
        widths.append(16 * nbo)
        pm = chain_map(w.shape[0], nbo)
    return torch.cat(parts).contiguous(), widths


def _p(t):
    return t.data_ptr() if t is not None else 0


# ---- set abstraction -------------------------------------------------------------------------------

def sa_first_map(c_feat):
    """Physical input order of csrc/fused_sa.hip: block 0 = [dx,dy,dz(,qx,qy,qz)], then features.
    Original order (pointnet2_modules.py:222 / :233): [xyz_diff(3), features(C)] or
    [xyz_diff(3), grouped_xyz(3)]."""
    if c_feat == 0:
        return [0, 1, 2, 3, 4, 5] + [-1] * 10
    return [0, 1, 2] + [-1] * 13 + [3 + c for c in range(c_feat)]


class FusedSA:
    """Packed eval-mode weights of one ``PointnetSAModulePWCLONet`` + launcher."""

    def __init__(self, module):
        convs = list(module.mlp_module)
        cin = convs[0].conv.weight.shape[1]
        self.c_feat = cin - 3 if cin != 6 else 0
        assert self.c_feat % 16 == 0
        self.packed, self.widths = pack_stack(module.mlp_module, sa_first_map(self.c_feat))
        self.c_out = convs[-1].conv.weight.shape[0]
        assert self.c_out == self.widths[-1], "last layer width must be a multiple of 16"
        self.nsample = module.nsample
        self.macs = stack_macs(module.mlp_module)

    def __call__(self, xyz, new_xyz, feat_pm, idx):
        """xyz (B,N,3), new_xyz (B,S,3), feat_pm (B,N,C) point-major or None, idx (B,S,K) int32
        -> (B,S,Cout) point-major."""
        B, N, _ = xyz.shape
        S, K = idx.shape[1], idx.shape[2]
        out = torch.empty((B, S, self.c_out), dtype=torch.float32, device=xyz.device)
        _lib.annotate(family="mlp", flops=2.0 * B * S * K * self.macs,
                      bytes=4.0 * B * (S * K + S * K * (3 + self.c_feat) + 3 * S + S * self.c_out))
        _lib.call("sa_fused_kernel_wrapper", xyz.device, B, N, S, K, self.c_feat, *self.widths,
                  _p(xyz), _p(new_xyz), _p(feat_pm), _p(idx), _p(self.packed), _p(out))
        return out


# ---- FPS + sampled coordinates, point-major warp -------------------------------------------------------

FPS_CHAIN_INTS = 12      # ints per cloud of a sampling-chain record (csrc/sampling.hip)


def fps_with_xyz(xyz, npoint, tie_out=None, tie_iters=0, prefix_in=None):
    """xyz (B,N,3) -> (idx (B,npoint) int32, new_xyz (B,npoint,3)): FPS with the following
    gather_operation folded into the sampler.  ``tie_out`` / ``prefix_in`` (B, FPS_CHAIN_INTS) int32: the
    sampling-chain record of include/pwclo_ops.h (furthest_point_sampling_chain_kernel_wrapper)."""
    B, N, _ = xyz.shape
    idx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device)
    tmp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device) if N > 24576 else None
    _lib.annotate(family="fps", units=float(B) * (npoint - 1) * N, iters=npoint - 1,
                  bytes=4.0 * B * (3 * N + 4 * npoint))
    _lib.call("furthest_point_sampling_chain_kernel_wrapper", xyz.device, B, N, npoint, _p(xyz), _p(tmp),
              _p(idx), _p(new_xyz), _p(tie_out), int(tie_iters), _p(prefix_in))
    return idx, new_xyz


def fps_slab_supported(n):
    """Level-1 clouds the bucket-pruned sampler takes (csrc/sampling.hip: fps_slab_kernel): 8 x-slabs, LDS table.
    OPT-IN (PWCLO_FPS_SLAB=1): exact and tested, but measured SLOWER than the register-resident kernel at n = 8192
    (2.1 ms against 1.82 ms per level-1 call: the ~4 of 128 buckets a new sample touches still cost a serial
    update -> 16-slot arg-max -> two wave reductions chain, ~1400 cycles, where the unpruned loop's sixteen
    independent chains take ~1050)."""
    return (os.environ.get("PWCLO_FPS_SLAB", "0") != "0" and 4096 <= n <= 9216 and _lib.load().knn_point_slabs(n) == 8)


def fps_slab_with_xyz(xyz, npoint, tie_out=None, tie_iters=0):
    """``fps_with_xyz`` for the pyramid's first level: builds the cloud's neighbour-search structure first (the level's
    knn needs it anyway), samples with the slab-pruned kernel and returns the structure for ``knn_prebuilt``.
    -> (idx (B,npoint) int32, new_xyz (B,npoint,3), workspace)."""
    B, N, _ = xyz.shape
    lib = _lib.load()
    ws = torch.empty((lib.knn_point_build_bytes(B, N),), dtype=torch.uint8, device=xyz.device)
    tab = torch.empty((B, 32), dtype=torch.int32, device=xyz.device)
    status = torch.empty((B,), dtype=torch.int32, device=xyz.device)
    idx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device)
    _lib.annotate(family="knn", units=0.0, bytes=4.0 * B * 7 * N)
    _lib.call("knn_build_kernel_wrapper", xyz.device, B, N, _p(xyz), _p(ws), _p(tab))
    _lib.annotate(family="fps", units=float(B) * (npoint - 1) * N, iters=npoint - 1,
                  bytes=4.0 * B * (3 * N + 4 * npoint))
    _lib.call("furthest_point_sampling_slab_kernel_wrapper", xyz.device, B, N, npoint, _p(xyz), _p(idx), _p(new_xyz),
              _p(tie_out), int(tie_iters), _p(ws), _p(tab), _p(status))
    return idx, new_xyz, (ws, tab, status)


def knn_prebuilt(nsample, xyz, new_xyz, ws):
    """knn on a structure built by ``fps_slab_with_xyz`` (same cloud): the search pass alone."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty((B, S, nsample), dtype=torch.int32, device=xyz.device)
    _lib.annotate(family="knn", units=float(B) * S * N, bytes=4.0 * B * (3 * N + 3 * S + S * nsample))
    _lib.call("knn_point_prebuilt_kernel_wrapper", xyz.device, B, N, S, int(nsample), _p(new_xyz), _p(idx), 0, _p(ws[0]))
    return idx


def quat_warp_pm(xyz, q, t):
    """xyz (B,N,3) point-major, q (B,4), t (B,3) -> (B,N,3)."""
    B, N, _ = xyz.shape
    q = q.reshape(B, 4).contiguous()
    t = t.reshape(B, 3).contiguous()
    out = torch.empty_like(xyz)
    _lib.call("quat_warp_pm_kernel_wrapper", xyz.device, B, N, _p(xyz), _p(q), _p(t), _p(out))
    return out


def knn(nsample, xyz, new_xyz):
    from .pointnet2_ops import _ext
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    _lib.annotate(family="knn", units=float(B) * S * N, bytes=4.0 * B * (3 * N + 3 * S + S * nsample))
    return _ext.knn_point(nsample, xyz, new_xyz)


def knn_keep(nsample, xyz, new_xyz):
    """``knn`` that also hands back the search structure it built for ``xyz`` (sorted rows + block boxes of all its
    clouds), or None when the call used the exhaustive kernel: ``(idx, (workspace, clouds, n) | None)``."""
    from .pointnet2_ops import _ext
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    ws_bytes = _lib.load().knn_point_workspace_bytes(B, N) if S >= _ext._KNN_MIN_S else 0
    if ws_bytes <= 0:
        return knn(nsample, xyz, new_xyz), None
    idx = torch.empty((B, S, nsample), dtype=torch.int32, device=xyz.device)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=xyz.device)
    _lib.annotate(family="knn", units=float(B) * S * N, bytes=4.0 * B * (3 * N + 3 * S + S * nsample))
    _lib.call("knn_point_ws_kernel_wrapper", xyz.device, B, N, S, int(nsample), _p(xyz.contiguous()),
              _p(new_xyz.contiguous()), _p(idx), 0, _p(ws))
    return idx, (ws, B, N)


def knn_on(struct, first_cloud, nsample, new_xyz):
    """Search pass alone on clouds [first_cloud, first_cloud + B) of a structure ``knn_keep`` returned (B = new_xyz's
    batch): the same lists as ``knn(nsample, those clouds, new_xyz)``, without sorting them again."""
    ws, built_b, N = struct
    B, S, _ = new_xyz.shape
    idx = torch.empty((B, S, nsample), dtype=torch.int32, device=new_xyz.device)
    _lib.annotate(family="knn", units=float(B) * S * N, bytes=4.0 * B * (3 * N + 3 * S + S * nsample))
    _lib.call("knn_point_prebuilt_slice_kernel_wrapper", new_xyz.device, B, N, S, int(nsample), _p(new_xyz.contiguous()),
              _p(idx), 0, _p(ws), int(first_cloud), int(built_b))
    return idx


# ---- point-wise MLP over concatenated sources ----------------------------------------------------------

class FusedPointwise:
    """SharedMLP (1 or 2 layers) over cat(sources, dim=channels); sources point-major (B,S,Ci)."""

    def __init__(self, shared_mlp, source_channels):
        self.src_c = list(source_channels) + [0] * (3 - len(source_channels))
        cin = sum(source_channels)
        assert all(c % 16 == 0 for c in source_channels)
        assert list(shared_mlp)[0].conv.weight.shape[1] == cin
        self.packed, widths = pack_stack(shared_mlp, list(range(cin)))
        assert len(widths) in (1, 2)
        self.w1, self.w2 = widths[0], (widths[1] if len(widths) == 2 else 0)
        self.c_out = widths[-1]
        self.macs, self.c_in = stack_macs(shared_mlp), cin

    def __call__(self, *sources):
        B, S, _ = sources[0].shape
        src = list(sources) + [None] * (3 - len(sources))
        out = torch.empty((B, S, self.c_out), dtype=torch.float32, device=sources[0].device)
        _lib.annotate(family="mlp", flops=2.0 * B * S * self.macs, bytes=4.0 * B * S * (self.c_in + self.c_out))
        _lib.call("pointwise_fused_kernel_wrapper", out.device, B, S, *self.src_c, self.w1, self.w2,
                  _p(src[0]), _p(src[1]), _p(src[2]), _p(self.packed), _p(out))
        return out

    # source layouts for which csrc/fused_layers.hip has a stack + linear-tail kernel (stack and tail fit the 160 KiB of LDS)
    TAIL_CASES = {(32, 64, 64, 128, 64), (64, 64, 32, 128, 64)}

    def tail_supported(self, job):
        return ((*self.src_c, self.w1, self.w2) in self.TAIL_CASES and job.cin == self.c_out and job.cout == 128
                and not job.out_bf16)

    def with_tail(self, job, *sources):
        """``(self(*sources), run_linear_jobs([(job, out)])[0])`` as ONE launch: `job` (a consumer's hoisted partial
        product, e.g. the next refinement level's set-upconv seeds) runs as a third layer on the output in registers."""
        B, S, _ = sources[0].shape
        src = list(sources) + [None] * (3 - len(sources))
        out = torch.empty((B, S, self.c_out), dtype=torch.float32, device=sources[0].device)
        tail = torch.empty((B, S, job.cout), dtype=torch.float32, device=sources[0].device)
        _lib.annotate(family="mlp", flops=2.0 * B * S * (self.macs + job.cin * job.cout),
                      bytes=4.0 * B * S * (self.c_in + self.c_out + job.cout))
        _lib.call("pointwise_tail_fused_kernel_wrapper", out.device, B, S, *self.src_c, self.w1, self.w2, job.cout,
                  _p(src[0]), _p(src[1]), _p(src[2]), _p(self.packed), _p(job.packed), _p(out), _p(tail),
                  job.packed.numel())
        return out, tail


# ---- set-upconv ----------------------------------------------------------------------------------------

class FusedUpconv:
    """``PointnetFPModulePWCLONet`` (knn=True, nsample<=8, 64-channel coarse features)."""

    def __init__(self, module):
        assert module.knn and module.use_xyz
        c1 = list(module.mlp)[0].conv.weight.shape[1] - 3
        assert c1 == 64, "set-upconv kernel is built for 64-channel coarse features"
        # original order (pointnet2_modules.py:490): [grouped feat (64), xyz_diff (3)]
        first = list(range(64)) + [64, 65, 66] + [-1] * 13
        self.packed, widths = pack_stack(module.mlp, first)
        assert widths == [128, 64]
        c2 = list(module.post_mlp)[0].conv.weight.shape[1] - 64
        self.post = FusedPointwise(module.post_mlp, [64, c2])
        self.nsample = module.nsample
        self.macs = stack_macs(module.mlp)

    def pooled(self, xyz2, xyz1, feat1, idx):
        B, S, _ = xyz2.shape
        N = xyz1.shape[1]
        out = torch.empty((B, S, 64), dtype=torch.float32, device=xyz2.device)
        K = idx.shape[2]
        _lib.annotate(family="mlp", flops=2.0 * B * S * K * self.macs,
                      bytes=4.0 * B * (S * K + S * K * 67 + 3 * S + 64 * S))
        _lib.call("upconv_fused_kernel_wrapper", xyz2.device, B, N, S, idx.shape[2], _p(xyz2), _p(xyz1),
                  _p(feat1), _p(idx), _p(self.packed), _p(out))
        return out

    def __call__(self, xyz2, xyz1, feat2, feat1, idx):
        """xyz2 (B,S,3) fine, xyz1 (B,N,3) coarse, feat2 (B,S,C2), feat1 (B,N,64), idx (B,S,K)
        = knn(K, xyz1, xyz2) -> (B,S,64)."""
        return self.post(self.pooled(xyz2, xyz1, feat1, idx), feat2)


# ---- attentive cost volume -----------------------------------------------------------------------------

class FusedCostVolume:
    """``CostVolume`` with in_channel1 == in_channel2 in {16,32,64}, mlp1=[128,64,64], mlp2=[128,64]."""

    def __init__(self, module):
        c1, c2, _ = module.in_channel
        assert c1 == c2 and c1 in (16, 32, 64)
        self.c = c1
        self.nsample, self.nsample_q = module.nsample, module.nsample_q
        self.kp = cv_pix_slots(module.nsample_q)               # per-pixel buffer layout, fixed at pack time
        self.wfmt_a2 = default_wfmt() if self.kp in (6, 32) else WFMT_F32
        if self.wfmt_a2 == WFMT_BF16:
            self.wfmt_a2 = WFMT_F32        # the un-hoisted cv_a1 writes fp32 per-pixel features; bf16 rows belong to the hoisted set
        geo = list(range(10)) + [-1] * 6
        # mlp_convs input (costvolume.py:105-110): [geometry10, feat1 (C), feat2 gathered (C)]
        self.w_a1, w = pack_stack(module.mlp_convs, geo + [10 + c for c in range(2 * c1)])
        assert w == [128, 64, 64]
        wx1, wd = pack_stack(module.mlp_conv_xyz_1, geo)
        w2, wd2 = pack_stack(module.mlp2_convs, list(range(128)), self.wfmt_a2)     # [enc (64) | feat (64)], :133
        assert wd == [64] and wd2 == [128, 64]
        self.w_a2 = torch.cat((wx1, w2)).contiguous()
        wx2, _ = pack_stack(module.mlp_conv_xyz_2, geo)
        w3, wd3 = pack_stack(module.mlp3_convs, list(range(128 + c1)))  # [enc2 | feat1 | first], :176
        assert wd3 == [128, 64]
        self.w_b = torch.cat((wx2, w3)).contiguous()
        self.macs_a1 = stack_macs(module.mlp_convs)
        self.macs_a2 = stack_macs(module.mlp_conv_xyz_1) + stack_macs(module.mlp2_convs)
        self.macs_b = stack_macs(module.mlp_conv_xyz_2) + stack_macs(module.mlp3_convs)

    def __call__(self, xyz1, feat1, xyz2, feat2, idx_q=None, idx=None, taps=None, tap=""):
        """xyz1 (B,S,3) (warped) frame-1 points, feat1 (B,S,C), xyz2 (B,N,3), feat2 (B,N,C),
        all point-major -> (B,S,64).  ``taps``: dict that receives the two neighbour lists."""
        B, S, _ = xyz1.shape
        N = xyz2.shape[1]
        dev = xyz1.device
        kq, k = self.nsample_q, self.nsample
        if idx_q is None:
            idx_q = knn(kq, xyz2, xyz1)
        kp = self.kp
        pix = torch.empty((B, S * kp, 64), dtype=torch.float32, device=dev)
        c = self.c
        _lib.annotate(family="mlp", flops=2.0 * B * S * kq * self.macs_a1,
                      bytes=4.0 * B * (S * kq * (1 + 3 + c + 64) + S * (3 + c)))
        _lib.call("cv_fused_a1_kernel_wrapper", dev, B, N, S, kq, self.c, _p(xyz1), _p(feat1), _p(xyz2),
                  _p(feat2), _p(idx_q), _p(self.w_a1), _p(pix), kp)
        first = torch.empty((B, S, 64), dtype=torch.float32, device=dev)
        _lib.annotate(family="mlp", kernel=_a2_kernel_name(kp, B, S, self.wfmt_a2),
                      flops=2.0 * B * S * kq * self.macs_a2,
                      bytes=4.0 * B * (S * kq * (1 + 3 + 64) + S * (3 + 64)))
        _lib.call("cv_fused_a2_kernel_wrapper", dev, B, N, S, kq, _p(xyz1), _p(xyz2), _p(idx_q),
                  _p(self.w_a2), _p(pix), _p(first), kp, self.wfmt_a2, self.w_a2.numel())
        if idx is None:
            idx = knn(k, xyz1, xyz1)
        if taps is not None:
            taps[tap + ".idx_q"], taps[tap + ".idx"] = idx_q, idx
        out = torch.empty((B, S, 64), dtype=torch.float32, device=dev)
        _lib.annotate(family="mlp", flops=2.0 * B * S * k * self.macs_b,
                      bytes=4.0 * B * (S * k * (1 + 3 + 64) + S * (3 + c + 64)))
        _lib.call("cv_fused_b_kernel_wrapper", dev, B, S, k, self.c, _p(xyz1), _p(feat1), _p(first),
                  _p(idx), _p(self.w_b), _p(out))
        return out


def masked_pool(emb, mask):
    """emb, mask (B,N,64) point-major -> (B,64) = sum_n emb * softmax_n(mask)."""
    B, N, C = emb.shape
    assert C == 64 and mask.shape == emb.shape
    out = torch.empty((B, 64), dtype=torch.float32, device=emb.device)
    _lib.call("masked_pool_kernel_wrapper", emb.device, B, N, _p(emb), _p(mask), _p(out))
    return out


class FusedPoseHead:
    """``PoseCalculator`` (+ the pose composition of the refinement levels) in one launch."""

    def __init__(self, module):
        g = lambda blk: (blk.conv.weight.detach().squeeze(-1).contiguous(), blk.conv.bias.detach().contiguous())
        self.w_qt, self.b_qt = g(module.conv1d_q_t)
        self.w_q, self.b_q = g(module.conv1d_q)
        self.w_t, self.b_t = g(module.conv1d_t)

    def __call__(self, emb, mask, pose_params, level_row, q_prev=None, t_prev=None, warp_next=None):
        """emb, mask (B,N,64); pose_params (B,4,7) output buffer, `level_row` = which row to fill;
        q_prev (B,4) / t_prev (B,3) = coarse pose to refine (None at level 4).  -> q (B,4), t (B,3).
        ``warp_next`` (B,M,3): also returns quat_warp_pm(warp_next, q, t), computed by the same launch."""
        B, N, _ = emb.shape
        q = torch.empty((B, 4), dtype=torch.float32, device=emb.device)
        t = torch.empty((B, 3), dtype=torch.float32, device=emb.device)
        row = pose_params.data_ptr() + 4 * 7 * level_row
        args = (B, N, _p(emb), _p(mask), _p(self.w_qt), _p(self.b_qt), _p(self.w_q), _p(self.b_q), _p(self.w_t),
                _p(self.b_t), _p(q_prev), _p(t_prev), _p(q), _p(t), row, 28)
        if warp_next is None:
            _lib.call("pose_head_fused_kernel_wrapper", emb.device, *args)
            return q, t
        warp_next = warp_next.contiguous()
        warped = torch.empty_like(warp_next)
        _lib.call("pose_head_warp_fused_kernel_wrapper", emb.device, *args, warp_next.shape[1], _p(warp_next), _p(warped))
        return q, t, warped


# ---- hoisted first layers (csrc/fused_hoisted.hip) ------------------------------------------------------

class LinearJob:
    """out = src . W^T + bias over points (no activation): one hoisted partial product.  ``out_bf16``: the rows are
    written (and later gathered) as bf16 -- set by consumers packed with dtype "bf16" (csrc/mlp_core.hpp)."""

    def __init__(self, w, b, out_bf16=False):
        self.cout, self.cin = w.shape
        assert self.cin in (16, 32, 64) and self.cout in (16, 32, 64, 128)
        self.packed = pack_layer(w, b, list(range(self.cin)))
        self.out_bf16 = bool(out_bf16)


def run_linear_jobs(jobs):
    """jobs: list of (LinearJob, src (B,N,cin) contiguous) -> list of (B,N,cout) tensors, one launch
    per <= 8 jobs."""
    import ctypes
    outs = []
    for start in range(0, len(jobs), 8):
        chunk = jobs[start:start + 8]
        n = len(chunk)
        srcs = [s_ for _, s_ in chunk]
        res = [torch.empty(s_.shape[:-1] + (j.cout,), dtype=torch.bfloat16 if j.out_bf16 else torch.float32,
                           device=s_.device) for j, s_ in chunk]
        npts = [s_.shape[0] * s_.shape[1] for s_ in srcs]
        ia = lambda v: (ctypes.c_int * n)(*v)
        pa = lambda v: (ctypes.c_void_p * n)(*v)
        _lib.annotate(family="mlp", flops=2.0 * sum(p_ * j.cin * j.cout for p_, (j, _) in zip(npts, chunk)),
                      bytes=4.0 * sum(p_ * (j.cin + j.cout) for p_, (j, _) in zip(npts, chunk)))
        _lib.call("linear_jobs_kernel_wrapper", srcs[0].device, n, ia(npts), ia([j.cin for j, _ in chunk]),
                  ia([j.cout for j, _ in chunk]), pa([_p(s_) for s_ in srcs]),
                  pa([_p(j.packed) for j, _ in chunk]), pa([_p(r) for r in res]),
                  ia([int(j.out_bf16) for j, _ in chunk]))
        outs.extend(res)
    return outs


def cv_pix_slots(k):
    """Neighbour slots per query of cv_a1's per-pixel buffer: the caller's choice, passed to cv_a1 and cv_a2 as
    `pix_slots` (csrc/mlp_core.hpp: cv_pix_slots_valid).  PWCLO_DENSE6=0 selects the padded layout for K = 6."""
    if k == 6 and os.environ.get("PWCLO_DENSE6", "1") != "0":
        return 6
    return 32 if k > 16 else (16 if k > 8 else 8)


def kstep_major_map(n):
    """Physical order of a lone geometry block for the hoisted kernels: logical channel c at lane
    group c % 4, component c // 4 (position 4 * (c % 4) + c // 4), so only ceil(n / 4) MFMA k-steps
    carry data (csrc/fused_hoisted.hip: geometry_block_h / diff_block_h)."""
    out = []
    for pos in range(16):
        c = 4 * (pos % 4) + pos // 4
        out.append(c if c < n else -1)
    return out


def _zeros_like_bias(w):
    return torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)


def pack_layer_any(w, b, phys_map, nbo=None, wfmt=WFMT_F32):
    """``pack_layer_bf3`` for the split format when the layer has an even number of 16-channel input
    blocks (csrc/mlp_core.hpp: mlp_layer_any / layer_floats_any), ``pack_layer`` otherwise."""
    if wfmt == WFMT_BF16X3 and (len(phys_map) // 16) % 2 == 0:
        return pack_layer_bf3(w, b, phys_map, nbo)
    if wfmt == WFMT_BF16 and (len(phys_map) // 16) % 2 == 0:
        return pack_layer_bf16(w, b, phys_map, nbo)
    return pack_layer(w, b, phys_map, nbo)


def _pack_rest(layers, cout_prev, wfmt):
    """Pack layers 2.. of a stack fed by a previous layer with `cout_prev` real outputs."""
    parts, widths = [], []
    for layer in layers:
        w, b = fold_conv_bn(layer)
        nbo = (w.shape[0] + 15) // 16
        parts.append(pack_layer_any(w, b, chain_map(cout_prev, (cout_prev + 15) // 16), nbo, wfmt))
        widths.append(16 * nbo)
        cout_prev = w.shape[0]
    return parts, widths


class FusedSAHoisted:
    """``PointnetSAModulePWCLONet`` with the feature part of layer 1 hoisted to a per-point map."""

    def __init__(self, module):
        layers = list(module.mlp_module)
        w1, b1 = fold_conv_bn(layers[0])
        cin = w1.shape[1]
        self.c_feat = cin - 3 if cin != 6 else 0
        nbo1 = (w1.shape[0] + 15) // 16
        self.wfmt = default_wfmt()
        if self.c_feat:
            # original order [xyz_diff(3), feat(C)] (pointnet2_modules.py:222)
            self.pre_job = LinearJob(_pad_rows(w1[:, 3:], 16 * nbo1), _pad_rows(b1, 16 * nbo1),
                                     out_bf16=self.wfmt == WFMT_BF16)
            first = pack_layer(w1[:, :3], _zeros_like_bias(w1), kstep_major_map(3), nbo1)
        else:
            self.pre_job = None
        # level 0 (6 -> 8 -> 8 -> 16): 8-channel layers on 16-wide MFMA blocks.  Producing them k-step major lets the
        # consuming layer skip the two k-steps that would multiply padding: 2 + 2 + 2 MFMAs per block instead of 2 + 4 + 4.
        w2 = layers[1].conv.weight.shape[0]
        self.kmajor = int(self.c_feat == 0 and len(layers) == 3 and w1.shape[0] <= 8 and w2 <= 8
                          and os.environ.get("PWCLO_SA_KMAJOR", "1") != "0")
        if self.kmajor:
            first = pack_layer(w1, b1, kstep_major_map(6), nbo1, kmajor_out=True)
            wb2, wb3 = fold_conv_bn(layers[1]), fold_conv_bn(layers[2])
            second = pack_layer(wb2[0], wb2[1], kstep_major_map(w1.shape[0]), 1, kmajor_out=True)
            third = pack_layer(wb3[0], wb3[1], kstep_major_map(w2), (wb3[0].shape[0] + 15) // 16)
            rest, widths = [second, third], [16, 16 * ((wb3[0].shape[0] + 15) // 16)]
        else:
            if not self.c_feat:
                first = pack_layer(w1, b1, kstep_major_map(6), nbo1)
            rest, widths = _pack_rest(layers[1:], w1.shape[0], self.wfmt)
        self.packed = torch.cat([first] + rest).contiguous()
        self.widths = [16 * nbo1] + widths
        self.c_out = layers[-1].conv.weight.shape[0]
        self.macs = stack_macs(module.mlp_module) - self.c_feat * w1.shape[0]   # per pixel, after hoisting
        self.nsample = module.nsample

    def jobs(self, feat_pm):
        return [(self.pre_job, feat_pm)] if self.pre_job is not None else []

    def __call__(self, xyz, new_xyz, pre, idx):
        B, N, _ = xyz.shape
        S, K = idx.shape[1], idx.shape[2]
        out = torch.empty((B, S, self.c_out), dtype=torch.float32, device=xyz.device)
        _lib.annotate(family="mlp", flops=2.0 * B * S * K * self.macs,
                      bytes=4.0 * B * (S * K * (1 + 3 + self.widths[0]) + 3 * S + S * self.c_out))
        _lib.call("sa_fused_h_kernel_wrapper", xyz.device, B, N, S, K, *self.widths, _p(xyz), _p(new_xyz),
                  _p(pre), _p(idx), _p(self.packed), _p(out), self.wfmt, self.packed.numel(), self.kmajor)
        return out


def _pad_rows(t, rows):
    """Zero-pad a (cout, ...) weight / (cout,) bias to `rows` output channels."""
    if t.shape[0] == rows:
        return t
    pad = torch.zeros((rows - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat((t, pad), dim=0)


class FusedUpconvHoisted:
    """``PointnetFPModulePWCLONet`` (knn branch): layer 1 = W_feat.feat1[n] (hoisted) + W_diff.diff."""

    def __init__(self, module):
        assert module.knn and module.use_xyz
        layers = list(module.mlp)
        w1, b1 = fold_conv_bn(layers[0])
        assert w1.shape == (128, 67), "set-upconv kernel is built for 64-channel coarse features"
        self.wfmt = default_wfmt()
        self.pre_job = LinearJob(w1[:, :64], b1, out_bf16=self.wfmt == WFMT_BF16)   # original order [feat(64), diff(3)], :490
        first = pack_layer(w1[:, 64:67], _zeros_like_bias(w1), kstep_major_map(3), 8)
        rest, widths = _pack_rest(layers[1:], 128, self.wfmt)
        assert widths == [64]
        self.packed = torch.cat([first] + rest).contiguous()
        c2 = list(module.post_mlp)[0].conv.weight.shape[1] - 64
        self.post = FusedPointwise(module.post_mlp, [64, c2])
        self.macs = stack_macs(module.mlp)

    def jobs(self, feat1):
        return [(self.pre_job, feat1)]

    def __call__(self, xyz2, xyz1, feat2, pre, idx):
        B, S, _ = xyz2.shape
        N, K = xyz1.shape[1], idx.shape[2]
        pooled = torch.empty((B, S, 64), dtype=torch.float32, device=xyz2.device)
        lane = (self.wfmt == WFMT_F32 and os.environ.get("PWCLO_LANE_UP", "1") != "0" and B * ((S + 15) // 16) > 2048)
        _lib.annotate(family="mlp", kernel="upconv_lane_kernel<16>" if lane else _kname("upconv_h_kernel<8, 1, 16>", self.wfmt),
                      flops=2.0 * B * S * K * (self.macs - 64 * 128),
                      bytes=4.0 * B * (S * K * (1 + 3 + 128) + 3 * S + 64 * S))
        _lib.call("upconv_fused_h_kernel_wrapper", xyz2.device, B, N, S, K, _p(xyz2), _p(xyz1), _p(pre),
                  _p(idx), _p(self.packed), _p(pooled), self.wfmt, self.packed.numel())
        return self.post(pooled, feat2)


def upconv_post_supported(ups, B, S):
    """The one-launch form (csrc/fused_hoisted.hip: upconv_lane_post_kernel) exists for fp32 tiles and pays once the
    jobs' 16-query tiles fill the chip's wave slots at least once (refinement levels 2 and 1 at batch 32)."""
    return (os.environ.get("PWCLO_UP_POST", "1") != "0" and all(u.wfmt == WFMT_F32 for u in ups)
            and len(ups) * B * ((S + 15) // 16) >= int(os.environ.get("PWCLO_UP_POST_MIN", "512")))


def run_upconv_post(ups, xyz2, xyz1, feat2, pres, idx):
    """``[u(xyz2, xyz1, feat2, pre, idx) for u, pre in zip(ups, pres)]`` for 1-2 ``FusedUpconvHoisted`` that share every
    input but the hoisted coarse rows: one launch, post-MLP included."""
    import ctypes
    B, S, _ = xyz2.shape
    N, K = xyz1.shape[1], idx.shape[2]
    n = len(ups)
    c2 = feat2.shape[2]
    outs = [torch.empty((B, S, 64), dtype=torch.float32, device=xyz2.device) for _ in ups]
    pa = lambda v: (ctypes.c_void_p * n)(*v)
    tiles = n * B * ((S + 15) // 16)
    _lib.annotate(family="mlp", kernel="upconv_lane_post_kernel<%d, %d>" % (c2 // 16, 16 if tiles >= 4096 else 8 if tiles >= 2048 else 4),
                  flops=sum(2.0 * B * S * (K * (u.macs - 64 * 128) + u.post.macs) for u in ups),
                  bytes=4.0 * n * B * (S * K * (1 + 3 + 128) + 3 * S + S * c2 + 64 * S))
    _lib.call("upconv_post_fused_h_kernel_wrapper", xyz2.device, n, B, N, S, K, c2, _p(xyz2), _p(xyz1), _p(idx), _p(feat2),
              pa([_p(t) for t in pres]), pa([_p(u.packed) for u in ups]), pa([_p(u.post.packed) for u in ups]),
              pa([_p(o) for o in outs]), ups[0].packed.numel(), ups[0].post.packed.numel())
    return outs


class FusedCostVolumeHoisted:
    """``CostVolume``: centre / neighbour feature parts of mlp_convs[0] and mlp3_convs[0] hoisted."""

    def __init__(self, module):
        c1, c2, _ = module.in_channel
        assert c1 == c2 and c1 in (16, 32, 64)
        self.c = c = c1
        self.nsample, self.nsample_q = module.nsample, module.nsample_q
        self.kp = cv_pix_slots(module.nsample_q)               # per-pixel buffer layout, fixed at pack time
        self.wfmt = default_wfmt()
        if self.kp not in (6, 32) and self.wfmt != WFMT_F32:
            self.wfmt = WFMT_F32                               # cv_a2 has reduced formats for 6 / 32 pixel slots only
        self.wfmt_a2 = self.wfmt
        h16 = self.wfmt == WFMT_BF16                           # hoisted rows and the per-pixel buffer stored as bf16
        geo = list(range(10)) + [-1] * 6
        la = list(module.mlp_convs)
        w1, b1 = fold_conv_bn(la[0])                                   # [geo(10) | feat1 (C) | feat2 (C)]
        self.job_u = LinearJob(w1[:, 10:10 + c], b1, out_bf16=h16)
        self.job_v = LinearJob(w1[:, 10 + c:10 + 2 * c], _zeros_like_bias(w1), out_bf16=h16)
        first = pack_layer(w1[:, :10], _zeros_like_bias(w1), kstep_major_map(10), 8)
        rest, widths = _pack_rest(la[1:], 128, self.wfmt)
        assert widths == [64, 64]
        self.w_a1 = torch.cat([first] + rest).contiguous()
        wx1, wd = pack_stack(module.mlp_conv_xyz_1, geo)
        w2, wd2 = pack_stack(module.mlp2_convs, list(range(128)), self.wfmt_a2)
        assert wd == [64] and wd2 == [128, 64]
        self.w_a2 = torch.cat((wx1, w2)).contiguous()
        wx2, _ = pack_stack(module.mlp_conv_xyz_2, kstep_major_map(10))
        lb = list(module.mlp3_convs)
        w3, b3 = fold_conv_bn(lb[0])                                   # [enc2 (64) | feat1 (C) | first (64)]
        self.job_u2 = LinearJob(w3[:, 64:64 + c], b3, out_bf16=h16)
        self.job_v2 = LinearJob(w3[:, 64 + c:], _zeros_like_bias(w3), out_bf16=h16)
        first_b = pack_layer_any(w3[:, :64], _zeros_like_bias(w3), list(range(64)), 8, self.wfmt)
        rest_b, wdb = _pack_rest(lb[1:], 128, self.wfmt)
        assert wdb == [64]
        self.w_b = torch.cat([wx2, first_b] + rest_b).contiguous()
        self.macs_a1 = stack_macs(module.mlp_convs)
        self.macs_a2 = stack_macs(module.mlp_conv_xyz_1) + stack_macs(module.mlp2_convs)
        self.macs_b = stack_macs(module.mlp_conv_xyz_2) + stack_macs(module.mlp3_convs)

    def jobs(self, feat1, feat2):
        """The three partial products that only need the inputs: u, v (first aggregate), u2."""
        return [(self.job_u, feat1), (self.job_v, feat2), (self.job_u2, feat1)]

    def __call__(self, xyz1, xyz2, u, v, u2, idx_q=None, idx=None, taps=None, tap=""):
        B, S, _ = xyz1.shape
        N = xyz2.shape[1]
        dev = xyz1.device
        kq, k, c = self.nsample_q, self.nsample, self.c
        if idx_q is None:
            idx_q = knn(kq, xyz2, xyz1)
        kp = self.kp
        merged = (kp == 6 and kq == 6 and self.wfmt == WFMT_F32 and self.wfmt_a2 == WFMT_F32
                  and B * ((S + 15) // 16) >= int(os.environ.get("PWCLO_CV_MERGED_MIN", "512"))
                  and os.environ.get("PWCLO_LANE6", "1") != "0" and os.environ.get("PWCLO_CV_MERGED", "1") != "0")
        if merged:
            return self._merged(xyz1, xyz2, u, v, u2, idx_q, idx, taps, tap)
        pix = torch.empty((B, S * kp, 64), dtype=torch.bfloat16 if self.wfmt == WFMT_BF16 else torch.float32, device=dev)
        _lib.annotate(family="mlp", kernel=_kname("cv_a1_h_kernel<%d, 1, 16>" % kp, self.wfmt),
                      flops=2.0 * B * S * kq * (self.macs_a1 - 2 * c * 128),
                      bytes=4.0 * B * (S * kq * (1 + 3 + 128 + 64) + S * (3 + 128)))
        _lib.call("cv_fused_a1_h_kernel_wrapper", dev, B, N, S, kq, _p(xyz1), _p(u), _p(xyz2), _p(v), _p(idx_q),
                  _p(self.w_a1), _p(pix), kp, self.wfmt, self.w_a1.numel())
        first = torch.empty((B, S, 64), dtype=torch.float32, device=dev)
        _lib.annotate(family="mlp", kernel=_a2_kernel_name(kp, B, S, self.wfmt_a2),
                      flops=2.0 * B * S * kq * self.macs_a2,
                      bytes=4.0 * B * (S * kq * (1 + 3 + 64) + S * (3 + 64)))
        _lib.call("cv_fused_a2_kernel_wrapper", dev, B, N, S, kq, _p(xyz1), _p(xyz2), _p(idx_q),
                  _p(self.w_a2), _p(pix), _p(first), kp, self.wfmt_a2, self.w_a2.numel())
        if idx is None:
            idx = knn(k, xyz1, xyz1)
        if taps is not None:
            taps[tap + ".idx_q"], taps[tap + ".idx"] = idx_q, idx
        (v2,) = run_linear_jobs([(self.job_v2, first)])
        return self._second(xyz1, u2, v2, first, idx)

    def _merged(self, xyz1, xyz2, u, v, u2, idx_q, idx, taps, tap):
        """First aggregate as ONE kernel (csrc/fused_hoisted.hip: cv_a_lane6_kernel; refinement levels 2 and 1 at batch
        32): no per-pixel buffer, and -- unless PWCLO_CV_V2=0 -- cv_b's neighbour partial product v2 from its epilogue."""
        B, S, _ = xyz1.shape
        N, dev, k = xyz2.shape[1], xyz1.device, self.nsample
        fold_v2 = os.environ.get("PWCLO_CV_V2", "1") != "0" and not self.job_v2.out_bf16
        first = torch.empty((B, S, 64), dtype=torch.float32, device=dev)
        v2 = torch.empty((B, S, 128), dtype=torch.float32, device=dev) if fold_v2 else None
        _lib.annotate(family="mlp", kernel="cv_a_lane6_kernel<%d, %s>" % (4 if fold_v2 and B * ((S + 15) // 16) <= 1024 else 8,
                                                                          "true" if fold_v2 else "false"),
                      flops=2.0 * B * S * (6 * (self.macs_a1 - 2 * self.c * 128 + self.macs_a2) + (64 * 128 if fold_v2 else 0)),
                      bytes=4.0 * B * (S * 6 * (1 + 3 + 128) + S * (3 + 3 * 128 + 64 + (128 if fold_v2 else 0))))
        _lib.call("cv_fused_a_lane6_kernel_wrapper", dev, B, N, S, _p(xyz1), _p(u), _p(xyz2), _p(v), _p(idx_q),
                  _p(self.w_a1), _p(self.w_a2), _p(self.job_v2.packed) if fold_v2 else 0, _p(first), _p(v2),
                  self.w_a1.numel(), self.w_a2.numel(), self.job_v2.packed.numel() if fold_v2 else 0)
        if idx is None:
            idx = knn(k, xyz1, xyz1)
        if taps is not None:
            taps[tap + ".idx_q"], taps[tap + ".idx"] = idx_q, idx
        if not fold_v2:
            (v2,) = run_linear_jobs([(self.job_v2, first)])
        return self._second(xyz1, u2, v2, first, idx)

    def _second(self, xyz1, u2, v2, first, idx):
        B, S, _ = xyz1.shape
        dev, k, c = xyz1.device, self.nsample, self.c
        out = torch.empty((B, S, 64), dtype=torch.float32, device=dev)
        _lib.annotate(family="mlp", kernel=_kname("cv_b_h_kernel<4, 1, %d>" % (4 if B * ((S * 4 + 15) // 16) <= int(os.environ.get("PWCLO_COARSE_W4_TILES", "2047")) else 16), self.wfmt),
                      flops=2.0 * B * S * k * (self.macs_b - (c + 64) * 128),
                      bytes=4.0 * B * (S * k * (1 + 3 + 128 + 64) + S * (3 + 128 + 64)))
        _lib.call("cv_fused_b_h_kernel_wrapper", dev, B, S, k, _p(xyz1), _p(u2), _p(v2), _p(first), _p(idx),
                  _p(self.w_b), _p(out), self.wfmt, self.w_b.numel())
        return out


# ---- whole network --------------------------------------------------------------------------------------

class _Branches:
    """Fork/join helper for the captured forward: side streams + events become graph edges, so
    independent branches (the FPS chain vs. the per-level neighbour search / MLPs; the two
    set-upconvs vs. the warp -> cost-volume chain) may overlap on the GPU.  Only used while a
    hipGraph is being captured: every tensor created meanwhile is kept alive until the forward
    returns, so the graph's memory pool cannot hand a buffer of one branch to another."""

    def __init__(self, device, enabled):
        self.on = enabled
        self.main = torch.cuda.current_stream(device)
        self.side = [torch.cuda.Stream(device=device) for _ in range(3)] if enabled else []
        self.keep = []

    def hold(self, *tensors):
        if self.on:
            self.keep.extend(tensors)
        return tensors[0] if len(tensors) == 1 else tensors

    def fork(self, k):
        """Context: run the body on side stream k, after everything queued on main so far."""
        if not self.on:
            return contextlib.nullcontext()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side[k].wait_event(ev)
        return torch.cuda.stream(self.side[k])

    def mark(self, k):
        """Event at the current tail of side stream k (or None when branching is off)."""
        if not self.on:
            return None
        ev = torch.cuda.Event()
        ev.record(self.side[k])
        return ev

    def wait(self, ev):
        if ev is not None:
            self.main.wait_event(ev)

    def join(self, k):
        self.wait(self.mark(k))


class FusedPWCLONet:
    """Eval-mode forward of a ``PWCLONet`` on the fused kernels (point-major activations).

    Built from (and sharing nothing mutable with) an existing module: weights are folded and
    packed once; call again after loading a new ``state_dict``.  The siamese pyramid runs both
    frames as one batch, the level-4 FPS of ``flow_feature_encoding`` reuses ``psa_4``'s (same
    cloud, same result: SURVEY.md appendix B) and the two set-upconvs of a level share one
    neighbour search (identical inputs)."""

    def __init__(self, net):
        from .pwclonet import PWCLO_utils as pw
        assert not net.training, "the fused path implements eval-mode semantics"
        import os
        self.branch = os.environ.get("PWCLO_BRANCH", "1") != "0"   # fork/join streams under graph capture
        # hoisted first layers (per-point partial products, csrc/fused_hoisted.hip); 0 = section-3 kernels
        self.hoist = os.environ.get("PWCLO_HOIST", "1") != "0"
        # Prefix shortcut of the sampling chain (csrc/sampling.hip "Sampling chains"): levels 2..4 are
        # written directly from level 1's tie record (0 = always run the full sampler at every level).
        self.fps_chain = os.environ.get("PWCLO_FPS_CHAIN", "1") != "0"
        SA, UP, CV = ((FusedSAHoisted, FusedUpconvHoisted, FusedCostVolumeHoisted) if self.hoist else
                      (FusedSA, FusedUpconv, FusedCostVolume))
        self.pw = pw
        self.sa = [SA(m) for m in (net.psa_1, net.psa_2, net.psa_3, net.psa_4)]
        self.sa_cfg = [(m.npoint, m.nsample) for m in (net.psa_1, net.psa_2, net.psa_3, net.psa_4)]
        self.cv3 = CV(net.cost_volume)
        self.ffe = SA(net.flow_feature_encoding)
        self.ffe_cfg = (net.flow_feature_encoding.npoint, net.flow_feature_encoding.nsample)
        self.l4_pred = FusedPointwise(net.l4_flow_predictor.mlp_convs, [128, 64])
        self.l4_head = FusedPoseHead(net.pose_calculator_4)
        self.pwr = []
        for lvl, m in ((3, net.pose_warp_refinement_3), (2, net.pose_warp_refinement_2),
                       (1, net.pose_warp_refinement_1)):
            c = m.in_channel[0]
            d = dict(up_f=UP(m.setupconv_features), up_m=UP(m.setupconv_mask), cv=CV(m.cost_volume),
                     pred_f=FusedPointwise(m.flow_predictor_features.mlp_convs, [c, 64, 64]),
                     pred_m=None if m.last_pose_estimation else
                     FusedPointwise(m.flow_predictor_mask.mlp_convs, [64, 64, c]),
                     head=FusedPoseHead(m.pose_calculator), last=m.last_pose_estimation)
            self.pwr.append(d)

    def _refine(self, br, d, row, pose, x1, f1, x2, f2, x1_prev, emb_prev, mask_prev, q_prev, t_prev,
                taps=None, tap="", warped=None, warp_next=None, st_up=None, st_q=None, cvj=None, pres=None, nxt=None,
                carry=None):
        """``warped``: quat_warp_pm(x1, q_prev, t_prev) when the previous level's pose head already produced it;
        ``warp_next``: the next (finer) level's cloud, warped by this level's head with the pose it composes (returned as a
        fifth value).  ``st_up`` / ``st_q``: search structures (``knn_keep``) the pyramid built for the 2B clouds that
        x1_prev (frame 1: clouds [0, B)) and x2 (frame 2: clouds [B, 2B)) belong to, or None.
        Hoisted partial products that something earlier already produced: ``cvj`` = (u, v, u2) of this level's cost volume
        (they need the pyramid features only: ``rest`` computes them beside the set abstractions' seeds), ``pres`` = (pre_f,
        pre_m), the set-upconv seeds (written by the previous level's flow predictors as a linear tail).  ``nxt`` = the next
        (finer) level's modules: when given, this level's predictors write that level's seeds into ``carry["pres"]``."""
        B = x1.shape[0]
        idx_up = br.hold(knn_on(st_up, 0, 8, x1) if st_up is not None and x1.shape[1] >= 256 else knn(8, x1_prev, x1))
        if taps is not None:
            taps[tap + ".up.idx"] = idx_up
        if self.hoist:          # the per-point partial products of this level nobody has produced yet, in one launch
            jobs = ([] if pres is not None else d["up_f"].jobs(emb_prev) + d["up_m"].jobs(mask_prev)) + \
                   ([] if cvj is not None else d["cv"].jobs(f1, f2))
            outs = run_linear_jobs(jobs) if jobs else []
            br.hold(*outs, None)
            pre_f, pre_m = pres if pres is not None else outs[:2]
            u, v, u2 = cvj if cvj is not None else outs[-3:]
        one_launch = self.hoist and upconv_post_supported((d["up_f"], d["up_m"]), x1.shape[0], x1.shape[1])
        if one_launch:          # both set-upconvs and their post-MLPs as one launch, beside the warp -> cost-volume chain
            with br.fork(1):
                up_feat, up_mask = br.hold(*run_upconv_post((d["up_f"], d["up_m"]), x1, x1_prev, f1, (pre_f, pre_m), idx_up))
        else:
            with br.fork(1):        # set-upconv of the features ...
                up_feat = br.hold(d["up_f"](x1, x1_prev, f1, pre_f if self.hoist else emb_prev, idx_up))
            with br.fork(2):        # ... and of the mask are independent of the warp -> cost-volume chain
                up_mask = br.hold(d["up_m"](x1, x1_prev, f1, pre_m if self.hoist else mask_prev, idx_up))
        if warped is None:
            warped = br.hold(quat_warp_pm(x1, q_prev, t_prev))
        if taps is not None:
            taps[tap + ".warped"] = warped
        idx_q = None
        if self.hoist and st_q is not None and warped.shape[1] >= 256:
            idx_q = br.hold(knn_on(st_q, B, d["cv"].nsample_q, warped))
        resid = br.hold(d["cv"](warped, x2, u, v, u2, idx_q=idx_q, taps=taps, tap=tap + ".cv") if self.hoist else
                        d["cv"](warped, f1, x2, f2, taps=taps, tap=tap + ".cv"))
        br.join(1)
        if not one_launch:
            br.join(2)
        if (nxt is not None and carry is not None and not d["last"] and d["pred_f"].tail_supported(nxt["up_f"].pre_job)
                and d["pred_m"].tail_supported(nxt["up_m"].pre_job)):
            emb, pre_f_next = d["pred_f"].with_tail(nxt["up_f"].pre_job, f1, resid, up_feat)
            mask, pre_m_next = d["pred_m"].with_tail(nxt["up_m"].pre_job, up_mask, emb, f1)
            carry["pres"] = br.hold(pre_f_next, pre_m_next)
        else:
            emb = d["pred_f"](f1, resid, up_feat)
            mask = up_mask if d["last"] else d["pred_m"](up_mask, emb, f1)
        if warp_next is not None:
            q, t, w_next = br.hold(*d["head"](emb, mask, pose, row, q_prev, t_prev, warp_next=warp_next))
            return q, t, emb, mask, w_next
        q, t = d["head"](emb, mask, pose, row, q_prev, t_prev)
        return q, t, emb, mask

    @torch.no_grad()
    def sample(self, xyz_f1, xyz_f2, br=None):
        """Stage 1 -- everything that depends on the input coordinates only: both frames point-major
        in one (2B,N,3) batch and the furthest-point-sampling chain of the four pyramid levels.
        Returns the state ``rest`` consumes.  (Split out so that a pipeline can run the sampling
        chains of successive batches back to back on one stream: graphed.StagedPipeline.)"""
        B, _, N0 = xyz_f1.shape
        x = torch.empty((2 * B, N0, 3), dtype=torch.float32, device=xyz_f1.device)  # both frames, point-major
        _lib.call("ingest_pairs_kernel_wrapper", x.device, B, N0, _p(xyz_f1.contiguous()),
                  _p(xyz_f2.contiguous()), _p(x))
        return self._sample_chain(B, x, br)

    @torch.no_grad()
    def sample_frames(self, frame1, frame2, num_points, br=None):
        """Stage 1 from the prediction module's inputs: two (B, n_total, c>=3) point-major frames, of
        which ``[:, :num_points, :3]`` is used (prediction_modules.py:144-160) -- one pass, no permutes."""
        B, n_total, c = frame1.shape
        assert frame2.shape == frame1.shape and c >= 3 and n_total >= num_points
        x = torch.empty((2 * B, num_points, 3), dtype=torch.float32, device=frame1.device)
        _lib.call("ingest_frames_kernel_wrapper", x.device, B, num_points, n_total, c, _p(frame1.contiguous()),
                  _p(frame2.contiguous()), _p(x))
        return self._sample_chain(B, x, br)

    def _sample_chain(self, B, x, br):
        if br is None:
            br = _Branches(x.device, False)
        # The sampling chain of all four levels depends only on the input cloud: it runs ahead on its
        # own branch while the main branch does neighbour search + MLP level by level.
        # Levels 2..4 sample the previous level's samples: level 1 records whether any of its first
        # decisions was an exact tie; where none was, the later levels are prefixes (fast path in the
        # sampler, identical output -- csrc/sampling.hip "Sampling chains").
        samples, ready = [], []
        npoints = [n for n, _ in self.sa_cfg]
        chain = self.fps_chain and all(a >= b_ for a, b_ in zip(npoints, npoints[1:])) and len(npoints) > 1
        flag = torch.empty((x.shape[0], FPS_CHAIN_INTS), dtype=torch.int32, device=x.device) if chain else None
        ws0 = None
        with br.fork(0):
            src = x
            for lvl, npoint in enumerate(npoints):
                if lvl == 0 and fps_slab_supported(src.shape[1]):
                    _, src, ws0 = fps_slab_with_xyz(src, npoint, tie_out=flag, tie_iters=npoints[1] if chain else 0)
                    br.hold(src, *ws0)
                elif chain and lvl == 0:
                    _, src = br.hold(*fps_with_xyz(src, npoint, tie_out=flag, tie_iters=npoints[1]))
                elif chain:
                    _, src = br.hold(*fps_with_xyz(src, npoint, prefix_in=flag))
                else:
                    _, src = br.hold(*fps_with_xyz(src, npoint))
                samples.append(src)
                ready.append(br.mark(0))
            if flag is not None:
                br.hold(flag)
        return dict(B=B, x=x, samples=samples, ready=ready, br=br, ws0=ws0)

    @torch.no_grad()
    def rest(self, state, return_intermediates=False):
        """Stage 2 -- neighbour search, feature pyramid, cost volumes, pose refinement."""
        B, x, samples, ready, br = state["B"], state["x"], state["samples"], state["ready"], state["br"]
        f = None
        lv = []
        reuse = os.environ.get("PWCLO_KNN_REUSE", "1") != "0"
        built = {}                       # pyramid level -> search structure of its 2B clouds (knn_keep)
        # neighbour lists of every knn call (tests compare them with the oracle's); only when asked for
        taps = {} if return_intermediates else None
        early_users, cvj = {}, {}
        if self.hoist and os.environ.get("PWCLO_EARLY_CV", "1") != "0":
            early_users = {1: [self.pwr[2]["cv"]], 2: [self.pwr[1]["cv"]], 3: [self.cv3, self.pwr[0]["cv"]]}
        tails = self.hoist and os.environ.get("PWCLO_PW_TAIL", "1") != "0"
        for lvl, (fsa, (npoint, nsample)) in enumerate(zip(self.sa, self.sa_cfg)):
            br.wait(ready[lvl])
            new_x = samples[lvl]
            if lvl == 0 and state.get("ws0") is not None:     # the sampler already built this cloud's search structure
                idx = br.hold(knn_prebuilt(nsample, x, new_x, state["ws0"]))
            elif reuse:
                # keep the structure built for this level's candidate clouds (both frames): the refinement levels search
                # one frame's half of it again (set-upconv lists: frame 1; cost-volume lists: frame 2)
                idx, st = knn_keep(nsample, x, new_x)
                br.hold(idx)
                if st is not None:
                    built[lvl] = st              # structure of pyramid cloud `lvl` (0 = the input clouds)
                    br.hold(st[0])
            else:
                idx = br.hold(knn(nsample, x, new_x))
            if taps is not None:
                taps["psa_%d.knn_idx" % (lvl + 1)] = idx
            idx_last = idx
            if self.hoist:
                pre = None
                if f is not None:
                    jobs = fsa.jobs(f)
                    # `f` = the features of pyramid level `lvl` (both frames): the partial products the cost volumes take
                    # from them depend on nothing else -- same launch as this level's set-abstraction seeds
                    users = early_users.get(lvl, [])
                    for cv in users:
                        jobs = jobs + cv.jobs(f[:B], f[B:])
                    outs = run_linear_jobs(jobs)
                    br.hold(*outs, None)
                    pre = outs[0]
                    for i, cv in enumerate(users):
                        cvj[id(cv)] = tuple(outs[1 + 3 * i:4 + 3 * i])
                f = br.hold(fsa(x, new_x, pre, idx))
            else:
                f = br.hold(fsa(x, new_x, f, idx))
            x = new_x
            lv.append((x, f))
        (x11, f11), (x12, f12), (x13, f13), (x14, f14) = [(a[:B], b[:B]) for a, b in lv]
        (x21, f21), (x22, f22), (x23, f23), _ = [(a[B:], b[B:]) for a, b in lv]

        # flow_feature_encoding samples the same cloud as psa_4(frame 1): reuse x14
        if self.ffe_cfg == self.sa_cfg[3] and os.environ.get("PWCLO_FFE_REUSE", "1") != "0":
            # same search as psa_4's on frame 1 (queries x14 among x13, same nsample; clouds are searched independently)
            idx_ffe = idx_last[:B]
        else:
            idx_ffe = knn(self.ffe_cfg[1], x13, x14)
        if self.hoist:
            flow = self.cv3(x13, x23, *(cvj.get(id(self.cv3)) or run_linear_jobs(self.cv3.jobs(f13, f23))), taps=taps, tap="cv3")
            emb4 = self.ffe(x13, x14, run_linear_jobs(self.ffe.jobs(flow))[0], idx_ffe)
        else:
            flow = self.cv3(x13, f13, x23, f23, taps=taps, tap="cv3")
            emb4 = self.ffe(x13, x14, flow, idx_ffe)
        if taps is not None:
            taps["ffe.knn_idx"] = idx_ffe
        mask4 = self.l4_pred(f14, emb4)
        pose = torch.empty((B, 4, 7), dtype=torch.float32, device=x.device)   # rows = levels 1..4
        if os.environ.get("PWCLO_HEAD_WARP", "1") != "0":
            # every pose head also warps the next finer cloud with the pose it has just composed (one launch fewer per level)
            q4, t4, w3 = br.hold(*self.l4_head(emb4, mask4, pose, 3, warp_next=x13))
            c3, c2 = {}, {}
            q3, t3, emb3, mask3, w2 = self._refine(br, self.pwr[0], 2, pose, x13, f13, x23, f23, x14, emb4, mask4, q4, t4,
                                                   taps, "pwr3", warped=w3, warp_next=x12, st_up=built.get(4), st_q=built.get(3),
                                                   cvj=cvj.get(id(self.pwr[0]["cv"])), nxt=self.pwr[1] if tails else None, carry=c3)
            q2, t2, emb2, mask2, w1 = self._refine(br, self.pwr[1], 1, pose, x12, f12, x22, f22, x13, emb3, mask3, q3, t3,
                                                   taps, "pwr2", warped=w2, warp_next=x11, st_up=built.get(3), st_q=built.get(2),
                                                   cvj=cvj.get(id(self.pwr[1]["cv"])), pres=c3.get("pres"),
                                                   nxt=self.pwr[2] if tails else None, carry=c2)
            q1, t1, emb1, mask1 = self._refine(br, self.pwr[2], 0, pose, x11, f11, x21, f21, x12, emb2, mask2, q2, t2,
                                               taps, "pwr1", warped=w1, st_up=built.get(2), st_q=built.get(1),
                                               cvj=cvj.get(id(self.pwr[2]["cv"])), pres=c2.get("pres"))
        else:
            q4, t4 = self.l4_head(emb4, mask4, pose, 3)
            q3, t3, emb3, mask3 = self._refine(br, self.pwr[0], 2, pose, x13, f13, x23, f23, x14, emb4, mask4, q4, t4,
                                               taps, "pwr3", st_up=built.get(4), st_q=built.get(3), cvj=cvj.get(id(self.pwr[0]["cv"])))
            q2, t2, emb2, mask2 = self._refine(br, self.pwr[1], 1, pose, x12, f12, x22, f22, x13, emb3, mask3, q3, t3,
                                               taps, "pwr2", st_up=built.get(3), st_q=built.get(2), cvj=cvj.get(id(self.pwr[1]["cv"])))
            q1, t1, emb1, mask1 = self._refine(br, self.pwr[2], 0, pose, x11, f11, x21, f21, x12, emb2, mask2, q2, t2,
                                               taps, "pwr1", st_up=built.get(2), st_q=built.get(1), cvj=cvj.get(id(self.pwr[2]["cv"])))
        if return_intermediates:
            return pose, dict(x11=x11, f11=f11, f13=f13, flow=flow, emb4=emb4, mask4=mask4, emb3=emb3,
                              mask3=mask3, emb2=emb2, mask2=mask2, emb1=emb1, mask1=mask1, q=(q1, q2, q3, q4),
                              t=(t1, t2, t3, t4), lists=taps)
        return pose

    @torch.no_grad()
    def __call__(self, xyz_f1, xyz_f2, return_intermediates=False):
        """xyz_f1, xyz_f2 (B,3,N) -> pose_params (B,4,7) [+ dict of point-major intermediates]."""
        br = _Branches(xyz_f1.device, self.branch and torch.cuda.is_current_stream_capturing())
        return self.rest(self.sample(xyz_f1, xyz_f2, br), return_intermediates)

    @torch.no_grad()
    def forward_frames(self, frame1, frame2, num_points, return_intermediates=False):
        """The forward from two (B, n_total, c>=3) point-major frames (see ``sample_frames``)."""
        br = _Branches(frame1.device, self.branch and torch.cuda.is_current_stream_capturing())
        return self.rest(self.sample_frames(frame1, frame2, num_points, br), return_intermediates)
