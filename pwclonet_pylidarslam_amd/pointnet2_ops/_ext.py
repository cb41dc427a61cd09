"""The ``pointnet2_ops._ext`` function surface on top of libpwclo_hip.so.

Mirrors the reference's pybind module (``P2/_ext-src/src/bindings.cpp:6-19``): same nine names,
argument order and return values, and the host-side conventions of its ``.cpp`` files --
dtype / contiguity checks that raise ``RuntimeError`` (``utils.h:5-25``), zero-initialised
outputs on the inputs' device (relied on by the grad kernels and by ball_query's "no hit"
case), "CPU not supported" for host tensors.  Two extra entry points cover the pure-PyTorch
ops of the same path that get native kernels here: ``knn_point`` and ``quat_warp``.

Unlike the reference there is no JIT fallback and no silent path: every function launches a
HIP kernel through the C ABI or raises.
"""
import os as _os

import torch

from .. import _lib


def _chk(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _float(t, name):
    _chk(t.is_contiguous(), name + " must be a contiguous tensor")
    _chk(t.dtype == torch.float32, name + " must be a float tensor")


def _int(t, name):
    _chk(t.is_contiguous(), name + " must be a contiguous tensor")
    _chk(t.dtype == torch.int32, name + " must be an int tensor")


def _gpu(t, *others):
    _chk(t.is_cuda, "CPU not supported")
    for o in others:
        _chk(o.is_cuda and o.device == t.device, "all tensors must be on the same CUDA/HIP device")


def _p(t):
    return t.data_ptr()


def gather_points(points, idx):
    """sampling.cpp:15-38.  (B,C,N) f32, (B,M) i32 -> (B,C,M) f32."""
    _float(points, "points"); _int(idx, "idx"); _gpu(points, idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = torch.zeros((B, C, M), dtype=torch.float32, device=points.device)
    _lib.call("gather_points_kernel_wrapper", points.device, B, C, N, M, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    """sampling.cpp:40-65.  (B,C,M), (B,M), n -> (B,C,n)."""
    _float(grad_out, "grad_out"); _int(idx, "idx"); _gpu(grad_out, idx)
    B, C, M = grad_out.shape
    out = torch.zeros((B, C, n), dtype=torch.float32, device=grad_out.device)
    _lib.call("gather_points_grad_kernel_wrapper", grad_out.device, B, C, n, M, _p(grad_out), _p(idx),
              _p(out))
    return out


# How the large-cloud sampler's spatial order is built: "device" = the hand-written kernels of csrc/sampling.hip
# (fps_spatial_order: 0.4 ms for 16 clouds of 120 000 rows, cells half the size), "torch" = two torch argsorts + a scatter
# (1.5 ms; library sorts).  Either order gives bit-identical samples.  bench.py --config 5 documents the one case that still
# picks "torch" (two sampler streams in flight) and why.
LARGE_CLOUD_ORDER = "torch" if _os.environ.get("PWCLO_FPS_ORDER_TORCH", "0") != "0" else "device"
_PRI_TABLES = {}
_ORDER_BUFFERS = {}      # (device, stream, B, N) -> (workspace, sorted points, permutation) of _fps_sorted_order


def _fps_priorities(n, device):
    """Sampling priority of every ORIGINAL index of an n-point cloud (csrc/sampling.hip: the reference's thread
    partition + tie-keeping tree order ties by (bitrev(k mod bs), k div bs), bs = 2^floor(log2 n) <= 512)."""
    key = (n, str(device))
    if key not in _PRI_TABLES:
        bs = 1
        while bs * 2 <= n and bs < 512:
            bs *= 2
        bits = bs.bit_length() - 1
        rev = torch.tensor([int(format(r, "0%db" % bits)[::-1], 2) if bits else 0 for r in range(bs)], dtype=torch.int64)
        k = torch.arange(n, dtype=torch.int64)
        _PRI_TABLES[key] = ((rev[k % bs] << 23) | (k // bs)).to(device)
    return _PRI_TABLES[key]


def _fps_sorted_order(points):
    """Spatially coherent order for the large-cloud sampler: Morton order of the coordinates (10 bits per axis, per-cloud
    box), cut into cells of 1024 consecutive positions, each cell ordered by ascending sampling priority.
    points (B,N,3) -> (sorted (B,N,3) f32, perm (B,N) i32: original index of every sorted position).  Torch sorts:
    plumbing around the kernel, ~0.3 ms for 16 clouds of 1e5 points against the 8 ms the pruned update saves."""
    B, N, _ = points.shape
    if LARGE_CLOUD_ORDER == "device":
        # hand-written counting sort + per-cell priority sort (csrc/sampling.hip: fps_spatial_order); the torch sorts below
        # are kept as the A/B reference (PWCLO_FPS_ORDER_TORCH=1) -- either order is exact for the sampler
        lib = _lib.load()
        # the three scratch / output buffers are only read by the sampler launched next on THIS stream, so one set per
        # (stream, shape) is reused call after call: with several sampler streams in flight the caching allocator otherwise
        # keeps asking the runtime for memory (blocks handed between streams are not reusable at once) and every such
        # hipMalloc stalls the device
        key = (str(points.device), torch.cuda.current_stream(points.device).cuda_stream, B, N)
        bufs = _ORDER_BUFFERS.get(key)
        if bufs is None or torch.cuda.is_current_stream_capturing():
            bufs = (torch.empty((lib.fps_spatial_order_workspace_bytes(B, N),), dtype=torch.uint8, device=points.device),
                    torch.empty_like(points), torch.empty((B, N), dtype=torch.int32, device=points.device))
            if not torch.cuda.is_current_stream_capturing():
                if len(_ORDER_BUFFERS) >= 8:
                    _ORDER_BUFFERS.clear()
                _ORDER_BUFFERS[key] = bufs
        ws, sorted_pts, perm = bufs
        _lib.call("fps_spatial_order_kernel_wrapper", points.device, B, N, _p(points), _p(sorted_pts), _p(perm), _p(ws))
        return sorted_pts, perm
    lo = points.amin(dim=1, keepdim=True)
    span = (points.amax(dim=1, keepdim=True) - lo).clamp_min(1e-20)
    q = ((points - lo) * (1023.0 / span)).to(torch.int64).clamp_(0, 1023)

    def spread(v):                                  # 10 bits -> bits 0, 3, 6, ...
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    code = spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
    order = torch.argsort(code, dim=1)
    rank = torch.empty_like(order)
    rank.scatter_(1, order, torch.arange(N, device=points.device).unsqueeze(0).expand(B, -1))
    key = ((rank // 1024) << 32) | _fps_priorities(N, points.device).unsqueeze(0)
    perm = torch.argsort(key, dim=1)
    sorted_pts = torch.gather(points, 1, perm.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    return sorted_pts, perm.to(torch.int32).contiguous()


def furthest_point_sampling(points, nsamples):
    """sampling.cpp:66-87.  (B,N,3) f32 -> (B,nsamples) i32.  The reference's (B,N) `tmp` scratch
    is only allocated when the cloud is too large for the register-resident kernel; those clouds (N > 24576) are
    also brought into a spatially coherent order first, which lets the cooperative sampler skip -- exactly -- the
    distance update of every wave whose cell the new sample cannot reach (PWCLO_FPS_SORTED=0: plain order)."""
    _float(points, "points"); _gpu(points)
    B, N, _ = points.shape
    out = torch.zeros((B, nsamples), dtype=torch.int32, device=points.device)
    if N > 24576:
        tmp_t = torch.full((B, N), 1e10, dtype=torch.float32, device=points.device)
        if _os.environ.get("PWCLO_FPS_SORTED", "1") != "0" and N >= 2 * 1024 * 16:
            sorted_pts, perm = _fps_sorted_order(points)
            _lib.call("furthest_point_sampling_sorted_kernel_wrapper", points.device, B, N, nsamples, _p(points),
                      _p(sorted_pts), _p(perm), _p(tmp_t), _p(out), 0)
            return out
        _lib.call("furthest_point_sampling_kernel_wrapper", points.device, B, N, nsamples, _p(points), _p(tmp_t),
                  _p(out))
        return out
    _lib.call("furthest_point_sampling_kernel_wrapper", points.device, B, N, nsamples, _p(points), 0, _p(out))
    return out


def three_nn(unknowns, knows):
    """interpolate.cpp:14-40.  Returns [dist2 (B,n,3) f32, idx (B,n,3) i32]."""
    _float(unknowns, "unknowns"); _float(knows, "knows"); _gpu(unknowns, knows)
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    idx = torch.zeros((B, n, 3), dtype=torch.int32, device=unknowns.device)
    dist2 = torch.zeros((B, n, 3), dtype=torch.float32, device=unknowns.device)
    _lib.call("three_nn_kernel_wrapper", unknowns.device, B, n, m, _p(unknowns), _p(knows), _p(dist2),
              _p(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """interpolate.cpp:42-70.  (B,c,m), (B,n,3) i32, (B,n,3) f32 -> (B,c,n)."""
    _float(points, "points"); _int(idx, "idx"); _float(weight, "weight"); _gpu(points, idx, weight)
    B, c, m = points.shape
    n = idx.shape[1]
    out = torch.zeros((B, c, n), dtype=torch.float32, device=points.device)
    _lib.call("three_interpolate_kernel_wrapper", points.device, B, c, m, n, _p(points), _p(idx),
              _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """interpolate.cpp:71-99.  (B,c,n), idx, weight, m -> (B,c,m)."""
    _float(grad_out, "grad_out"); _int(idx, "idx"); _float(weight, "weight")
    _gpu(grad_out, idx, weight)
    B, c, n = grad_out.shape
    out = torch.zeros((B, c, m), dtype=torch.float32, device=grad_out.device)
    _lib.call("three_interpolate_grad_kernel_wrapper", grad_out.device, B, c, n, m, _p(grad_out),
              _p(idx), _p(weight), _p(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """ball_query.cpp:8-32.  NOTE the (new_xyz, xyz) order.  -> (B,M,nsample) i32."""
    _float(new_xyz, "new_xyz"); _float(xyz, "xyz"); _gpu(new_xyz, xyz)
    B, M, _ = new_xyz.shape
    N = xyz.shape[1]
    idx = torch.zeros((B, M, nsample), dtype=torch.int32, device=new_xyz.device)
    _lib.call("query_ball_point_kernel_wrapper", new_xyz.device, B, N, M, float(radius), int(nsample),
              _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points(points, idx):
    """group_points.cpp:12-36.  (B,C,N) f32, (B,S,K) i32 -> (B,C,S,K)."""
    _float(points, "points"); _int(idx, "idx"); _gpu(points, idx)
    B, C, N = points.shape
    S, K = idx.shape[1], idx.shape[2]
    out = torch.empty((B, C, S, K), dtype=torch.float32, device=points.device)  # fully overwritten
    _lib.call("group_points_kernel_wrapper", points.device, B, C, N, S, K, _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    """group_points.cpp:38-62.  (B,C,S,K), (B,S,K), n -> (B,C,n)."""
    _float(grad_out, "grad_out"); _int(idx, "idx"); _gpu(grad_out, idx)
    B, C, S, K = grad_out.shape
    out = torch.zeros((B, C, n), dtype=torch.float32, device=grad_out.device)
    _lib.call("group_points_grad_kernel_wrapper", grad_out.device, B, C, n, S, K, _p(grad_out), _p(idx),
              _p(out))
    return out


def scatter_grad_deterministic(grad_out, idx, n):
    """Atomics-free ``group_points_grad`` (idx (B,S,K), grad_out (B,C,S,K)) / ``gather_points_grad`` (idx
    (B,M), grad_out (B,C,M)) -> (B,C,n): same sums in a fixed order (ascending output position per
    source point), bit-identical from run to run.  The inverse index is a stable torch sort."""
    _float(grad_out, "grad_out"); _int(idx, "idx"); _gpu(grad_out, idx)
    B, C = grad_out.shape[0], grad_out.shape[1]
    perm, seg = _inverse_index(idx, n)
    P = perm.shape[1]
    out = torch.empty((B, C, n), dtype=torch.float32, device=grad_out.device)
    _lib.call("group_points_grad_sorted_kernel_wrapper", grad_out.device, B, C, n, P, 1, _p(grad_out), _p(perm),
              _p(seg), _p(out))
    return out


def _inverse_index(idx, n):
    """perm (B,P) i32 = positions sorted by source point (stable), seg (B,n+1) i32 = segment starts."""
    B = idx.shape[0]
    flat = idx.reshape(B, -1)
    order = torch.sort(flat, dim=1, stable=True)
    perm = order.indices.to(torch.int32).contiguous()
    bounds = torch.arange(n + 1, device=idx.device, dtype=flat.dtype).unsqueeze(0).expand(B, -1).contiguous()
    seg = torch.searchsorted(order.values.contiguous(), bounds).to(torch.int32).contiguous()
    return perm, seg


def _slice_ptr(stack, c_off, c):
    """Address of channel ``c_off`` of cloud 0 of a contiguous (B,Ctot,S,K) tensor + its batch stride in floats."""
    _float(stack, "stack")
    if not (0 <= c_off and c_off + c <= stack.shape[1]):
        raise ValueError("channel slice [%d, %d) outside a tensor of %d channels" % (c_off, c_off + c, stack.shape[1]))
    P = stack.shape[2] * stack.shape[3]
    return stack.data_ptr() + 4 * c_off * P, stack.shape[1] * P


def group_points_into(points, idx, stack, c_off):
    """``stack[:, c_off:c_off+C] = group_points(points, idx)`` without the intermediate tensor (the concatenation of
    P2/pointnet2_modules.py:222-230 done by the grouping kernel itself)."""
    _float(points, "points"); _int(idx, "idx"); _gpu(points, idx, stack)
    B, C, N = points.shape
    S, K = idx.shape[1], idx.shape[2]
    if tuple(stack.shape[0:1] + stack.shape[2:]) != (B, S, K):
        raise ValueError("stack %s does not match idx %s" % (tuple(stack.shape), tuple(idx.shape)))
    ptr, stride = _slice_ptr(stack, c_off, C)
    _lib.call("group_points_strided_kernel_wrapper", points.device, B, C, N, S, K, _p(points), _p(idx), ptr, stride)


def group_points_grad_from(grad_stack, c_off, c, idx, n, deterministic=False, inverse=None):
    """``group_points_grad(grad_stack[:, c_off:c_off+c].contiguous(), idx, n)`` without the copy.  ``deterministic``: the
    atomics-free kernel (``inverse`` = a cached ``_inverse_index(idx, n)``)."""
    _int(idx, "idx"); _gpu(grad_stack, idx)
    B, S, K = idx.shape
    if tuple(grad_stack.shape[0:1] + grad_stack.shape[2:]) != (B, S, K):
        raise ValueError("grad_stack %s does not match idx %s" % (tuple(grad_stack.shape), tuple(idx.shape)))
    ptr, stride = _slice_ptr(grad_stack, c_off, c)
    if deterministic:
        perm, seg = inverse if inverse is not None else _inverse_index(idx, n)
        out = torch.empty((B, c, n), dtype=torch.float32, device=grad_stack.device)
        _lib.call("group_points_grad_sorted_strided_kernel_wrapper", grad_stack.device, B, c, n, S * K, 1, ptr, stride,
                  _p(perm), _p(seg), _p(out))
        return out
    out = torch.zeros((B, c, n), dtype=torch.float32, device=grad_stack.device)
    _lib.call("group_points_grad_strided_kernel_wrapper", grad_stack.device, B, c, n, S, K, ptr, stride, _p(idx), _p(out))
    return out


def geometry_encode_into(centre_xyz, src_xyz, idx, stack, c_off):
    """``stack[:, c_off:c_off+10]`` = [p, q, q - p, |q - p|] of PW/costvolume.py:92-105 (centre_xyz (B,3,S), src_xyz (B,3,N),
    idx (B,S,K))."""
    _float(centre_xyz, "centre_xyz"); _float(src_xyz, "src_xyz"); _int(idx, "idx"); _gpu(centre_xyz, src_xyz, idx, stack)
    B, S, K = idx.shape
    if centre_xyz.shape != (B, 3, S) or src_xyz.shape[:2] != (B, 3) or tuple(stack.shape[0:1] + stack.shape[2:]) != (B, S, K):
        raise ValueError("geometry_encode: centre %s src %s idx %s stack %s" % (tuple(centre_xyz.shape), tuple(src_xyz.shape),
                                                                                 tuple(idx.shape), tuple(stack.shape)))
    ptr, stride = _slice_ptr(stack, c_off, 10)
    _lib.call("geometry_encode_kernel_wrapper", idx.device, B, src_xyz.shape[2], S, K, _p(centre_xyz), _p(src_xyz), _p(idx), ptr,
              stride)


def xyz_diff_into(centre_xyz, src_xyz, idx, stack, c_off):
    """``stack[:, c_off:c_off+3] = group_points(src_xyz, idx) - centre_xyz.unsqueeze(3)`` (centre_xyz (B,3,S), src_xyz (B,3,N))."""
    _float(centre_xyz, "centre_xyz"); _float(src_xyz, "src_xyz"); _int(idx, "idx"); _gpu(centre_xyz, src_xyz, idx, stack)
    B, S, K = idx.shape
    if centre_xyz.shape != (B, 3, S) or src_xyz.shape[:2] != (B, 3) or tuple(stack.shape[0:1] + stack.shape[2:]) != (B, S, K):
        raise ValueError("xyz_diff: centre %s src %s idx %s stack %s" % (tuple(centre_xyz.shape), tuple(src_xyz.shape),
                                                                          tuple(idx.shape), tuple(stack.shape)))
    ptr, stride = _slice_ptr(stack, c_off, 3)
    _lib.call("xyz_diff_kernel_wrapper", idx.device, B, src_xyz.shape[2], S, K, _p(centre_xyz), _p(src_xyz), _p(idx), ptr, stride)


def geometry_encode_grad_from(grad_stack, c_off, centre_xyz, src_xyz, idx, want_centre=True, want_src=True,
                              deterministic=False, inverse=None):
    """-> (d_centre_xyz (B,3,S) or None, d_src_xyz (B,3,N) or None) from ``grad_stack[:, c_off:c_off+10]``.
    ``deterministic``: the neighbours' gradients go through the atomics-free sorted scatter instead of fp32 atomics."""
    _int(idx, "idx"); _gpu(grad_stack, centre_xyz, src_xyz, idx)
    B, S, K = idx.shape
    N = src_xyz.shape[2]
    ptr, stride = _slice_ptr(grad_stack, c_off, 10)
    dev = idx.device
    dc = torch.empty((B, 3, S), dtype=torch.float32, device=dev) if want_centre else None
    pairs = torch.empty((B, 3, S, K), dtype=torch.float32, device=dev) if (want_src and deterministic) else None
    ds = torch.zeros((B, 3, N), dtype=torch.float32, device=dev) if (want_src and not deterministic) else None
    if dc is not None or ds is not None or pairs is not None:
        p0 = lambda t: _p(t) if t is not None else 0
        _lib.call("geometry_encode_grad_kernel_wrapper", dev, B, N, S, K, _p(centre_xyz), _p(src_xyz), _p(idx), ptr, stride,
                  p0(dc), p0(ds), p0(pairs))
    if pairs is not None:
        ds = group_points_grad_from(pairs, 0, 3, idx, N, deterministic=True, inverse=inverse)
    return dc, ds


def broadcast_centre_into(feats, k, stack, c_off):
    """``stack[:, c_off:c_off+C] = feats.unsqueeze(3).expand(-1, -1, -1, k)`` (feats (B,C,S))."""
    _float(feats, "feats"); _gpu(feats, stack)
    B, C, S = feats.shape
    if tuple(stack.shape[0:1] + stack.shape[2:]) != (B, S, k):
        raise ValueError("broadcast_centre: feats %s stack %s k %d" % (tuple(feats.shape), tuple(stack.shape), k))
    ptr, stride = _slice_ptr(stack, c_off, C)
    _lib.call("broadcast_centre_kernel_wrapper", feats.device, B, C, S, k, _p(feats), ptr, stride)


def broadcast_centre_grad_from(grad_stack, c_off, c):
    """``grad_stack[:, c_off:c_off+c].sum(3)`` without the slice copy."""
    _gpu(grad_stack)
    B, _, S, K = grad_stack.shape
    ptr, stride = _slice_ptr(grad_stack, c_off, c)
    out = torch.empty((B, c, S), dtype=torch.float32, device=grad_stack.device)
    _lib.call("broadcast_centre_grad_kernel_wrapper", grad_stack.device, B, c, S, K, ptr, stride, _p(out))
    return out


# ---- native replacements of pure-PyTorch ops (include/pwclo_ops.h section 2) ---------------------

_KNN_MIN_S = int(_os.environ.get("PWCLO_KNN_MIN_S", "256"))     # fewer queries: the exhaustive kernel (no build pass)


def knn_point(nsample, xyz, new_xyz, return_dist=False, exhaustive=None):
    """Native kernel behind ``pytorch_utils.knn_point``.  xyz (B,N,3), new_xyz (B,S,3) ->
    idx (B,S,nsample) i32 ascending by distance (ties: lower index); optionally the keys."""
    _float(xyz, "xyz"); _float(new_xyz, "new_xyz"); _gpu(xyz, new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty((B, S, nsample), dtype=torch.int32, device=xyz.device)
    dist = torch.empty((B, S, nsample), dtype=torch.float32, device=xyz.device) if return_dist else None
    ws_bytes = _lib.load().knn_point_workspace_bytes(B, N) if (exhaustive is not True and (S >= _KNN_MIN_S or exhaustive is False)) else 0
    if ws_bytes > 0:    # exact spatially pruned search (same output), needs scratch for the sorted rows
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=xyz.device)
        _lib.call("knn_point_ws_kernel_wrapper", xyz.device, B, N, S, int(nsample), _p(xyz), _p(new_xyz),
                  _p(idx), _p(dist) if return_dist else 0, _p(ws))
    else:
        _lib.call("knn_point_kernel_wrapper", xyz.device, B, N, S, int(nsample), _p(xyz), _p(new_xyz),
                  _p(idx), _p(dist) if return_dist else 0)
    return (dist, idx) if return_dist else idx


def quat_warp(xyz, q, t):
    """Native kernel behind ``PWCLO_utils.warp``.  xyz (B,3,N), q (B,4[,1]), t (B,3[,1])."""
    B, _, N = xyz.shape
    q = q.reshape(B, 4).contiguous()
    t = t.reshape(B, 3).contiguous()
    _float(xyz, "xyz"); _float(q, "q"); _float(t, "t"); _gpu(xyz, q, t)
    out = torch.empty_like(xyz)
    _lib.call("quat_warp_kernel_wrapper", xyz.device, B, N, _p(xyz), _p(q), _p(t), _p(out))
    return out
