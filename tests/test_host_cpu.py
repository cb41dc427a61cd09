"""CPU tests (``-m "not gpu"``) of the boundary and the host logic: the C-ABI library loads and
exports every symbol declared in include/pwclo_ops.h, the Python bindings agree with the header,
the module mirror keeps the reference's state_dict names, weight packing matches an explicit
emulation of the MFMA operand layout, the product refuses CPU tensors (no fallback), and the
2-rank (gloo) benchmark plumbing shards and reduces correctly.
"""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def header_functions():
    text = open(os.path.join(ROOT, "include", "pwclo_ops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"\b(?:void|int|long long|const char \*|void \*)\s*\*?\s*(\w+)\s*\(", text)


def test_library_exports_every_declared_symbol():
    from pwclonet_pylidarslam_amd import _lib, build
    lib_path = build.build()                      # hipcc cross-compiles without a GPU
    names = header_functions()
    assert len(names) >= 25 and "group_points_kernel_wrapper" in names and "knn_point_kernel_wrapper" in names
    lib = ctypes.CDLL(lib_path)
    for n in names:
        assert hasattr(lib, n), "libpwclo_hip.so does not export %s" % n
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)
    loaded = _lib.load()
    assert loaded.pwclo_abi_version() == 1
    assert loaded.pwclo_last_error() == 0


def test_reference_launcher_names_and_arity():
    """The nine launchers keep the reference's names and parameter lists."""
    from pwclonet_pylidarslam_amd import _lib
    expected = {"gather_points_kernel_wrapper": 7, "gather_points_grad_kernel_wrapper": 7,
                "furthest_point_sampling_kernel_wrapper": 6, "group_points_kernel_wrapper": 8,
                "group_points_grad_kernel_wrapper": 8, "query_ball_point_kernel_wrapper": 8,
                "three_nn_kernel_wrapper": 7, "three_interpolate_kernel_wrapper": 8,
                "three_interpolate_grad_kernel_wrapper": 8}
    for name, arity in expected.items():
        assert len(_lib.SIGNATURES[name][0]) == arity, name


def test_no_cpu_fallback_and_host_checks():
    from pwclonet_pylidarslam_amd.pointnet2_ops import _ext, pointnet2_utils, pytorch_utils
    x = torch.rand(1, 16, 3)
    for fn in (lambda: _ext.furthest_point_sampling(x, 4),
               lambda: _ext.knn_point(2, x, x),
               lambda: pointnet2_utils.furthest_point_sample(x, 4),
               lambda: pytorch_utils.knn_point(2, x, x),
               lambda: _ext.group_points(x.transpose(1, 2).contiguous(), torch.zeros(1, 2, 2, dtype=torch.int32))):
        with pytest.raises(RuntimeError, match="CPU not supported"):
            fn()
    with pytest.raises(RuntimeError, match="must be a float tensor"):
        _ext.gather_points(x.double(), torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="must be an int tensor"):
        _ext.gather_points(x, torch.zeros(1, 2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="contiguous"):
        _ext.gather_points(x.transpose(1, 2), torch.zeros(1, 2, dtype=torch.int32))


def test_module_mirror_keeps_reference_state_dict():
    from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device="cpu", scalar_last=False))
    ref = json.load(open(os.path.join(GOLDEN, "state_shapes.json")))
    sd = net.state_dict()
    assert list(sd) == sorted(sd, key=list(sd).index) and set(sd) == set(ref)
    assert all(list(sd[k].shape) == ref[k] for k in ref)
    assert sum(p.numel() for p in net.parameters()) == 775068
    with pytest.raises(RuntimeError, match="CPU not supported"):
        net.eval()(torch.rand(1, 3, 128), None, torch.rand(1, 3, 128), None)
    # a wrapper API detail the trainer relies on: forward returns (pose_params, log_dict)
    assert net.nb_levels == 4 and net.log_mode == "host"


def test_bn_folding_and_weight_packing_layout():
    """pack_layer output == the operand each MFMA lane expects (csrc/mlp_core.hpp), checked by an
    explicit emulation of v_mfma_f32_16x16x4_f32's A/B/D lane maps."""
    from pwclonet_pylidarslam_amd import fused
    from pwclonet_pylidarslam_amd.pointnet2_ops import pytorch_utils as pt
    torch.manual_seed(0)
    mlp = pt.SharedMLP([19, 24, 32], bn=True).eval()
    for layer in mlp:
        layer.bn.bn.running_mean.uniform_(-1, 1)
        layer.bn.bn.running_var.uniform_(0.5, 2)
        layer.bn.bn.weight.data.uniform_(0.5, 1.5)
        layer.bn.bn.bias.data.uniform_(-1, 1)
    x = torch.randn(1, 19, 40, 1)
    with torch.no_grad():
        ref1 = mlp.layer0(x)
        ref2 = mlp(x)
    w, b = fused.fold_conv_bn(mlp.layer0)
    torch.testing.assert_close(F.relu(torch.einsum("oc,bcnk->bonk", w, x) + b.view(1, -1, 1, 1)), ref1,
                               rtol=1e-5, atol=1e-5)
    # physical order: block 0 = [ch0,ch1,ch2, pad...], block 1.. = ch 3..18  (the SA layout)
    phys = fused.sa_first_map(16)
    packed, widths = fused.pack_stack(mlp, phys)
    assert widths == [32, 32]

    def run_layer(buf, nbi, nbo, act):            # act: (16*nbi, npix) physical channels
        wts = buf[:nbo * nbi * 256].view(nbo, nbi, 64, 4)
        bias = buf[nbo * nbi * 256:nbo * nbi * 256 + nbo * 16]
        out = torch.zeros(16 * nbo, act.shape[1])
        for o in range(nbo):
            for lane in range(64):
                row, g = lane % 16, lane // 16   # A operand: lane holds W[row][k = g] of each k-step
                for m in range(nbi):
                    for r in range(4):
                        out[16 * o + row] += wts[o, m, lane, r] * act[16 * m + 4 * g + r]
            out[16 * o:16 * o + 16] += bias[16 * o:16 * o + 16, None]
        return F.relu(out)

    act = torch.zeros(32, 40)
    for p, c in enumerate(phys):
        if c >= 0:
            act[p] = x[0, c, :, 0]
    n1 = fused.layer_floats(2, 2) if hasattr(fused, "layer_floats") else 2 * 2 * 256 + 32
    h1 = run_layer(packed[:n1], 2, 2, act)
    torch.testing.assert_close(h1[:24], ref1[0, :, :, 0], rtol=1e-5, atol=1e-5)
    assert (h1[24:] == 0).all()                   # padded output channels stay exactly zero
    h2 = run_layer(packed[n1:], 2, 2, h1)
    torch.testing.assert_close(h2, ref2[0, :, :, 0], rtol=1e-5, atol=1e-5)


def test_synthetic_generators():
    from pwclonet_pylidarslam_amd import synthetic
    a1, a2, q, t = synthetic.kitti_like_pair(5, 2048, 2)
    b1, _, _, _ = synthetic.kitti_like_pair(5, 2048, 2)
    assert a1.shape == (2, 2048, 4) and a1.dtype == np.float32 and np.array_equal(a1, b1)
    # the reference's filter_pcd (kitti_odometry_dataset.py:149-172): no ground, 30 m box
    for pc in (a1, a2):
        assert (pc[..., 1] <= 1.1).all() and (np.abs(pc[..., 0]) < 30).all() and (np.abs(pc[..., 2]) < 30).all()
        for cloud in pc:
            assert len(np.unique(cloud[:, :3], axis=0)) == 2048          # duplicate free
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
    u1, u2 = synthetic.uniform_pair(9, 512, 3)
    assert u1.shape == (3, 512, 4) and not np.array_equal(u1, u2)


def test_graph_wrappers_require_eval_mode():
    from pwclonet_pylidarslam_amd.graphed import GraphedForward
    from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device="cpu", scalar_last=False))
    with pytest.raises(AssertionError):
        GraphedForward(net.train())
    net.eval()
    assert net._fused is None
    net.train()
    assert net._fused is None


WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from pwclonet_pylidarslam_amd import dist_util
rank, world = dist_util.init("gloo")
lo, hi = dist_util.shard(65, rank, world)
dist_util.fence(torch.device("cpu"))
slow = dist_util.max_over_ranks(1.0 + rank)          # rank 1 is the slow one
total = dist_util.sum_over_ranks(hi - lo)
print("RESULT", rank, world, lo, hi, slow, total, flush=True)
dist_util.finish()
"""


def test_two_rank_gloo_benchmark_plumbing(tmp_path):
    """world_size 2 on CPU: disjoint shards covering the global batch, MAX-reduced step time."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    rows = sorted(l.split()[1:] for o in outs for l in o.splitlines() if l.startswith("RESULT"))
    assert len(rows) == 2, outs
    (r0, w0, lo0, hi0, s0, t0), (r1, w1, lo1, hi1, s1, t1) = rows
    assert (w0, w1) == ("2", "2") and (lo0, hi0, lo1, hi1) == ("0", "33", "33", "65")
    assert float(s0) == float(s1) == 2.0 and float(t0) == float(t1) == 65.0


def test_prediction_module_adapter_contract():
    """Input adapter of the prediction module (prediction_modules.py:130-154): dict keys numpy_pc_{i},
    list of two frames, and the reference's two RuntimeErrors; state_dict keys carry the `pwclonet.` prefix."""
    import pytest
    import torch
    from pwclonet_pylidarslam_amd.prediction import PWCLONetPredictionModule
    mod = PWCLONetPredictionModule(dict(device="cpu", num_input_channels=3, sequence_len=2, num_points=64,
                                        nb_levels=4, scalar_last=False, posenet_config={}))
    a, b = torch.zeros(2, 80, 3), torch.ones(2, 80, 3)
    f = mod._frames({"numpy_pc_0": a, "numpy_pc_1": b, "other": 1})
    assert f[0] is a and f[1] is b
    f = mod._frames([a, b])
    assert f[0] is a and f[1] is b
    with pytest.raises(RuntimeError, match="key `numpy_pc_1` not found"):
        mod._frames({"numpy_pc_0": a})
    with pytest.raises(RuntimeError, match="either dict or list"):
        mod._frames((a, b))
    with pytest.raises(AssertionError):
        PWCLONetPredictionModule(dict(device="cpu", sequence_len=3))
    keys = list(mod.state_dict().keys())
    assert len(keys) == 510 and all(k.startswith("pwclonet.") for k in keys)
    with pytest.raises(RuntimeError, match="CPU not supported"):       # no CPU path in the product
        mod.eval()([a, b])


def test_pointnet2_ops_package_name_and_setup():
    """SURVEY section 8b "package / build": the names the reference imports (pointnet2_ops, pointnet2_ops._ext,
    pointnet2_ops.pointnet2_utils ...; P2/__init__.py:1-3, pointnet2_utils.py:7-9) resolve to this implementation's
    modules themselves, the nine pybind names exist, and setup.py declares the same distribution name."""
    import subprocess
    import sys
    import pointnet2_ops
    import pointnet2_ops._ext as ext
    import pointnet2_ops.pointnet2_modules as mods
    from pointnet2_ops import pointnet2_utils, pytorch_utils
    import pwclonet_pylidarslam_amd.pointnet2_ops as ours
    assert ext is ours._ext and pointnet2_utils is ours.pointnet2_utils
    assert mods is ours.pointnet2_modules and pytorch_utils is ours.pytorch_utils
    assert pointnet2_utils._ext is ext
    for name in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn", "three_interpolate",
                 "three_interpolate_grad", "ball_query", "group_points", "group_points_grad"):   # bindings.cpp:6-19
        assert callable(getattr(ext, name)), name
    assert hasattr(mods, "PointnetSAModulePWCLONet") and hasattr(mods, "PointnetFPModulePWCLONet")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    out = subprocess.run([sys.executable, "setup.py", "--name", "--version"], cwd=root, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[-2:] == ["pointnet2_ops", pointnet2_ops.__version__]


# ---- multi-GPU launch path and the data-parallel training unit ---------------------

def test_bench_spawns_its_own_ranks_when_not_under_torchrun():
    """`python bench.py --gpus 2` with no torchrun around it: the parent starts two rank processes (before any
    GPU call), they rendezvous on 127.0.0.1, fence, MAX-reduce the timed region, and rank 0 prints ONE line.
    --dry-run replaces the GPU step by a host sleep over gloo, so the plumbing runs in the CPU container."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    row = json.loads(lines[0])
    assert row["dry_run"] is True and row["n_gpus"] == 2 and row["steps"] == 3
    assert row["global_pairs"] == 2 * 32 * 3            # both ranks' shards were summed
    assert row["ms_per_step"] >= 2.0                    # the slower rank (2 ms sleeps) defines the step time


def test_bench_refuses_more_gpus_than_visible():
    """Without enough devices the supervising process stops with a clear message before spawning anything."""
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 2 and "GPU(s) visible" in out.stderr, (out.returncode, out.stderr)


DDP_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch
import torch.nn as nn
from pwclonet_pylidarslam_amd import dist_util
from pwclonet_pylidarslam_amd.loss import PWCLONetLossModule
from pwclonet_pylidarslam_amd.training import PWCLONetWithLoss, ddp_wrap, gradient_bucket_values

class StandIn(nn.Module):            # a CPU stand-in with the network's call signature (the product has no CPU path)
    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(6, 28)
    def forward(self, x1, p1, x2, p2):
        return self.lin(torch.cat((x1.mean(2), x2.mean(2)), 1)).reshape(-1, 4, 7), {}

rank, world = dist_util.init("gloo")
torch.manual_seed(0)                 # same initial weights on both ranks
unit = PWCLONetWithLoss(StandIn(), PWCLONetLossModule(dict(with_exp_weights=True, init_weights=[0.0, -2.5],
                                                          loss_option="l2_norm", nb_levels=4, scalar_last=False)))
ddp = ddp_wrap(unit)
g = torch.Generator().manual_seed(100 + rank)          # different data per rank
x1, x2 = torch.randn(4, 3, 50, generator=g), torch.randn(4, 3, 50, generator=g)
gt = torch.randn(4, 7, generator=g)
loss, pose, log = ddp(x1, x2, gt)
loss.backward()
s_ddp = unit.loss_module.exp_weighting.s_param.grad.clone()
w_ddp = unit.pwclonet.lin.weight.grad.clone()
# the same step without DDP, local data only
unit.zero_grad()
loss2, _, _ = unit(x1, x2, gt)
loss2.backward()
s_loc = unit.loss_module.exp_weighting.s_param.grad.clone()
both = [torch.zeros_like(s_loc) for _ in range(world)]
torch.distributed.all_gather(both, s_loc)
want = sum(both) / world
print("RESULT", rank, gradient_bucket_values(unit), s_ddp.tolist(), want.tolist(), s_loc.tolist(),
      float(w_ddp.abs().sum()), flush=True)
dist_util.finish()
"""


def test_ddp_unit_reduces_network_and_loss_weights(tmp_path):
    """2 ranks, gloo: under ``ddp_wrap(PWCLONetWithLoss(net, loss))`` the loss module's learnable ``s_param`` takes
    part in the gradient all-reduce (its gradient equals the mean of the ranks' local gradients and is identical
    on both ranks), which the round-1 step -- DDP around the network alone -- did not do."""
    import ast
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    rows = [l for o in outs for l in o.splitlines() if l.startswith("RESULT")]
    assert len(rows) == 2, outs
    import re
    vals = []
    for r in rows:
        lists = [ast.literal_eval(m) for m in re.findall(r"\[[^\]]*\]", r)]
        head = r.split()
        vals.append((int(head[1]), int(head[2]), lists[0], lists[1], lists[2]))
    vals.sort()
    (_, n0, s0, want0, loc0), (_, n1, s1, want1, loc1) = vals
    assert n0 == n1 == 6 * 28 + 28 + 2                    # the stand-in's parameters + the two loss weights
    assert s0 == s1                                       # replicas agree after the all-reduce
    assert all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(s0, want0))
    assert loc0 != loc1                                   # ... although their local gradients differ


def test_training_unit_parameter_count():
    """the gradient message is 775 068 + 2 fp32 values."""
    from pwclonet_pylidarslam_amd.loss import PWCLONetLossModule
    from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
    from pwclonet_pylidarslam_amd.training import PWCLONetWithLoss, gradient_bucket_values
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device="cpu", scalar_last=False))
    unit = PWCLONetWithLoss(net, PWCLONetLossModule(dict(with_exp_weights=True, init_weights=[0.0, -2.5],
                                                         loss_option="l2_norm", nb_levels=4, scalar_last=False)))
    assert gradient_bucket_values(unit) == 775068 + 2
    assert sorted(k for k in unit.state_dict() if "s_param" in k) == ["loss_module.exp_weighting.s_param"]


def test_pointwise_conv_shapes_of_the_network_are_the_tested_ones():
    """tests/test_gpu_conv.py checks csrc/conv1x1.hip on NET_SHAPES: that list is every bias-free 1x1 Conv2d of the
    network (P2/pytorch_utils.py:170-199 inside the SharedMLPs); layers with a bias (the pose heads' Conv1d) stay
    on torch."""
    import importlib.util
    import torch
    from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
    spec = importlib.util.spec_from_file_location("tgc", os.path.join(os.path.dirname(__file__), "test_gpu_conv.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device="cpu", scalar_last=False, log_mode="none"))
    shapes = set()
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            assert m.bias is None and m.kernel_size == (1, 1)
            shapes.add((m.in_channels, m.out_channels))
        elif isinstance(m, torch.nn.Conv1d):
            assert m.bias is not None
    assert shapes == set(mod.NET_SHAPES)


def test_conv_blocks_on_cpu_are_plain_torch():
    """The HIP routes of pytorch_utils (conv1x1, BatchNorm, stack tails) are taken for GPU tensors only: on CPU tensors
    a Conv2d block and shared_mlp_max give exactly what the torch modules give, in train and eval mode -- the package
    never computes on the CPU itself, torch does."""
    import copy
    import torch
    from pwclonet_pylidarslam_amd.pointnet2_ops import pytorch_utils as pt
    from pwclonet_pylidarslam_amd import conv1x1
    torch.manual_seed(0)
    mlp = pt.SharedMLP([19, 16, 32], bn=True)
    ref = copy.deepcopy(mlp)
    x = torch.randn(2, 19, 12, 8)
    for mode in ("train", "eval"):
        getattr(mlp, mode)()
        getattr(ref, mode)()
        want = x
        for layer in ref:                                   # the reference's forward: conv -> bn -> relu per layer
            for m in layer:
                want = m(want)
        assert torch.equal(pt.shared_mlp_max(mlp, x), want.max(dim=3)[0])
    assert not conv1x1.supported(x, mlp[0].conv)            # CPU tensor: never routed to the kernels
    assert conv1x1._wgrad_split(192, 128) and conv1x1._wgrad_split(6, 8)
    assert not conv1x1._wgrad_split(160, 144)               # 9 x 10 tiles: no 8-worker split with <= 4 x 4 per wave
