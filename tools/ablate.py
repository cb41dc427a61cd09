#!/usr/bin/env python3
"""Developer tool: step time of the fused forward with one kernel family replaced by a trivial
stand-in (wrong outputs, same shapes) -- shows how much each family contributes to the
pipelined step time, including contention effects.  Not a benchmark."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pwclonet_pylidarslam_amd
pwclonet_pylidarslam_amd.configure_hw_queues(8)
import bench
from pwclonet_pylidarslam_amd import fused, _lib
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
from pwclonet_pylidarslam_amd.graphed import PipelinedForward, GraphedForward

dev = torch.device("cuda:0")
orig = dict(fps=fused.fps_with_xyz, knn=fused.knn)

def fake_fps(xyz, npoint, tie_out=None, tie_iters=0, prefix_in=None):
    idx = torch.arange(npoint, device=xyz.device, dtype=torch.int32).unsqueeze(0).expand(xyz.shape[0], -1).contiguous()
    if tie_out is not None:
        tie_out.zero_()
    return idx, xyz[:, :npoint].contiguous()

def fake_knn(k, xyz, new_xyz):
    S = new_xyz.shape[1]
    return (torch.arange(S * k, device=xyz.device, dtype=torch.int32).reshape(1, S, k) % xyz.shape[1]) \
        .expand(xyz.shape[0], -1, -1).contiguous()

def run(tag, inflight):
    torch.manual_seed(0)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device="cuda:0", scalar_last=False,
                        log_mode="none")).to(dev).eval().prepare_fused()
    x1, x2 = bench.make_batch(32, 8192, 1000, dev)
    f = PipelinedForward(net, depth=inflight) if inflight > 1 else GraphedForward(net)
    step = (lambda: f(x1, x2)[0]) if inflight > 1 else (lambda: f(x1, x2))
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    print(f"{tag:24s} inflight={inflight}: {dt*1e3:6.3f} ms/step  {32/dt:7.0f} pairs/s", flush=True)

only_mlp = "--mlp-only" in sys.argv
for inflight in ((1, 2, 3, 4, 6) if only_mlp else (4,)):
    if only_mlp:
        fused.fps_with_xyz, fused.knn = fake_fps, fake_knn; run("no FPS, no knn (MLP+glue)", inflight)
        continue
    fused.fps_with_xyz, fused.knn = orig["fps"], orig["knn"]
    run("full", inflight)
    fused.fps_with_xyz = fake_fps; run("no FPS", inflight); fused.fps_with_xyz = orig["fps"]
    fused.knn = fake_knn; run("no knn", inflight); fused.knn = orig["knn"]
    fused.fps_with_xyz, fused.knn = fake_fps, fake_knn; run("no FPS, no knn (MLP+glue)", inflight)
