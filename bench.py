#!/usr/bin/env python3
"""Headline benchmark: PWCLO-Net forward frame-pairs/s on 2x8192-point KITTI-shaped pairs,
batch 32 per GPU, fp32, eval mode (BASELINE.json metric / configs[2]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A step = one forward over one batch of 32 synthetic frame pairs already
resident in HBM.  The path has no cross-rank exchange in forward (independent frame pairs), so
ranks are pure replicas ("weak" scaling); the only collective is the max-over-ranks of the timed
region.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pwclonet_pylidarslam_amd import _lib, synthetic  # noqa: E402
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3


def make_batch(batch, npoints, seed, device):
    """`batch` KITTI-shaped pairs -> two (B,3,N) fp32 tensors on `device`.  8 distinct ray-cast
    scenes, the rest are jittered copies (ray casting 32 scenes on the host would dominate the
    bench's wall clock; the kernels' work does not depend on which scene a cloud comes from)."""
    scenes = min(batch, 8)
    pc1, pc2, _, _ = synthetic.kitti_like_pair(seed, npoints, scenes)
    rng = np.random.default_rng(seed + 1)
    reps = (batch + scenes - 1) // scenes
    out = []
    for pc in (pc1, pc2):
        x = np.concatenate([pc[:, :, :3] + (rng.normal(0, 2e-3, pc[:, :, :3].shape) if r else 0.0)
                            for r in range(reps)], axis=0)[:batch].astype(np.float32)
        out.append(torch.from_numpy(np.ascontiguousarray(x)).permute(0, 2, 1).contiguous().to(device))
    return out[0], out[1]


class KernelTimer:
    """HIP-event timing of one launcher of the C ABI, on the stream it is launched on (torch's
    current stream -- _lib.call forwards that stream to the library)."""

    def __init__(self, name, bytes_fn):
        self.name, self.bytes_fn = name, bytes_fn
        self.events, self.bytes = [], 0.0
        self._orig = None

    def __enter__(self):
        self._orig = _lib.call
        orig, me = self._orig, self

        def call(name, device, *args):
            if name != me.name:
                return orig(name, device, *args)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig(name, device, *args)
            e.record()
            me.events.append((s, e))
            me.bytes += me.bytes_fn(*args)

        _lib.call = call
        return self

    def __exit__(self, *exc):
        _lib.call = self._orig

    def result(self):
        ms = sum(s.elapsed_time(e) for s, e in self.events)
        return len(self.events), ms, self.bytes


def group_points_bytes(b, c, n, s, k, *ptrs):
    return 4.0 * b * (s * k + c * n + c * s * k)  # idx + source + out


def cpu_baseline(net, npoints, pairs):
    """Oracle (CPU restatement of the reference path, bit-identical to the imported reference) on
    `pairs` sequential B=1 2xN pairs.  Baseline only."""
    from oracle import model as omodel
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    x1, x2 = make_batch(pairs, npoints, 999, torch.device("cpu"))
    omodel.pwclonet_forward(sd, x1[:1], x2[:1])  # warm-up
    t0 = time.perf_counter()
    for i in range(pairs):
        omodel.pwclonet_forward(sd, x1[i:i + 1], x2[i:i + 1])
    dt = time.perf_counter() - t0
    return {"value": pairs / dt, "unit": "frame-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d sequential B=1 2x%d-pt pairs, oracle.model (torch CPU convs on %d threads, "
                      "C ext ops + knn single-threaded)" % (pairs, npoints, torch.get_num_threads())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frame pairs per GPU per step")
    ap.add_argument("--npoints", type=int, default=8192)
    ap.add_argument("--log-mode", default="host", choices=["host", "device", "none"],
                    help="host = the reference's in-forward D2H log_dict (default)")
    ap.add_argument("--launch", default="graph", choices=["graph", "eager"],
                    help="graph = replay one captured hipGraph per step (default); eager = Python launches")
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight (graph launch only): 2 = one batch's FPS chain overlaps the "
                         "other's neighbour-search/MLP kernels; 1 = strictly serial steps")
    ap.add_argument("--unfused", action="store_true",
                    help="reference-shaped module graph on the HIP ops (torch conv/BN) instead of the fused kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=4)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    _lib.load()
    torch.manual_seed(1234)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False,
                        log_mode=args.log_mode)).to(dev).eval()
    if not args.unfused:
        net.prepare_fused()
    x1, x2 = make_batch(args.batch, args.npoints, 1000 + rank, dev)

    pipe = None
    if args.launch == "graph" and args.inflight > 1:
        from pwclonet_pylidarslam_amd.graphed import PipelinedForward
        pipe = PipelinedForward(net, depth=args.inflight)

        def step():
            return pipe(x1, x2)[0]
    elif args.launch == "graph":
        from pwclonet_pylidarslam_amd.graphed import GraphedForward
        graphed = GraphedForward(net)

        def step():
            return graphed(x1, x2)
    else:
        def step():
            with torch.no_grad():
                pose, _ = net(x1, None, x2, None)
            return pose

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timer = KernelTimer("group_points_kernel_wrapper", group_points_bytes)
    fence()
    t0 = time.perf_counter()
    with timer:
        for _ in range(args.steps):
            pose = step()
    fence()
    dt = time.perf_counter() - t0
    assert torch.isfinite(pose).all()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        launches, ms, nbytes = timer.result()
        achieved = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        out = {
            "metric": "PWCLO-Net forward frame-pairs/sec, 2x8192-pt KITTI pair, batch 32",
            "value": world * args.batch * args.steps / dt, "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: synthetic KITTI-shaped 2x%d-pt pairs, "
                                   "batch %d per GPU, full 4-level pyramid, eval mode, fp32"
                                   % (args.npoints, args.batch),
                       "global_batch": world * args.batch, "npoints": args.npoints,
                       "parallelism": "replicas x%d (no forward collective)" % world,
                       "log_dict": args.log_mode if args.launch == "eager" else "device (graph replay)",
                       "launch": args.launch, "batches_in_flight": args.inflight if pipe else 1,
                       "kernels": "module graph + torch conv/BN" if args.unfused
                       else "fused gather+MFMA-MLP kernels (BN folded)"},
            "roofline": {"kernel": "group_points_kernel", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "launches": launches,
                         "avg_launch_us": 1e3 * ms / max(launches, 1),
                         "algorithmic_bytes_per_step": nbytes / max(args.steps, 1)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(net, args.npoints, args.cpu_pairs)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
