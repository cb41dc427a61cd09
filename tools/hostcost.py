#!/usr/bin/env python3
"""Developer tool: host-side cost of submitting one captured forward (graph replay) vs GPU time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
from pwclonet_pylidarslam_amd.graphed import PipelinedForward, GraphedForward

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device="cuda:0", scalar_last=False,
                    log_mode="none")).to(dev).eval().prepare_fused()
x1, x2 = bench.make_batch(32, 8192, 1000, dev)
for depth in (1, 2, 3):
    pipe = PipelinedForward(net, depth=depth)
    for _ in range(4): pipe(x1, x2)
    torch.cuda.synchronize()
    n = 24
    t0 = time.perf_counter()
    per = []
    for _ in range(n):
        a = time.perf_counter(); pipe(x1, x2); per.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"depth {depth}: host submit {1e3*(t1-t0)/n:.3f} ms/step (min {1e3*min(per):.3f} max {1e3*max(per):.3f}), "
          f"total {1e3*(t2-t0)/n:.3f} ms/step", flush=True)
