#!/usr/bin/env python3
"""Developer tool: furthest point sampling of large clouds (BASELINE.json configs[4]: 120k -> 8192) --
wall time per call and per iteration for the cooperative multi-workgroup sampler."""
import torch, time, sys
sys.path.insert(0, "/root/repo")
import pwclonet_pylidarslam_amd
from pwclonet_pylidarslam_amd.pointnet2_ops import _ext as E
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (b, n, m) in ((8, 120000, 8192), (8, 24576, 8192), (8, 16384, 8192)):
    x = ((torch.rand(b, n, 3, generator=g) * 2 - 1) * 40).to(dev)
    E.furthest_point_sampling(x, 64); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = E.furthest_point_sampling(x, m); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"fps b={b} n={n} m={m}: {dt*1e3:.1f} ms  {dt/m*1e6:.2f} us/iter  unique={len(torch.unique(out[0]))}", flush=True)
